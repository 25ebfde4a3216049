mkdir -p gpurun_out/r3
python3 tools/fused_check.py --no-time > gpurun_out/r3/fused_check2.txt 2>&1 || { tail -20 gpurun_out/r3/fused_check2.txt; exit 1; }
tail -12 gpurun_out/r3/fused_check2.txt
bash tools/run_fused_variants.sh sched0 sched2 nocov noload none nocov_noload > gpurun_out/r3/fused_variants2.txt 2>&1
cat gpurun_out/r3/fused_variants2.txt
