mkdir -p gpurun_out/r3
timeout -k 10 500 python3 tools/fused_check.py > gpurun_out/r3/fused_check5.txt 2>&1; echo rc=$? >> gpurun_out/r3/fused_check5.txt
tail -34 gpurun_out/r3/fused_check5.txt
grep -q "rc=0" gpurun_out/r3/fused_check5.txt || exit 1
bash tools/run_fused_variants.sh nocov none > gpurun_out/r3/fused_variants5.txt 2>&1
cat gpurun_out/r3/fused_variants5.txt
