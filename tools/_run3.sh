mkdir -p gpurun_out/r3
timeout -k 10 500 python3 tools/fused_check.py > gpurun_out/r3/fused_check3.txt 2>&1; echo rc=$? >> gpurun_out/r3/fused_check3.txt
tail -40 gpurun_out/r3/fused_check3.txt
grep -q "rc=0" gpurun_out/r3/fused_check3.txt || exit 1
bash tools/run_fused_variants.sh nocov noload none > gpurun_out/r3/fused_variants3.txt 2>&1
cat gpurun_out/r3/fused_variants3.txt
