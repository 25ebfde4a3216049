mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_hip_wide.py tests/test_hip_backward.py -m gpu -x -q > gpurun_out/r3/gputest_b.txt 2>&1; echo rc=$? >> gpurun_out/r3/gputest_b.txt
tail -8 gpurun_out/r3/gputest_b.txt
bash tools/bench_suite.sh > gpurun_out/r3/bench_suite_a.txt 2>&1
cat gpurun_out/r3/bench_suite_a.txt
bash tools/shape_sweep.sh > gpurun_out/r3/shape_sweep_a.txt 2>&1
tail -30 gpurun_out/r3/shape_sweep_a.txt
