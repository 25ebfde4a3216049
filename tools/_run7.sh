mkdir -p gpurun_out/r3
timeout -k 10 600 python3 -m pytest tests/test_hip_poisson.py -m gpu -x -q > gpurun_out/r3/gputest_poisson.txt 2>&1; echo rc=$? >> gpurun_out/r3/gputest_poisson.txt
tail -3 gpurun_out/r3/gputest_poisson.txt
timeout -k 10 300 python3 tools/poisson_step.py 2>&1 | tail -2
