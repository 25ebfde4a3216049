mkdir -p gpurun_out/r3
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3/gputest_a.txt 2>&1; echo rc=$? >> gpurun_out/r3/gputest_a.txt
tail -15 gpurun_out/r3/gputest_a.txt
python3 bench.py --steps 5 --warmup 1 > gpurun_out/r3/bench_a.json 2> gpurun_out/r3/bench_a.err; tail -c 1500 gpurun_out/r3/bench_a.json
