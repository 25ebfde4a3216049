mkdir -p gpurun_out/r3
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > gpurun_out/r3/gputest_c.txt 2>&1; echo rc=$? >> gpurun_out/r3/gputest_c.txt
tail -12 gpurun_out/r3/gputest_c.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke_c.txt 2>&1; tail -3 gpurun_out/r3/smoke_c.txt
python3 bench.py > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err; tail -c 900 gpurun_out/r3/bench_c.json
