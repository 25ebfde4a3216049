mkdir -p gpurun_out/r2
python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r2/clk_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | head -8; echo ---; sleep 1.5; done > gpurun_out/r2/clk_watch.txt 2>&1
wait $BP
tail -1 gpurun_out/r2/clk_bench.json | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['ms_per_step'],2), round(r['roofline']['achieved'],2))"
head -30 gpurun_out/r2/clk_watch.txt
