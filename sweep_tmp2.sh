python3 bench.py --config 5 --N 24576 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b5.log 2>&1
python3 - <<PY
import json
r=json.loads(open("gpurun_out/b5.log").read().strip().splitlines()[-1])
print(round(r["ms_per_step"],2), round(r["roofline"]["achieved"],1), round(r["kernels"]["stage2_LuT_Wt"]["achieved_TFLOPs"],1), r["elbo"], {k:(round(v["ms_per_eval"],2) if isinstance(v,dict) else round(v,2)) for k,v in r["kernels"].items()})
PY
