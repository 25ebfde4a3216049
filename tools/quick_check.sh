timeout -k 10 600 python -m pytest tests/test_hip_golden.py tests/test_hip_scale.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b.log 2>&1
python3 - <<PY
import json
r=json.loads(open("gpurun_out/b.log").read().strip().splitlines()[-1])
print(round(r["ms_per_step"],2), round(r["roofline"]["achieved"],1), round(r["kernels"]["stage2_LuT_Wt"]["achieved_TFLOPs"],1), r["elbo"], {k:(round(v["ms_per_eval"],2) if isinstance(v,dict) else round(v,2)) for k,v in r["kernels"].items()})
PY
