for f in 0 1 2 3; do
  GPZ_DEBUG_FLAGS=$f GPZ_SUPER_COLS=16 python3 bench.py --N 49152 --steps 2 --warmup 1 --no-cpu-baseline --chunk 0 > gpurun_out/b_$f.log 2>&1
  python3 - <<PY
import json
r=json.loads(open("gpurun_out/b_$f.log").read().strip().splitlines()[-1])
print("flags=$f", round(r["ms_per_step"],2), round(r["roofline"]["achieved"],1), round(r["kernels"]["stage2_LuT_Wt"]["achieved_TFLOPs"],1), r["elbo"])
PY
done
