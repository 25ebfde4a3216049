for w in 16 24 32 48; do
GPZ_W_COLS=$w python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('W=$w', r['ms_per_step'], r['roofline']['achieved'], r['kernels']['stage2_LuT_Wt']['achieved_TFLOPs'])"
done
