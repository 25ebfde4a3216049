#!/bin/bash
# Stage-1 / stage-2 efficiency of the fused forward over problem shapes (config-3 kernel, fp32): looks for shapes whose
# tile schedule is unbalanced.   bash tools/shape_sweep.sh > gpurun_out/shape_sweep.txt
for shape in "7000 3000 20" "7000 2048 32" "100000 384 16" "20000 1000 10" "40000 1000 10" "100000 1024 16" "30000 4096 8" "12288 2048 32" "13000 2048 32" "5000 512 64" "200000 256 32" "3000 3000 4" "50000 1500 12"; do
  set -- $shape
  python3 bench.py --N $1 --M $2 --L $3 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']
pk = r['roofline']['kernel'].startswith('panel_kernel')
print((('N=%7d M=%5d L=%3d | %8.2f ms | both   %6.1f TF (one launch: %6.1f) |' if pk else 'N=%7d M=%5d L=%3d | %8.2f ms | stage1 %6.1f TF stage2 %6.1f TF |') + ' factor (potrf + trtri) %6.2f ms | sclk %.0f') % ($1,$2,$3,r['ms_per_step'],r['roofline']['achieved'],k['stage2_LuT_Wt']['achieved_TFLOPs'],k['potrf_ms_per_eval'] if 'factor' in k else k['potrf_ms_per_eval'] + k['trtri_ms_per_eval'],r['clocks']['sclk_MHz_mean'] if r.get('clocks') else 0))"
done
