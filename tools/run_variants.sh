#!/bin/bash
# On the GPU box: `python bench.py` once per GEMM variant built by tools/ablate_gemm.sh (gpzoo_amd/libgpzoo_hip_<tag>.so),
# plus the shipped library ("base").   bash tools/run_variants.sh tag1 tag2 ...
mkdir -p gpurun_out/r2
for tag in base "$@"; do
  if [ $tag = base ]; then lib=""; else lib="$PWD/gpzoo_amd/libgpzoo_hip_$tag.so"; fi
  GPZ_HIP_LIB=$lib python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2/var_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
r = json.loads(open(f"gpurun_out/r2/var_{sys.argv[1]}.json").read())
print(sys.argv[1], round(r["ms_per_step"], 2), "ms  stage1", round(r["roofline"]["achieved"], 2), "TF  stage2",
      round(r["kernels"]["stage2_LuT_Wt"]["achieved_TFLOPs"], 2), "TF")
PY
done
