#!/bin/bash
# On the GPU box: time config 3 once per fused-kernel variant built by tools/ablate_fused.sh, plus the shipped library.
#   bash tools/run_fused_variants.sh tag1 tag2 ...      (env assignments may ride along as tag=base:VAR=VALUE)
for spec in base "$@"; do
  tag=${spec%%:*}; envs=""
  if [ "$spec" != "$tag" ]; then envs=${spec#*:}; fi
  if [ $tag = base ]; then lib=""; else lib="$PWD/gpzoo_amd/libgpzoo_hip_$tag.so"; fi
  echo "== $spec"
  env GPZ_HIP_LIB=$lib $envs python3 tools/fused_check.py --time-only 2>/dev/null | grep "config 3"
done
