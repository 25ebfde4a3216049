#!/bin/bash
# Build timing variants of the GEMM (csrc/gemm.hip) as gpzoo_amd/libgpzoo_hip_<tag>.so; select one with
# GPZ_HIP_LIB=<path> python bench.py ...   Usage: tools/ablate_gemm.sh tag1:"-DGPZ_SCHED=3" tag2:"-DGPZ_ABL=3" ...
# -DGPZ_ABL=<bits> builds (1 no global loads, 2 no LDS staging, 4 no per-tile barrier) give WRONG results by
# construction: they only tell what the MFMA pipes wait for.
set -e
cd "$(dirname "$0")/.."
python3 -m gpzoo_amd.build > /dev/null
C=gpzoo_amd/csrc
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $flags -c $C/gemm.hip -o /tmp/gemm_$tag.o
  objs=$(ls $C/*.o | grep -v "/gemm.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpzoo_amd/libgpzoo_hip_$tag.so $objs /tmp/gemm_$tag.o -ldl
  echo built gpzoo_amd/libgpzoo_hip_$tag.so
done
