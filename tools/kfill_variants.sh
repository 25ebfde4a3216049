#!/bin/bash
# Timing variants of the covariance fill's block shape (csrc/kfill.hip) as gpzoo_amd/libgpzoo_hip_<tag>.so; select with
# GPZ_HIP_LIB=<path>.   Usage: tools/kfill_variants.sh tag:"-DGPZ_KF_TX=256 -DGPZ_KF_TY=1 -DGPZ_KF_ROWS=1" ...
set -e
cd "$(dirname "$0")/.."
python3 -m gpzoo_amd.build > /dev/null
C=gpzoo_amd/csrc
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $flags -c $C/kfill.hip -o /tmp/kfill_$tag.o
  objs=$(ls $C/*.o | grep -v "/kfill.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpzoo_amd/libgpzoo_hip_$tag.so $objs /tmp/kfill_$tag.o -ldl
  echo built gpzoo_amd/libgpzoo_hip_$tag.so
done
