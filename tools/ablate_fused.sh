#!/bin/bash
# Build timing variants of the wide-tile kernels (csrc/gemmw.hip; SRC=gemmp: the panel kernel, -DGPZ_P_ABL=...) as
# gpzoo_amd/libgpzoo_hip_<tag>.so; select one with GPZ_HIP_LIB=<path>.
# Usage: [SRC=gemmp] tools/ablate_fused.sh tag1:"-DGPZ_W_ABL=1" tag2:"-DGPZ_W_ABL=7" ...
# -DGPZ_W_ABL builds give WRONG results by construction: they only tell what the MFMA pipes wait for.
set -e
cd "$(dirname "$0")/.."
python3 -m gpzoo_amd.build > /dev/null
C=gpzoo_amd/csrc
SRC=${SRC:-gemmw}
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $flags -c $C/$SRC.hip -o /tmp/${SRC}_$tag.o
  objs=$(ls $C/*.o | grep -v "/$SRC.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpzoo_amd/libgpzoo_hip_$tag.so $objs /tmp/${SRC}_$tag.o -ldl
  echo built gpzoo_amd/libgpzoo_hip_$tag.so
done
