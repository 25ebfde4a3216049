#!/bin/bash
# Diagnostic build of the diagonal-block kernel with s_memtime stamps at its phase boundaries, and their read-out:
#   bash tools/diag_stamps.sh build   (build container: objects live here)  ->  gpzoo_amd/libgpzoo_hip_stamps.so
#   bash tools/diag_stamps.sh         (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = "build" ]; then
  python3 -m gpzoo_amd.build > /dev/null
  C=gpzoo_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -DGPZ_DIAG_STAMPS $GPZ_DIAG_EXTRA -c $C/diag128.hip -o /tmp/diag128_st.o
  objs=$(ls $C/*.o | grep -v "/diag128.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpzoo_amd/libgpzoo_hip_stamps.so $objs /tmp/diag128_st.o -ldl
  exit 0
fi
GPZ_HIP_LIB=$PWD/gpzoo_amd/libgpzoo_hip_stamps.so python3 - <<'PY'
import ctypes as C, torch
from gpzoo_amd import _lib, ops
lib = _lib.load()
g = torch.Generator().manual_seed(0)
B = torch.randn(4, 256, 256, generator=g, dtype=torch.float64)
A = (B @ B.transpose(-1, -2) / 256 + torch.eye(256, dtype=torch.float64)).cuda()
for _ in range(3):
    ops.cholesky(A)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
lib.gpz_debug_diag_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
assert lib.gpz_debug_diag_stamps(out) == 0
t = list(out)
names = {0: "start", 1: "loaded", 18: "factor done", 19: "factor written", 20: "inv level 1", 21: "inv level 2", 22: "Dinv written"}
for s in range(4):
    names[2 + 4 * s] = f"factor32[{s}]"; names[3 + 4 * s] = f"invert32[{s}]"; names[4 + 4 * s] = f"sub-panel[{s}]"; names[5 + 4 * s] = f"in-block trailing[{s}]"
prev = t[0]
for i in sorted(names):
    if t[i] == 0 and i: continue
    print(f"{names[i]:24s} +{(t[i] - prev):8d} ticks   (total {(t[i] - t[0]):8d})")
    prev = t[i]
print("s_memtime ticks at 100 MHz -> 1 tick = 10 ns")
PY
