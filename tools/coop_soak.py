#!/usr/bin/env python3
"""Stress test of the hand-offs inside the one-launch factorisation (csrc/coop.hip): fresh random SPD batches, every
launch under uneven load (a second stream keeps part of the chip busy with matmuls of varying size) and checked in
full: the factor against its matrix, the inverse against the factor, zeros above the diagonal.

    python tools/coop_soak.py [repetitions]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import _lib  # noqa: E402

lib = _lib.load()
lib.gpz_debug_factor_sync_words.restype = C.c_size_t
lib.gpz_debug_factor_sync_words.argtypes = [C.c_int64, C.c_int64]
lib.gpz_debug_factor_invert.restype = C.c_int
lib.gpz_debug_factor_invert.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator(device="cuda").manual_seed(7)
side = torch.cuda.Stream()
noise = [torch.randn(n, n, device="cuda") for n in (1024, 2048, 4096)]
worst_f = worst_i = 0.0
launches = 0
for rep in range(reps):
    for M, L in ((512, 8), (1024, 40), (2048, 32), (3072, 20), (2048, 3), (256, 300)):
        B = torch.randn(L, M, M + 8, generator=g, dtype=torch.float64, device="cuda")
        A = B @ B.transpose(-1, -2) / M + (0.05 + rep % 3) * torch.eye(M, dtype=torch.float64, device="cuda")
        nblk = M // 128
        Dinv = torch.empty(L * nblk * 128 * 128, dtype=torch.float64, device="cuda")
        Linv = torch.full((L, M, M), float("nan"), dtype=torch.float64, device="cuda")
        T = torch.empty(L, M, M, dtype=torch.float64, device="cuda")
        sync = torch.empty(lib.gpz_debug_factor_sync_words(M, L), dtype=torch.int32, device="cuda")
        info = torch.empty(L, dtype=torch.int32, device="cuda")
        W = A.clone()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):                       # uneven load beside the launch
            for k in range(1 + (rep + M // 256) % 4):
                n = noise[(rep + k) % 3]
                n @ n
        rc = lib.gpz_debug_factor_invert(W.data_ptr(), M, L, Dinv.data_ptr(), Linv.data_ptr(), T.data_ptr(), sync.data_ptr(),
                                         info.data_ptr(), None)
        torch.cuda.synchronize()
        assert rc == 0 and not bool(info.any()), (rep, M, L, rc, info)
        Lc = torch.tril(W)
        ef = float((Lc @ Lc.transpose(-1, -2) - A).abs().max() / A.abs().max())
        ei = float((Linv @ Lc - torch.eye(M, dtype=torch.float64, device="cuda")).abs().max())
        up = float(torch.triu(Linv, 1).abs().max())
        assert ef < 1e-13 and ei < 1e-11 and up == 0.0 and bool(torch.isfinite(Linv).all()), (rep, M, L, ef, ei, up)
        worst_f, worst_i = max(worst_f, ef), max(worst_i, ei)
        launches += 1
print(f"coop soak: {launches} launches ok; worst |L L^T - A| / |A| = {worst_f:.2e}, worst |Linv L - I| = {worst_i:.2e}")
