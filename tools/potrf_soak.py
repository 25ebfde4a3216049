#!/usr/bin/env python3
"""Repeat the batched Cholesky on fresh random SPD matrices of several orders and compare every result with LAPACK:
a stress test for the wave-to-wave hand-off inside diag128_kernel (factor32 -> invert32_follow) and the blocked driver."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(123)
worst = 0.0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for rep in range(reps):
    for M, batch in ((97, 7), (128, 33), (300, 20), (640, 9), (1024, 5), (2048, 3)):
        B = torch.randn(batch, M, M + 4, generator=g, dtype=torch.float64)
        A = B @ B.transpose(-1, -2) / M + (0.05 + rep % 3) * torch.eye(M, dtype=torch.float64)
        ref = torch.linalg.cholesky(A)
        got = ops.cholesky(A.cuda()).cpu()
        err = float((got - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err < 1e-9, (rep, M, batch, err)
        # the triangular solve uses the inverse-only mode of the same kernel
        R = torch.randn(batch, M, 5, generator=g, dtype=torch.float64)
        X = ops.solve_triangular_lower(got.cuda(), R.cuda()).cpu()
        res = float((ref @ X - R).abs().max())
        assert res < 1e-8, (rep, M, batch, res)
print(f"potrf soak: {reps} repetitions x 6 shapes ok, worst relative deviation from LAPACK {worst:.2e}")
