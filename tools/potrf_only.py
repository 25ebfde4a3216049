#!/usr/bin/env python3
"""Run the batched Cholesky alone (for rocprofv3 --kernel-trace --stats): L matrices of order M, fp64."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops  # noqa: E402

L, M = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
g = torch.Generator().manual_seed(0)
B = torch.randn(L, M, M, generator=g, dtype=torch.float64)
A = (B @ B.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64)).cuda()
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C = ops.cholesky(A)
    torch.cuda.synchronize()
    print(f"potrf L={L} M={M}: {1e3 * (time.perf_counter() - t0):.3f} ms (incl. copies)")
err = float(((C @ C.transpose(-1, -2)) - A).abs().max())
print("max |L L^T - A| =", err)
