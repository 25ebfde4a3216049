#!/usr/bin/env python3
"""Time VNNGP (NSF_benchmarks.ipynb shape: K=10 neighbours, M=1000 inducing points, Slide-seq-sized N)
through the gpzoo module: forward only, and forward + loss.backward() with every parameter trainable."""
import os
import sys
import time

import torch
import torch.nn as nn
from torch import distributions

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo.gp import VNNGP  # noqa: E402
from gpzoo.kernels import NSF_RBF  # noqa: E402


def main():
    torch.manual_seed(0)
    N, M, L, K = 40000, 1000, 10, 10
    dev = torch.device("cuda")
    X = (torch.rand(N, 2) * 200 - 100).to(dev)
    y = torch.randn(L, N, device=dev)
    gp = VNNGP(NSF_RBF(sigma=1.0, lengthscale=8.0, L=L), dim=2, M=M, K=K, jitter=1e-2)
    gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu())
    gp.mu = nn.Parameter(torch.zeros(L, M))
    gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    gp = gp.to(dev)

    def timed(fn, reps=6):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return 1e3 * min(ts[1:])

    def fwd():
        with torch.no_grad():
            gp(X)

    def step():
        gp.zero_grad()
        qF, qU, pU = gp(X)
        loss = -(distributions.Normal(qF.mean, 0.5).log_prob(y).sum() - (qF.scale ** 2).sum() / 0.5
                 - distributions.kl_divergence(qU, pU).sum())
        loss.backward()

    print(f"VNNGP N={N} M={M} L={L} K={K} f32: forward {timed(fwd):.2f} ms, forward+backward (all parameters) {timed(step):.2f} ms")


if __name__ == "__main__":
    main()
