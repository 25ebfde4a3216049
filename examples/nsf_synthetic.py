#!/usr/bin/env python3
"""Non-negative spatial factorisation on synthetic Slide-seq-shaped counts, written the way the reference's
notebooks are (Slideseq_NSF_newest_version.ipynb): NSF_RBF kernel, SVGP prior with inducing points drawn
from the spots, NSF2 likelihood, kernel hyper-parameters frozen, `train_batched` over mini-batches of spots.

    PYTHONPATH=. python examples/nsf_synthetic.py [--spots 40000 --genes 2000 --factors 10 --inducing 2000 --steps 200]
"""
import argparse
import time

import torch
import torch.nn as nn

from gpzoo.gp import SVGP
from gpzoo.kernels import NSF_RBF
from gpzoo.likelihoods import NSF2
from gpzoo.utilities import train_batched


def synthetic_counts(N, D, L, gen):
    """Smooth non-negative spatial factors on a 2-D tissue, gene loadings and Poisson counts."""
    X = (torch.rand(N, 2, generator=gen) - 0.5) * 200.0
    centres = (torch.rand(L, 2, generator=gen) - 0.5) * 160.0
    F = torch.exp(-((X[None] - centres[:, None]) ** 2).sum(-1) / (2 * 35.0 ** 2))          # (L,N)
    W = torch.rand(D, L, generator=gen) ** 3 * 4.0
    Y = torch.poisson(W @ F + 0.05, generator=gen)
    return X, Y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spots", type=int, default=40000)
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--factors", type=int, default=10)
    ap.add_argument("--inducing", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=7000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--samples", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda")
    gen = torch.Generator().manual_seed(0)
    X, Y = synthetic_counts(a.spots, a.genes, a.factors, gen)
    L, M = a.factors, a.inducing

    kernel = NSF_RBF(sigma=1.0, lengthscale=20.0, L=L)
    gp = SVGP(kernel, dim=2, M=M, jitter=1e-1)
    gp.Z = nn.Parameter(X[torch.randperm(a.spots, generator=gen)[:M]].clone(), requires_grad=False)
    gp.mu = nn.Parameter(torch.zeros(L, M))
    gp.Lu = nn.Parameter(1e-2 * torch.randn(L, M, M, generator=gen))
    kernel.sigma.requires_grad_(False); kernel.lengthscale.requires_grad_(False)   # as in the notebooks
    model = NSF2(gp, Y, L=L).to(dev)
    X, Y = X.to(dev), Y.to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)

    train_batched(model, opt, X, Y, dev, steps=3, E=a.samples, batch_size=a.batch)          # warm-up (workspaces, caches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = train_batched(model, opt, X, Y, dev, steps=a.steps, E=a.samples, batch_size=a.batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = max(a.steps // 10, 1)
    print(f"N={a.spots} spots, D={a.genes} genes, L={L} factors, M={M} inducing points, batch {a.batch}, E={a.samples}")
    print("loss (mean of first / last %d steps): %.1f -> %.1f" % (k, sum(losses[:k]) / k, sum(losses[-k:]) / k))
    print("%.1f ms per step (forward + backward + Adam), %.1f s for %d steps" % (1e3 * dt / a.steps, dt, a.steps))
    assert sum(losses[-k:]) < sum(losses[:k])


if __name__ == "__main__":
    main()
