#!/usr/bin/env python3
"""Un-whitened SVGP training step with EVERY parameter trainable (Z, sigma, lengthscale, mu, Lu) at the notebooks'
minibatch shape (N_b=7000, M=3000, L=20, fp32): forward + loss.backward()."""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo.gp import SVGP, WSVGP  # noqa: E402
from gpzoo.kernels import NSF_RBF  # noqa: E402
from gpzoo.likelihoods import GaussianLikelihood  # noqa: E402
from gpzoo.utilities import _elbo_terms  # noqa: E402
from gpzoo_amd.ops import deferred_info  # noqa: E402

torch.manual_seed(0)
N, Nb, M, L = 40000, 7000, 3000, 20
dev = torch.device("cuda")
X = (torch.rand(N, 2) * 200 - 100).to(dev)
y = torch.randn(L, N).to(dev)
for cls in (WSVGP, SVGP):
    gp = cls(NSF_RBF(sigma=1.0, lengthscale=8.0, L=L), dim=2, M=M, jitter=1e-1)
    gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu())
    gp.mu = nn.Parameter(torch.zeros(L, M))
    gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    model = GaussianLikelihood(gp, noise=0.5).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    times = []
    for it in range(5):
        idx = torch.randperm(N, device=dev)[:Nb]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        with deferred_info():        # as gpzoo.utilities.train* run a step: Kzz's info word is read once, behind the backward's launches
            loss = _elbo_terms(model, X[idx], y[:, idx], 1)
            loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    print(f"{cls.__name__:6s} all parameters trainable: step = {1e3 * min(times[1:]):7.2f} ms  (loss {loss.item():.1f})")
