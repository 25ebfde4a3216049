#!/usr/bin/env python3
"""Training-step latency at the small notebooks' sizes (mggp_test.ipynb: N=400, M=100, 1-D, 2 groups; SVGP.ipynb:
N=10000, M=500): forward + loss.backward() + Adam, every parameter trainable."""
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

dev = torch.device("cuda")
for (N, M, L, d) in ((400, 100, 1, 1), (10000, 500, 1, 1), (2000, 300, 8, 2)):
    torch.manual_seed(0)
    X = (torch.rand(N, d) * 20 - 10).to(dev)
    y = torch.randn(L, N).to(dev)
    for cls in (WSVGP, SVGP):
        def build():
            torch.manual_seed(1)
            gp = cls(NSF_RBF(sigma=1.0, lengthscale=2.0, L=L), dim=d, M=M, jitter=1e-2)
            gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu())
            gp.mu = nn.Parameter(torch.zeros(L, M))
            gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
            model = GaussianLikelihood(gp, noise=0.5).to(dev)
            return model, torch.optim.Adam(model.parameters(), lr=1e-3)
        model, opt = build()
        times = []
        for it in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.zero_grad()
            with deferred_info():        # as gpzoo.utilities.train* run a step: Kzz's info word is read once, behind the backward's launches
                loss = _elbo_terms(model, X, y, 1)
                loss.backward()
            opt.step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        times = sorted(times[5:])
        del loss, model, opt
        # the same step captured as a HIP graph (gpzoo.utilities.GraphedStep, a fresh model: no autograd state of an eager
        # step may be alive when the capture starts): one replay + one sync per step
        from gpzoo.utilities import GraphedStep
        model, opt = build()
        step = GraphedStep(lambda: _elbo_terms(model, X, y, 1), opt)
        gt = []
        for it in range(40):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step()
            step.check()
            gt.append(time.perf_counter() - t0)
        gt = sorted(gt[5:])
        print(f"N={N:6d} M={M:4d} L={L} {cls.__name__:6s}: step median {1e3 * times[len(times) // 2]:6.2f} ms, min {1e3 * times[0]:6.2f} ms"
              f" | as a HIP graph: median {1e3 * gt[len(gt) // 2]:6.2f} ms, min {1e3 * gt[0]:6.2f} ms")
