#!/usr/bin/env python3
"""The one workload the reference publishes wall-clock numbers for (NSF_benchmarks.ipynb: BASELINE.md section 1), timed on
this framework: NSF2(SVGP(NSF_RBF)) on an S1-shaped synthetic count matrix -- N=1037 spots x D=80 genes, L=4 factors,
E=20 samples, jitter 1e-1, fp32, all N per step, Adam lr 5e-3 on lengthscale, Z, mu, Lu, W (sigma and V frozen), full
training steps (forward + backward + Adam) for M in {100, 250, 500, 1000} inducing points.

    python tools/nsf_benchmark_step.py [steps]

Two loops per M: the notebook's own `train` cell verbatim (NSF_benchmarks.ipynb cell 16: `y log(rate) - rate` over the
materialised (E,D,N) rate, `loss.item()` every step, a host copy of q(F) every 10th) and `gpzoo.utilities.train`
(the same objective through the fused Poisson kernel, losses kept on the device).  The reference's figures -- 128 / 85 /
50 / 22 steps/s on an unnamed NVIDIA GPU, 46 / 25 / 12.6 / 2.7 on a ~30-thread CPU -- are printed beside ours as CONTEXT:
other hardware, real S1 data there, synthetic counts of the same shape here (there is no network for datasets)."""
import os
import sys
import time

import torch
import torch.nn as nn
from torch import distributions, optim

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo.gp import SVGP  # noqa: E402
from gpzoo.kernels import NSF_RBF  # noqa: E402
from gpzoo.likelihoods import NSF2  # noqa: E402
from gpzoo import utilities as U  # noqa: E402

REF_GPU = {100: 10000 / 78.0, 250: 10000 / 118.0, 500: 10000 / 201.0, 1000: 10000 / 454.0}    # NSF_benchmarks.ipynb:386-486
REF_CPU = {100: 10000 / 215.0, 250: 10000 / 396.0, 500: 10000 / 796.0, 1000: 10000 / 3726.0}  # NSF_benchmarks_cpu.ipynb:372-472

dev = torch.device("cuda")
N, D, L, E = 1037, 80, 4, 20
g = torch.Generator().manual_seed(1037)
X = (torch.rand(N, 2, generator=g) * 4 - 2)                     # rescale_spatial_coords' range (utilities.py:14-23)
Y = torch.poisson(3.0 * torch.rand(D, N, generator=g), generator=g)


def build_model(M):
    """NSF_benchmarks.ipynb cell 9"""
    idx = torch.multinomial(torch.ones(N), num_samples=M, replacement=False, generator=g)
    kernel = NSF_RBF(L=L, sigma=1.0, lengthscale=1.0)
    gp = SVGP(kernel, M=M, jitter=1e-1)
    gp.mu = nn.Parameter(torch.randn((L, M), generator=g))
    gp.Lu = nn.Parameter(torch.eye(M).expand(L, M, M).clone())
    gp.Z = nn.Parameter(X[idx].clone(), requires_grad=True)
    model = NSF2(gp, Y, L=L).to(dev)
    model.prior.kernel.lengthscale.requires_grad = True         # cell 13
    model.prior.kernel.sigma.requires_grad = False
    model.V.requires_grad = False
    return model


def notebook_train(model, optimizer, X, y, steps, E):
    """NSF_benchmarks.ipynb cell 16, verbatim but for tqdm"""
    losses, means, scales = [], [], []
    for it in range(steps):
        optimizer.zero_grad()
        pY, qF, qU, pU = model.forward(X=X, E=E)
        logpY = y * torch.log(pY.rate) - pY.rate
        ELBO = (logpY).mean(axis=0).sum()
        ELBO -= torch.sum(distributions.kl_divergence(qU, pU))
        loss = -ELBO
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
        if (it % 10) == 0:
            means.append(torch.exp(qF.mean.detach().cpu()).numpy())
            scales.append(qF.scale.detach().cpu().numpy())
    return losses


def timed(fn, steps):
    fn(10)                                   # warm-up: allocator, workspaces, library load
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn(steps)
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0), out


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
Xd, Yd = X.to(dev), Y.to(dev)
print(f"NSF2(SVGP(NSF_RBF)) N={N} D={D} L={L} E={E} fp32, {steps} full training steps per cell (steps/s; higher is better)")
print(f"{'M':>5s} {'notebook loop':>14s} {'utilities.train':>16s} {'(no per-step sync)':>19s} {'(HIP graph)':>12s} | {'reference GPU':>13s} {'reference CPU':>13s}   first -> last loss")
for M in (100, 250, 500, 1000):
    rates = []
    for loop in ("notebook", "train", "train_nosync", "train_graph"):
        torch.manual_seed(M)
        model = build_model(M)
        opt = optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=5e-3)
        if loop == "notebook":
            r, losses = timed(lambda k: notebook_train(model, opt, Xd, Yd, k, E), steps)
        elif loop == "train":
            r, losses = timed(lambda k: U.train(model, opt, Xd, Yd, dev, steps=k, E=E), steps)
        elif loop == "train_nosync":
            r, losses = timed(lambda k: U.train(model, opt, Xd, Yd, dev, steps=k, E=E, sync_losses=False), steps)
        else:       # one capture, then replays: GraphedStep by hand so that the warm-up call does not capture a second graph
            step = U.GraphedStep(lambda: (lambda r4: -(r4[0] - U._kl_u(r4[2], r4[3])))(model.expected_loglik(Xd, Yd, E=E)), opt)

            def replay(k):
                out = []
                for _ in range(k):
                    step()
                    out.append(step.check())
                return out
            r, losses = timed(replay, steps)
        rates.append(r)
        last = losses
    first, final = float(last[0]), float(last[-1])
    print(f"{M:5d} {rates[0]:14.1f} {rates[1]:16.1f} {rates[2]:19.1f} {rates[3]:12.1f} | {REF_GPU[M]:13.1f} {REF_CPU[M]:13.1f}   {first:.1f} -> {final:.1f}")
