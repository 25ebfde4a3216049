#!/usr/bin/env python3
"""Training step through WSVGP.forward_precomputed (the Slide-seq notebooks precompute W once for frozen Z / kernel and
index a minibatch of it per step): N_b=7000, M=3000, L=20, fp32; forward + loss.backward() + Adam."""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo.gp import WSVGP  # noqa: E402
from gpzoo.kernels import NSF_RBF  # noqa: E402

torch.manual_seed(0)
Nb, M, L = 7000, 3000, 20
dev = torch.device("cuda")
gp = WSVGP(NSF_RBF(sigma=1.0, lengthscale=8.0, L=L), dim=2, M=M, jitter=1e-1)
gp.mu = nn.Parameter(torch.zeros(L, M))
gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
for t in gp.kernel.parameters():
    t.requires_grad_(False)
gp = gp.to(dev)
W = (torch.randn(L, Nb, M, device=dev) / M ** 0.5)
y = torch.randn(L, Nb, device=dev)
opt = torch.optim.Adam([gp.mu, gp.Lu], lr=1e-2)
times = []
for it in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    qF, qU, _ = gp.forward_precomputed(W)
    loss = (((y - qF.mean) ** 2 + qF.scale ** 2).sum() / 0.5) + 0.5 * (gp.mu ** 2).sum()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
print(f"forward_precomputed step = {1e3 * min(times[1:]):7.2f} ms  (loss {loss.item():.1f})")
