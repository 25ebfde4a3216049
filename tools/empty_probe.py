#!/usr/bin/env python3
"""Empty X through the mirrored modules (scratch probe; the test is tests/test_hip_api.py::test_empty_inputs)."""
import sys
import torch, torch.nn as nn
sys.path.insert(0, '.')
from gpzoo.kernels import NSF_RBF, MGGP_NSF_RBF
from gpzoo.gp import WSVGP, SVGP
for cls in (WSVGP, SVGP):
    gp = cls(NSF_RBF(L=2), dim=2, M=5, jitter=1e-2)
    gp.mu = nn.Parameter(torch.randn(2, 5)); gp.Lu = nn.Parameter(0.1 * torch.randn(2, 5, 5) + torch.eye(5))
    gp = gp.cuda().double()
    X = torch.zeros(0, 2, dtype=torch.float64, device="cuda")
    try:
        out = gp(X)
        qF, qU, pU = out[0], out[1], out[2]
        print(cls.__name__, "ok", qF.mean.shape, qF.scale.shape)
        kl = torch.distributions.kl_divergence(qU, pU).sum() if cls is SVGP else None
        (qF.mean.sum() + (kl if kl is not None else 0 * gp.mu.sum())).backward()
        print("  grads", None if gp.mu.grad is None else gp.mu.grad.abs().sum().item())
    except Exception as e:
        import traceback; traceback.print_exc()
        print(cls.__name__, "raised", type(e).__name__, str(e)[:300])
k = NSF_RBF(L=2).cuda().double()
print(k(torch.zeros(0, 2, dtype=torch.float64, device="cuda"), torch.rand(4, 2, dtype=torch.float64, device="cuda")).shape)
print(k(torch.rand(4, 2, dtype=torch.float64, device="cuda"), torch.zeros(0, 2, dtype=torch.float64, device="cuda")).shape)
