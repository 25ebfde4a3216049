#!/usr/bin/env python3
"""fp32 backward forms against the fp64 evaluation of the same problem: which is closer to the truth?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops
from gpzoo_amd.configs import spec_for_config
from gpzoo_amd.synthetic import make_config

for (N, M, L, wh) in ((3000, 1024, 1, False), (3000, 1024, 1, True), (5000, 640, 2, False), (7000, 3000, 4, False)):
    c = make_config(3, N=N, M=M, L=L)
    res = {}
    gen = torch.Generator().manual_seed(5)
    gm0 = torch.randn(L, N, generator=gen)
    gs0 = torch.randn(L, N, generator=gen)
    for dt in (torch.float64, torch.float32):
        g = {k: (v.to(dt).cuda() if isinstance(v, torch.Tensor) and v.is_floating_point() else (v.cuda() if isinstance(v, torch.Tensor) else v)) for k, v in c.items()}
        spec, extra = spec_for_config(g, torch.device("cuda", 0))
        out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], wh, want_Lu=False, **extra)
        forms = (("f64", dict()),) if dt == torch.float64 else (("classic-narrow", dict(narrow_tiles=True, form="classic")), ("classic", dict(form="classic")), ("algebra", dict(form="algebra")))
        for name, kw in forms:
            r = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], wh, gm0.to(dt).cuda(), gs0.to(dt).cuda(), out["scale"], kernel_grads=True, **kw, **extra)
            res[name] = [t.double() for t in r]
    ref = res["f64"]
    print(f"N={N} M={M} L={L} whitened={wh}: max |err| / max |ref| of (mu, Lu, theta, Z)")
    for name in ("classic-narrow", "classic", "algebra"):
        print(f"  {name:15s}", "  ".join(f"{float((a - b).abs().max() / b.abs().max()):.2e}" for a, b in zip(res[name], ref)))
