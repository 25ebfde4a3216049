#!/usr/bin/env python3
"""fp32 vs fp64 evaluation of the same (fp32-representable) inputs at the benchmark size, 4 latents."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpzoo_amd.synthetic import make_config
from test_hip_scale import hip_eval
c32 = make_config(3, L=4)
c64 = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in c32.items()}
c64["dtype"] = torch.float64
a = hip_eval(c32, want_Lu=False); b = hip_eval(c64, want_Lu=False)
m, s = b["mean"], b["scale"]
print("elbo rel gap", abs(float(a["elbo"]) - float(b["elbo"])) / abs(float(b["elbo"])))
print("loglik rel gap", ((a["loglik"] - b["loglik"]).abs() / b["loglik"].abs()).tolist())
print("kl rel gap", ((a["kl"] - b["kl"]).abs() / b["kl"].abs()).tolist())
dm = (a["mean"].double() - m).abs(); ds = (a["scale"].double() - s).abs() / s
print("mean: max abs err %.3e (max |mean| %.3f), mean abs err %.3e" % (float(dm.max()), float(m.abs().max()), float(dm.mean())))
print("scale: max rel err %.3e, mean rel err %.3e, min scale %.4f" % (float(ds.max()), float(ds.mean()), float(s.min())))
v32, v64 = a["scale"].double() ** 2, s ** 2
print("var: mean signed err %.3e, mean var %.4f" % (float((v32 - v64).mean()), float(v64.mean())))
