#!/usr/bin/env python3
"""Does a small problem's product run below the config-3 rate because of the kernel or because of the clock?

    python3 tools/clock_effect.py [config=2]

Times the stage-1 / stage-2 launches of one evaluation (HIP events of the library's profiling slots) three ways:
  cold   evaluations back to back, as bench.py runs them (the factor path's latency-bound launches sit between the products)
  warm   ~40 ms of dense fp32 torch matmuls on the same stream right before every evaluation (the chip is at its
         sustained clock when the products start)
  burst  20 evaluations back to back after the matmuls, timed as one block
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops  # noqa: E402
from gpzoo_amd.configs import spec_for_config  # noqa: E402
from gpzoo_amd.synthetic import make_config  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = int(v)
dev = torch.device("cuda", 0)
dense_lu = kw.pop("dense_lu", 0)
zero_mu = kw.pop("zero_lu", 0)
c = make_config(cfg, **kw)
if dense_lu:      # a dense, O(1) lower factor instead of identity + 0.01 noise: does the stage-2 rate depend on the operand VALUES?
    c["Lu_raw"] = 0.5 * torch.randn(c["Lu_raw"].shape, generator=torch.Generator().manual_seed(7))
if zero_mu:
    c["Lu_raw"] = torch.zeros_like(c["Lu_raw"])
g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
spec, extra = spec_for_config(g, dev)
N, M = c["X"].shape[0], c["Z"].shape[0]
L = c["mu"].shape[0]
flops = L * float(M) * M * N


def step():
    return ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                            noise_sd=c["noise_sd"], want_Lu=False, **extra)["elbo"]


A = torch.randn(8192, 8192, device=dev)
B = torch.randn(8192, 8192, device=dev)


def heat(n=6):
    for _ in range(n):
        torch.mm(A, B)


def measure(label, pre, reps=10):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(reps):
        pre()
        e0.record()
        step()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    prof = ops.profile_read()
    ops.profile_enable(False)
    ms1, n1 = prof["stage1"]
    ms2, n2 = prof["stage2"]
    print("%-6s eval %.3f ms | stage1 %.3f ms = %.1f TF | stage2 %.3f ms = %.1f TF" % (
        label, tot / reps, ms1 / reps, flops * reps / ms1 / 1e9, ms2 / reps, flops * reps / ms2 / 1e9), flush=True)


print("config %d: N=%d M=%d L=%d" % (cfg, N, M, L))
measure("cold", lambda: None)
measure("warm", heat)
measure("cold", lambda: None)
heat(10)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record()
torch.cuda.synchronize()
print("burst  eval %.3f ms (20 back to back after the matmuls)" % (e0.elapsed_time(e1) / 20))
