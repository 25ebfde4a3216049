#!/usr/bin/env python3
"""Time the Poisson NSF expected log-likelihood + gradients at Slide-seq size (D=17702 genes,
N_b=7000 spots per step, 20 factors, E=3): fused gpz_poisson_nsf vs the reference's torch formulation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops  # noqa: E402


def main():
    g = torch.Generator().manual_seed(1)
    D, N, Lt = int(os.environ.get("GPZ_PS_D", 17702)), 7000, 20
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda")
    mean = (0.3 * torch.randn(Lt, N, generator=g)).to(dev)
    scale = (0.2 + 0.3 * torch.rand(Lt, N, generator=g)).to(dev)
    eps = torch.randn(E, Lt, N, generator=g).to(dev)
    W = (torch.rand(D, Lt, generator=g) + 0.05).to(dev)
    V = (0.5 + torch.rand(N, generator=g)).to(dev)
    y = torch.poisson(2.0 * torch.rand(D, N, generator=g), generator=g).to(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=5):
        best = 1e9
        for _ in range(reps):
            ev0.record(); out = fn(); ev1.record(); torch.cuda.synchronize()
            best = min(best, ev0.elapsed_time(ev1))
        return best, out

    t_f, out = timed(lambda: ops.poisson_nsf(mean, scale, eps, W, V, y))

    def torch_path():
        lv = [t.clone().requires_grad_(True) for t in (mean, scale, W, V)]
        rate = lv[3] * torch.matmul(lv[2], torch.exp(lv[0] + lv[1] * eps))
        ll = torch.distributions.Poisson(rate).log_prob(y).mean(0).sum()
        ll.backward()
        return ll.detach(), lv
    if E > 4:        # the torch formulation holds the (E, D, N) rate and its autograd copies: E = 20 is 10 GB each
        print(f"fused  : {t_f:8.2f} ms at E = {E}  (loglik {float(out[0]):.3f})")
        return
    t_t, ref = timed(torch_path, reps=3)
    flops = 8.0 * Lt * E * D * N
    print(f"fused  : {t_f:8.2f} ms  ({flops / t_f / 1e9:.1f} TFLOP/s fp32 MFMA incl. the rate computed in both passes, "
          f"y read twice = {2 * D * N * 4 / t_f / 1e6:.0f} GB/s)")
    print(f"torch  : {t_t:8.2f} ms   loglik fused {float(out[0]):.3f} vs torch {float(ref[0]):.3f}")


if __name__ == "__main__":
    main()
