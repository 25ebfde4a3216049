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
    floors(D, N, Lt, E, t_f)


# What bounds the two passes is ISSUE time, not the matrix pipes alone: on gfx950 a vector instruction does not issue in an
# MFMA's shadow, and the element-wise block between the products -- log, reciprocal and the Poisson terms
# per (gene, spot, sample) -- is most of the instructions.  Instructions per 16-gene x 64-spot x 1-sample tile and wave, from
# rocprofv3 PMC passes of this very command (profiles/r05/pmc_poisson_E3.json: SQ_INSTS_MFMA, SQ_INSTS_VALU (includes the
# MFMAs), SQ_INSTS_VALU_TRANS_F32; 20 factors = KS 5):
#   pass A (spot_mfma_kernel): 52 MFMA, 225 other vector instructions of which 32 transcendental
#   pass B (gene_mfma_kernel): 36 MFMA, 157 other vector instructions of which 16 transcendental
# Issue cycles per wave: 32 per v_mfma_f32_16x16x4_f32 (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = 32.0), 4 per plain and 8 per
# transcendental vector instruction (wave64 on a 16-lane SIMD; DESIGN section 5 measured 5.6 on average for such a mix).
TILE_INSTRS = {"A": (52, 225, 32), "B": (36, 157, 16)}
SIMDS, CLOCK_HZ = 1024, 2.4e9


def floors(D, N, Lt, E, t_ms):
    if Lt != 20:
        return
    tiles = E * ((D + 15) // 16) * ((N + 63) // 64)          # wave tiles per pass
    mfma = sum(m for m, _, _ in TILE_INSTRS.values()) * 32.0 * tiles / SIMDS / CLOCK_HZ * 1e3
    issue = sum(32.0 * m + 4.0 * (v - t) + 8.0 * t for m, v, t in TILE_INSTRS.values()) * tiles / SIMDS / CLOCK_HZ * 1e3
    print(f"floors : MFMA pipes alone {mfma:.2f} ms; MFMA + vector issue {issue:.2f} ms for the two passes at {CLOCK_HZ / 1e9:.1f} GHz "
          f"(+ 0.15 ms of exp / lgamma / finish kernels): the step runs at {(issue + 0.15) / t_ms:.2f} of its issue floor, "
          f"{mfma / t_ms:.2f} of the MFMA-only one")


if __name__ == "__main__":
    main()
