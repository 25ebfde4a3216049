#!/usr/bin/env python3
"""Forward with Wt retained for the backward pass (the training path): panel kernel (stores Wt from its accumulators)
against fill + tile products, at shapes where the library prefers the panel kernel."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops  # noqa: E402
from gpzoo_amd.configs import spec_for_config  # noqa: E402
from gpzoo_amd.synthetic import make_config  # noqa: E402


def main():
    for (N, M, L) in ((200000, 256, 32), (100000, 384, 16), (7000, 300, 20), (50000, 512, 8)):
        c = make_config(3, N=N, M=M, L=L)
        g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
        spec, extra = spec_for_config(g, torch.device("cuda", 0))
        for name, kw in (("tiles", dict(materialize_kzx=True)), ("panel", dict(panel_products=True))):
            for retain in (0.0, 0.9):
                ts = []
                for it in range(8):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                                           noise_sd=c["noise_sd"], want_Lu=False, retain_wt=retain, **extra, **kw)
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                    del out
                print("N=%6d M=%4d L=%3d %s retain=%.1f: %.3f ms" % (N, M, L, name, retain, 1e3 * min(ts[2:])))


if __name__ == "__main__":
    main()
