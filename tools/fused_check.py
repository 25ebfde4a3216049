#!/usr/bin/env python3
"""The three forward kernel paths (csrc/gemmw.hip wide tiles from memory, generated operand; csrc/gemm.hip narrow tiles) on the GPU box.

    python3 tools/fused_check.py            # parity over a set of shapes, then timing at config 3

Parity: with retain_wt the forward hands back Wt of every chunk; both paths must agree BIT FOR BIT (same
covariance arithmetic from cov.h, same k order, same MFMA chain), and mean / scale / ELBO to rounding (the column
statistics are summed in a different order).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gpzoo_amd import ops  # noqa: E402
from gpzoo_amd.configs import spec_for_config  # noqa: E402
from gpzoo_amd.synthetic import make_config  # noqa: E402

dev = torch.device("cuda", 0)


def run(c, g, spec, extra, materialize, retain=True, chunk=0, narrow=False):
    return ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                            noise_sd=c["noise_sd"], chunk=chunk, want_Lu=False, retain_wt=0.9 if retain else 0.0,
                            materialize_kzx=materialize, narrow_tiles=narrow, **extra)


def case(cfg, N, M, L, d=None, tag=""):
    c = make_config(cfg, N=N, M=M, L=L)
    if d == 1:
        c["X"] = c["X"][:, :1].contiguous()
        c["Z"] = c["Z"][:, :1].contiguous()
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, dev)
    a = run(c, g, spec, extra, True, narrow=True)      # kfill + 128 x 128-tile GEMMs: the round-2 path
    Mp = (M + 127) // 128 * 128
    ncp = (N + 127) // 128 * 128
    nwt = L * Mp * ncp
    wa = a["wt_cache"].view(torch.float32)[:nwt]
    ok = True
    for name, mat in (("wide+kfill", True), ("generated", False)):
        b = run(c, g, spec, extra, mat)
        wb = b["wt_cache"].view(torch.float32)[:nwt]
        same = bool(torch.equal(wa.view(torch.int32), wb.view(torch.int32)))
        nbad = int((wa.view(torch.int32) != wb.view(torch.int32)).sum())
        dm = float((a["mean"] - b["mean"]).abs().max())
        ds = float(((a["scale"] - b["scale"]).abs() / a["scale"]).max())
        de = abs(float(a["elbo"]) - float(b["elbo"])) / abs(float(a["elbo"]))
        good = same and dm < 1e-4 and ds < 1e-4 and de < 1e-6
        ok = ok and good
        print("%-26s %-10s cfg %d N=%6d M=%5d L=%3d d=%s  Wt bitwise %s (%d differ)  |dmean| %.2e  rel dscale %.2e  rel dELBO %.2e  %s"
              % (tag, name, cfg, N, M, L, d or 2, same, nbad, dm, ds, de, "OK" if good else "FAIL"), flush=True)
        del b, wb
    return ok


def timing(steps=3):
    c = make_config(3)
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, dev)
    modes = (("generated", False, False),) if "--time-only" in sys.argv else \
        (("narrow+kfill", True, True), ("wide+kfill", True, False), ("generated", False, False))
    for name, mat, narrow in modes:
        run(c, g, spec, extra, mat, retain=False, narrow=narrow)
        torch.cuda.synchronize()
        ops.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            o = run(c, g, spec, extra, mat, retain=False, narrow=narrow)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        prof = ops.profile_read()
        ops.profile_enable(False)
        fl = 32 * 2048.0 * 2048 * 200000
        ms1, ms2, msk = prof["stage1"][0] / steps, prof["stage2"][0] / steps, prof["kfill"][0] / steps
        print("config 3 %-12s %.1f ms/eval | stage1 %.2f ms = %.1f TF (%.3f of 157.3) | stage2 %.2f ms = %.1f TF | kfill %.2f ms | elbo %.10g"
              % (name, dt * 1e3, ms1, fl / ms1 / 1e9, fl / ms1 / 1e9 / 157.3, ms2,
                 fl / ms2 / 1e9, msk, float(o["elbo"])), flush=True)


if __name__ == "__main__":
    ok = True
    if "--time-only" not in sys.argv:
        ok &= case(3, 12288, 2048, 4, tag="config-3 chunk")
        ok &= case(3, 5000, 2048, 2, tag="ragged columns")
        ok &= case(3, 3000, 3000, 2, tag="M=3000 (Mp=3072)")
        ok &= case(3, 2000, 384, 3, tag="odd block count (Mp=384)")
        ok &= case(3, 1500, 640, 2, tag="Mp=640 (5 blocks)")
        ok &= case(3, 1500, 900, 2, tag="Mp=1024, padded rows")
        ok &= case(3, 777, 100, 2, tag="one block (Mp=128)")
        ok &= case(3, 1000, 250, 2, tag="Mp=256, padded rows")
        ok &= case(2, 50000, 512, 8, tag="config 2 (RBF)")
        ok &= case(2, 4000, 640, 3, d=1, tag="RBF d=1")
        ok &= case(3, 4000, 640, 3, d=1, tag="Matern d=1")
        ok &= case(3, 20000, 1024, 1, tag="L=1")
    if "--no-time" not in sys.argv:
        timing()
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)
