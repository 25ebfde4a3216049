#!/usr/bin/env python3
"""Phase times of the panel kernel's workgroup 0 (csrc/gemmp.hip, gpz_debug_panel_stamps) on BASELINE configs[1]:
per wave and panel, shader-clock cycles between the stamps 0 top, 1 panel written, 2 barrier, 3 stage 1 done, 4 Wt panel
in LDS (two barriers), 5 stage 2 done, 6 statistics, 7 barrier."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import _lib, ops  # noqa: E402
from gpzoo_amd.configs import spec_for_config  # noqa: E402
from gpzoo_amd.synthetic import make_config  # noqa: E402


def main():
    c = make_config(2, dtype=torch.float32)
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g)
    lib = _lib.load()
    lib.gpz_debug_panel_stamps.argtypes = [C.c_void_p]
    lib.gpz_debug_panel_stamps.restype = None
    args = (spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True)
    for _ in range(3):
        ops.svgp_forward(*args, **extra)
    buf = torch.zeros(16 * 16 * 8, dtype=torch.int64, device="cuda")
    lib.gpz_debug_panel_stamps(buf.data_ptr())
    ops.svgp_forward(*args, **extra)
    torch.cuda.synchronize()
    lib.gpz_debug_panel_stamps(None)
    t = buf.cpu().view(16, 16, 8)
    names = ["panel write", "barrier", "stage 1", "stats+Wt->LDS", "stage 2", "stats", "barrier"]
    rb = [0, 1, 4, 6, 3, 2, 5, 7, 12, 13, 10, 8, 15, 14, 11, 9]       # 32-row blocks of the 16 waves (row_block_of in csrc/gemmp.hip)
    for it in (3,):
        base = int(t[it, :, 0].min())
        print("panel %d (length %d cycles from first top to last end)" % (it, int(t[it, :, 7].max()) - base))
        for w in range(16):
            d = [int(t[it, w, i + 1] - t[it, w, i]) for i in range(7)]
            print("  wave %d (row block %d): start +%5d | " % (w, rb[w], int(t[it, w, 0]) - base) +
                  "  ".join("%s %6d" % (n, x) for n, x in zip(names, d)))
    per = (t[12, 0, 0] - t[2, 0, 0]).item() / 10.0
    print("cycles per panel (wave 0 top to top, panels 2..12): %.0f" % per)


if __name__ == "__main__":
    main()
