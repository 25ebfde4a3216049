"""Shared helpers for the parity tests (fixture loading, oracle glue)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        if v.dtype.kind in "US":
            out[k] = str(v)
        elif v.dtype == np.bool_ and v.ndim == 0:
            out[k] = bool(v)
        elif v.ndim == 0 and k in ("jitter", "noise_sd", "elbo"):
            out[k] = float(v)
        elif v.ndim == 0 and k == "input_dim":
            out[k] = int(v)
        else:
            out[k] = torch.from_numpy(v)
    return out


def oracle_kwargs(c):
    kw = {}
    if "gX" in c:
        kw = dict(gX=c["gX"], gZ=c["gZ"], embedding=c["embedding"], group_diff=c["group_diff"],
                  input_dim=c.get("input_dim", 2))
    return kw


def rtol_for(dtype):
    """north_star tolerances: 1e-5 rtol fp64, 1e-3 rtol fp32."""
    return 1e-5 if dtype == torch.float64 else 1e-3
