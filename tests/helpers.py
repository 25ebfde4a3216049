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


def spec_from_case(c, device):
    """KernelSpec for a golden case (mirrors what gpzoo_amd.kernels classes build)."""
    from gpzoo_amd import _lib
    from gpzoo_amd.ops import KernelSpec
    kind = c["kind"]
    sig, ell = c["sigma"].to(device), c["lengthscale"].to(device)
    batched = sig.dim() > 0
    if kind in ("rbf", "nsf_rbf", "batched_rbf"):
        return KernelSpec(_lib.KERNEL_RBF, sig.reshape(-1), ell.reshape(-1), batched)
    if kind == "matern32":
        return KernelSpec(_lib.KERNEL_MATERN32, sig.reshape(-1), ell.reshape(-1), batched)
    emb = c["embedding"].to(device)
    r2 = ((emb[:, None, :] - emb[None, :, :]) ** 2).sum(-1)
    a = c["group_diff"].to(device).reshape(-1)
    if kind == "mggp_rbf":
        ga, pw = a, c.get("input_dim", 2) / 2
    elif kind == "mggp_nsf_rbf":
        ga, pw = a ** 2, c.get("input_dim", 2) / 2
    else:
        ga, pw = a.abs(), c["X"].shape[1] / 2
    return KernelSpec(_lib.KERNEL_MGGP_RBF, sig.reshape(-1), ell.reshape(-1), batched, ga, r2, pw)


MULTIBLOCK = ("multiblock_wsvgp_matern32_f64", "multiblock_wsvgp_matern32_f32", "multiblock_svgp_nsf_rbf_f64",
              "multiblock_svgp_nsf_rbf_f32", "multiblock_mggp_wsvgp_mggp_nsf_rbf_f64", "multiblock_mggp_wsvgp_mggp_nsf_rbf_f32",
              # nine blocks (M = 1100, N = 1500, L = 2): an odd block count through every multi-level path
              "multiblock9_wsvgp_matern32_f64", "multiblock9_wsvgp_matern32_f32", "multiblock9_svgp_nsf_rbf_f64",
              "multiblock9_svgp_nsf_rbf_f32")


def load_multiblock(name):
    """A reference-generated fixture beyond one 128-block (M = 300, N = 2000, L = 3; M = 1100, N = 1500, L = 2): outputs and
    scalars come from the file, the inputs are regenerated from the stored seed exactly as tests/golden/make_golden.py drew them."""
    import sys
    sys.path.insert(0, GOLDEN)
    from inputs import make_inputs
    c = load_case(name)
    meta = {k: int(c.pop(k)) for k in ("seed", "N", "M", "d", "L")}
    span = float(c.pop("span"))
    n_groups = int(c.pop("n_groups")) if "n_groups" in c else 0
    dt = torch.float64 if name.endswith("f64") else torch.float32
    inp = make_inputs(meta["seed"], N=meta["N"], M=meta["M"], d=meta["d"], L=meta["L"], n_groups=n_groups, span=span)
    c.update({k: (v.to(dt) if v.is_floating_point() else v) for k, v in inp.items()})
    if c["kind"] in ("nsf_rbf", "mggp_nsf_rbf"):     # the NSF kernels keep (L,1,1) parameters
        c["sigma"], c["lengthscale"] = c["sigma"].reshape(-1, 1, 1), c["lengthscale"].reshape(-1, 1, 1)
    return c


def to_wire(obj):
    """Tensors -> numpy arrays (recursively through tuples / lists / dicts) before a multiprocessing queue: a torch CPU
    tensor is pickled as a file descriptor the SENDER has to serve, and a worker that has exited by the time the parent
    unpickles its result (it has nothing left to do after its last collective) makes the parent's q.get() raise EOFError
    now and then.  Arrays travel by value."""
    import torch
    if isinstance(obj, torch.Tensor):
        return ("__tensor__", obj.detach().cpu().numpy().copy())
    if isinstance(obj, tuple):
        return tuple(to_wire(o) for o in obj)
    if isinstance(obj, list):
        return [to_wire(o) for o in obj]
    if isinstance(obj, dict):
        return {k: to_wire(v) for k, v in obj.items()}
    return obj


def from_wire(obj):
    import torch
    if isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[0], str) and obj[0] == "__tensor__":
        return torch.from_numpy(obj[1])
    if isinstance(obj, tuple):
        return tuple(from_wire(o) for o in obj)
    if isinstance(obj, list):
        return [from_wire(o) for o in obj]
    if isinstance(obj, dict):
        return {k: from_wire(v) for k, v in obj.items()}
    return obj


def record(name: str, rec: dict, append: bool = False) -> None:
    """Write a measurement a test made to $GPZ_TEST_RECORD_DIR/<name> (nothing when the variable is unset: a test run
    leaves no files behind; tools/_run_all_gpu.sh sets it to gpurun_out/ when the numbers are wanted)."""
    import json
    d = os.environ.get("GPZ_TEST_RECORD_DIR")
    if not d:
        return
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, name), "a" if append else "w") as f:
            f.write(json.dumps(rec) + ("\n" if append else ""))
    except OSError:
        pass
