"""Turns a ``synthetic.make_config`` dict into gpzoo objects / a KernelSpec."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import gp, kernels


def kernel_for_config(c: dict) -> nn.Module:
    kind = c["kind"]
    dt = c["X"].dtype
    if kind == "rbf":
        k = kernels.RBF()
        k.sigma = nn.Parameter(c["sigma"].clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].clone())
    elif kind == "nsf_rbf":
        L = c["sigma"].numel()
        k = kernels.NSF_RBF(L=L)
        k.sigma = nn.Parameter(c["sigma"].reshape(L, 1, 1).clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].reshape(L, 1, 1).clone())
    elif kind == "matern32":
        k = kernels.batched_Matern32()
        k.sigma = nn.Parameter(c["sigma"].clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].clone())
    elif kind == "mggp_nsf_rbf":
        L = c["sigma"].numel()
        k = kernels.MGGP_NSF_RBF(n_groups=c["n_groups"], L=L)
        k.sigma = nn.Parameter(c["sigma"].reshape(L, 1, 1).clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].reshape(L, 1, 1).clone())
        k.group_diff_param = nn.Parameter(c["group_diff"].reshape(L, 1, 1).clone())
        k.embedding = nn.Parameter(k.embedding.to(dt), requires_grad=False)
    else:
        raise ValueError(kind)
    return k.to(c["X"].device)


def model_for_config(c: dict) -> nn.Module:
    """The gpzoo GP module of a configuration, parameters taken from the dict."""
    k = kernel_for_config(c)
    M, d = c["Z"].shape
    mggp = "gX" in c
    if mggp:
        m = (gp.MGGP_WSVGP if c["whitened"] else gp.MGGP_SVGP)(k, dim=d, M=M, jitter=c["jitter"], n_groups=c["n_groups"])
        m.groupsZ = nn.Parameter(c["gZ"].clone(), requires_grad=False)
    else:
        m = (gp.WSVGP if c["whitened"] else gp.SVGP)(k, dim=d, M=M, jitter=c["jitter"])
    m.Z = nn.Parameter(c["Z"].clone())
    m.mu = nn.Parameter(c["mu"].clone())
    m.Lu = nn.Parameter(c["Lu_raw"].clone())
    return m.to(c["X"].device)


def spec_for_config(c: dict, device=None):
    """(KernelSpec, extra kwargs for ops.svgp_forward) without building the nn.Module."""
    k = kernel_for_config(c)
    nlat = 1 if c["mu"].dim() == 1 else c["mu"].shape[0]
    spec = kernels.kernel_spec(k, c["X"], nlat)
    extra = dict(gX=c["gX"], gZ=c["gZ"]) if "gX" in c else {}
    return spec, extra
