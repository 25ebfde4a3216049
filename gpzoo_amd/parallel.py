"""Multi-GPU evaluation: latent GPs shard across ranks, one scalar all-reduce.

Every array on the path has a leading latent axis with no cross-latent
arithmetic, so rank r owns a contiguous block of latents
(``synthetic.shard_latents``), evaluates its partial ELBO with the fused HIP pass
and the only exchange is a sum of one fp64 scalar -- ``torch.distributed``
all-reduce, which is RCCL over xGMI with the ``nccl`` backend (``gloo`` in the
CPU tests).  X, Z and group ids are replicated (a few MB).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist

from .synthetic import shard_latents


def _slice(t: Optional[torch.Tensor], lat: range, L: int):
    if t is None or not isinstance(t, torch.Tensor) or t.dim() == 0 or t.shape[0] != L:
        return t
    return t[lat.start:lat.stop]


def shard_problem(problem: dict, L: int, world: int, rank: int) -> dict:
    """Restrict every per-latent tensor (leading dim == L) of ``problem`` to this rank's block."""
    lat = shard_latents(L, world, rank)
    per_latent = ("sigma", "lengthscale", "group_diff", "mu", "Lu_raw", "y")
    out = dict(problem)
    for k in per_latent:
        if k in out:
            out[k] = _slice(out[k], lat, L)
    out["latents"] = lat
    return out


def hip_local_elbo(p: dict) -> torch.Tensor:
    """Partial ELBO of the latents in ``p`` on this rank's GPU (the product path)."""
    from . import ops
    from .configs import spec_for_config
    spec, extra = spec_for_config(p)
    out = ops.svgp_forward(spec, p["X"], p["Z"], p["mu"], p["Lu_raw"], p["jitter"], p["whitened"],
                           y=p["y"], noise_sd=p["noise_sd"], want_moments=False, want_Lu=False, **extra)
    return out["elbo"]


def sharded_elbo(problem: dict, L: int, local_eval: Callable[[dict], torch.Tensor] = hip_local_elbo,
                 group=None) -> torch.Tensor:
    """ELBO of an L-latent model summed over all ranks of ``group``.

    ``problem`` holds the FULL model (or at least this rank's block, see ``presharded``);
    ranks with no latents (L < world) contribute zero."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    p = problem if problem.get("presharded") else shard_problem(problem, L, world, rank)
    lat = p.get("latents", shard_latents(L, world, rank))
    if len(lat) > 0:
        e = local_eval(p).to(torch.float64).reshape(())
    else:
        e = torch.zeros((), dtype=torch.float64, device=p["X"].device)
    if world > 1:
        e = e.clone()
        dist.all_reduce(e, op=dist.ReduceOp.SUM, group=group)
    return e
