"""Multi-GPU evaluation: latent GPs shard across ranks, one scalar all-reduce.

Every array on the path has a leading latent axis with no cross-latent
arithmetic, so rank r owns a contiguous block of latents
(``synthetic.shard_latents``), evaluates its partial ELBO with the fused HIP pass
and the only exchange is a sum of one fp64 scalar -- ``torch.distributed``
all-reduce, which is RCCL over xGMI with the ``nccl`` backend (``gloo`` in the
CPU tests).  X, Z and group ids are replicated (a few MB).

With fewer latents than ranks the spots shard instead (SURVEY.md §8e "Partitioning"): columns of
Kzx are independent through the solves and the reductions, so rank r evaluates all latents on a
contiguous slice of X / y, the factorisation is replicated, the log-likelihood parts add up and the
KL term is counted once (rank 0).
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


def shard_spots(problem: dict, world: int, rank: int) -> dict:
    """Restrict the per-spot arrays (X, y, gX) to this rank's contiguous slice of the N spots."""
    N = problem["X"].shape[0]
    sl = shard_latents(N, world, rank)          # same contiguous-block rule, applied to spots
    out = dict(problem)
    out["X"] = problem["X"][sl.start:sl.stop]
    out["y"] = problem["y"][..., sl.start:sl.stop]
    if out.get("gX") is not None:
        out["gX"] = problem["gX"][sl.start:sl.stop]
    out["spots"] = sl
    return out


def hip_local_terms(p: dict):
    """(sum of per-latent expected log-likelihoods, sum of per-latent KL) of ``p`` on this rank's GPU."""
    from . import ops
    from .configs import spec_for_config
    spec, extra = spec_for_config(p)
    out = ops.svgp_forward(spec, p["X"], p["Z"], p["mu"], p["Lu_raw"], p["jitter"], p["whitened"],
                           y=p["y"], noise_sd=p["noise_sd"], want_moments=False, want_Lu=False, **extra)
    return out["loglik"].sum(), out["kl"].sum()


def hip_local_elbo(p: dict) -> torch.Tensor:
    """Partial ELBO of the latents in ``p`` on this rank's GPU (the product path)."""
    ll, kl = hip_local_terms(p)
    return ll - kl


def _allreduce_scalar(e: torch.Tensor, group=None) -> torch.Tensor:
    """Sum one fp64 scalar over the ranks: RCCL on the device tensor (backend nccl), or through the host
    for gloo rehearsals."""
    if dist.get_backend(group) == "gloo" and e.is_cuda:
        h = e.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(e.device)
    e = e.clone()
    dist.all_reduce(e, op=dist.ReduceOp.SUM, group=group)
    return e


def sharded_elbo(problem: dict, L: int, local_eval: Callable[[dict], torch.Tensor] = hip_local_elbo,
                 group=None, local_terms: Optional[Callable] = None) -> torch.Tensor:
    """ELBO of an L-latent model summed over all ranks of ``group``.

    ``problem`` holds the FULL model (or this rank's latent block with ``presharded``).  L >= world:
    latents shard and ``local_eval`` returns each rank's partial ELBO.  L < world (and the full model at
    hand): spots shard, ``local_terms`` (default: the HIP pass) returns (log-lik sum, KL sum) and the KL
    is added by rank 0 only."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = problem["X"].device
    if world > 1 and L < world and not problem.get("presharded") and problem["X"].shape[0] >= world:
        terms = local_terms or (hip_local_terms if local_eval is hip_local_elbo else None)
        if terms is not None:
            ll, kl = terms(shard_spots(problem, world, rank))
            e = (ll.to(torch.float64) - (kl.to(torch.float64) if rank == 0 else 0.0)).reshape(())
            return _allreduce_scalar(e, group)
    p = problem if problem.get("presharded") else shard_problem(problem, L, world, rank)
    lat = p.get("latents", shard_latents(L, world, rank))
    if len(lat) > 0:
        e = local_eval(p).to(torch.float64).reshape(())
    else:
        e = torch.zeros((), dtype=torch.float64, device=dev)
    if world > 1:
        e = _allreduce_scalar(e, group)
    return e
