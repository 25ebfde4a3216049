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


class AbiCommunicator:
    """RCCL communicator held through the C ABI (gpz_comm_init / gpz_allreduce_sum_f64, include/gpzoo_hip.h):
    the exchange a non-Python client of the library uses.  Rank 0 draws the 128-byte id and it travels over an
    existing torch.distributed group of ANY backend (gloo is enough: it is only a bootstrap), or is given
    directly for a one-rank communicator.  One per process, bound to ``device``."""

    def __init__(self, device: torch.device, group=None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib.load(), C
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(self._lib.gpz_comm_unique_id(buf), "gpz_comm_unique_id")
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if world > 1:
            if dist.get_backend(group) == "nccl":
                ident = ident.to(device)
            dist.broadcast(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(ident.cpu().numpy().tobytes())
        self.device, self.world, self.rank = device, world, rank
        self._comm = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self._lib.gpz_comm_init(C.byref(self._comm), world, rank, raw), "gpz_comm_init")

    def allreduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a contiguous fp64 CUDA tensor, on torch's current stream."""
        from . import _lib
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.device == self.device):
            raise ValueError("AbiCommunicator.allreduce_sum_ needs a contiguous float64 tensor on its device")
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.gpz_allreduce_sum_f64(self._comm, self._C.c_void_p(t.data_ptr()), t.numel(),
                                                       self._C.c_void_p(s)), "gpz_allreduce_sum_f64")
        return t

    def close(self):
        if self._comm:
            self._lib.gpz_comm_destroy(self._comm)
            self._comm = self._C.c_void_p()


def _allreduce_scalar(e: torch.Tensor, group=None, comm: Optional["AbiCommunicator"] = None) -> torch.Tensor:
    """Sum one fp64 scalar over the ranks: RCCL on the device tensor (backend nccl), or through the host
    for gloo rehearsals; with ``comm`` the C-ABI collective (gpz_allreduce_sum_f64) instead of torch.distributed."""
    if comm is not None:
        return comm.allreduce_sum_(e.detach().to(torch.float64).clone().contiguous())
    if dist.get_backend(group) == "gloo" and e.is_cuda:
        h = e.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h.to(e.device)
    e = e.clone()
    dist.all_reduce(e, op=dist.ReduceOp.SUM, group=group)
    return e


def allreduce_shared_grads(params, group=None) -> None:
    """Training on latent shards: the per-latent parameters (mu, Lu, vector sigma / lengthscale) live on the rank that owns
    the latent and need no exchange, but a parameter every latent shares -- the inducing points Z, a scalar kernel
    hyper-parameter, MGGP's group_diff_param -- receives on each rank only its shard's part of the gradient.  Call this after
    ``loss.backward()`` with those parameters: their ``.grad`` tensors are summed over the ranks in one flattened all-reduce
    (RCCL with the ``nccl`` backend; the reference has no multi-GPU code -- its optimizer step, ``utilities.py:485-489``,
    then runs unchanged on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    # The flattened buffer must have the same length on every rank: it is sized from the parameter LIST, and a parameter
    # that received no gradient on this rank (its shard left it unused) contributes zeros -- sized from the gradients
    # that happen to exist, ranks could enter the collective with different lengths (a hang under RCCL).
    params = [p for p in params if p is not None]
    if not params:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    flat = torch.cat([g.reshape(-1).to(torch.float64) for g in grads])
    if flat.is_cuda and dist.get_backend(group) == "gloo":      # rehearsals: through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat = host.to(flat.device)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].reshape(g.shape).to(g.dtype))
        o += n


def sharded_elbo(problem: dict, L: int, local_eval: Callable[[dict], torch.Tensor] = hip_local_elbo,
                 group=None, local_terms: Optional[Callable] = None, comm: Optional[AbiCommunicator] = None) -> torch.Tensor:
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
            return _allreduce_scalar(e, group, comm)
    p = problem if problem.get("presharded") else shard_problem(problem, L, world, rank)
    lat = p.get("latents", shard_latents(L, world, rank))
    if len(lat) > 0:
        e = local_eval(p).to(torch.float64).reshape(())
    else:
        e = torch.zeros((), dtype=torch.float64, device=dev)
    if dist.is_initialized() or comm is not None:   # also at world size 1: the collective is part of the path
        e = _allreduce_scalar(e, group, comm)
    return e
