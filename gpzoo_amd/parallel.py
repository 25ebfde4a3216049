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

    def _call(self, name, *args):
        from . import _lib
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(getattr(self._lib, name)(self._comm, *args, self._C.c_void_p(st)), name)

    def _check(self, t: torch.Tensor, dtype=None):
        if not (t.is_cuda and t.is_contiguous() and t.device == self.device and (dtype is None or t.dtype == dtype)):
            raise ValueError("AbiCommunicator needs contiguous tensors on its device" +
                             ("" if dtype is None else f" of dtype {dtype}"))

    def allgather(self, t: torch.Tensor) -> torch.Tensor:
        """(world, *t.shape): every rank's ``t`` (same shape and dtype everywhere), gpz_allgather."""
        self._check(t)
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._call("gpz_allgather", self._C.c_void_p(t.data_ptr()), self._C.c_void_p(out.data_ptr()),
                   t.numel() * t.element_size())
        return out

    def reduce_scatter_sum(self, t: torch.Tensor) -> torch.Tensor:
        """``t`` (world, ...) fp32 on every rank -> this rank's slice of the element-wise sum, gpz_reduce_scatter_sum_f32."""
        self._check(t, torch.float32)
        if t.shape[0] != self.world:
            raise ValueError("reduce_scatter_sum: the leading extent must be the world size")
        out = torch.empty(tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._call("gpz_reduce_scatter_sum_f32", self._C.c_void_p(t.data_ptr()), self._C.c_void_p(out.data_ptr()),
                   out.numel())
        return out

    def allreduce_sum_f32_(self, t: torch.Tensor) -> torch.Tensor:
        self._check(t, torch.float32)
        self._call("gpz_allreduce_sum_f32", self._C.c_void_p(t.data_ptr()), t.numel())
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


def allreduce_shared_grads(params, group=None, comm: Optional["AbiCommunicator"] = None) -> None:
    """Training on latent shards: the per-latent parameters (mu, Lu, vector sigma / lengthscale) live on the rank that owns
    the latent and need no exchange, but a parameter every latent shares -- the inducing points Z, a scalar kernel
    hyper-parameter, MGGP's group_diff_param -- receives on each rank only its shard's part of the gradient.  Call this after
    ``loss.backward()`` with those parameters: their ``.grad`` tensors are summed over the ranks in one flattened all-reduce
    (RCCL with the ``nccl`` backend; the reference has no multi-GPU code -- its optimizer step, ``utilities.py:485-489``,
    then runs unchanged on every rank)."""
    if comm is None and (not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1):
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
    if comm is not None:                                        # the C-ABI collective (gpz_allreduce_sum_f64)
        comm.allreduce_sum_(flat)
    elif flat.is_cuda and dist.get_backend(group) == "gloo":    # rehearsals: through the host
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


# ------------------------------------------------------------------------------------------------------------------
# Latent-sharded Poisson NSF step (SURVEY §8e "Caveat", §8f #2).  With the Poisson likelihood of the factor models the
# rate  softplus(V) * (softplus(W) @ exp(F))  mixes the latents (reference likelihoods.py:49-53, 74-97), so latent
# shards are no longer independent: every rank needs q(F) of ALL latents for the genes it evaluates.  Two exchanges would
# do -- all-gather q(F)'s moments, 2 L N_b s bytes, or reduce-scatter partial rates, E D N_b s bytes; at Slide-seq
# sizes (L = 20, D = 17 702, N_b = 7000: 1.1 MB against 1.5 GB for E = 3) the first wins by three orders of magnitude:
#   rank r:  q(F_r) = GP forward of its latents (fused HIP pass)            (L_r, N_b) mean, scale
#            all-gather  ->  mean, scale (L, N_b) on every rank             RCCL / xGMI, 2 L N_b s bytes
#            gpz_poisson_nsf on its block of genes, all latents             log-lik part, dW rows (owned), dV part,
#                                                                           d mean / d scale parts (L, N_b)
#            all-reduce of the scalar; reduce-scatter of d mean / d scale   each latent's gradient returns to its owner
#   dV (the size factors are replicated) and the shared GP parameters (Z, scalar hyper-parameters) are summed by
#   allreduce_shared_grads after backward().
# ------------------------------------------------------------------------------------------------------------------

def _padded_block(L: int, world: int) -> int:
    return (L + world - 1) // world


def _dist_allgather(t: torch.Tensor, group, comm):
    if comm is not None:
        return comm.allgather(t.contiguous())
    world = dist.get_world_size(group)
    via_host = t.is_cuda and dist.get_backend(group) == "gloo"
    src = (t.detach().cpu() if via_host else t.detach()).contiguous().reshape(-1)
    out = torch.empty(world * src.numel(), dtype=src.dtype, device=src.device)     # flat: every backend takes this form
    dist.all_gather_into_tensor(out, src, group=group)
    out = out.reshape((world,) + tuple(t.shape))
    return out.to(t.device) if via_host else out


def _dist_reduce_scatter(t: torch.Tensor, group, comm):
    """t (world, ...) -> this rank's slice of the sum over ranks."""
    if comm is not None:
        if t.dtype == torch.float32:
            return comm.reduce_scatter_sum(t.contiguous())
        if t.dtype == torch.float64:               # the C ABI's fp64 exchange is the all-reduce: sum, keep the slice
            return comm.allreduce_sum_(t.detach().contiguous().clone())[comm.rank]
        raise ValueError(f"reduce-scatter through the C-ABI communicator takes float32 or float64, got {t.dtype}")
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo":          # gloo has no reduce-scatter: all-reduce, keep the slice
        h = t.detach().cpu().contiguous() if t.is_cuda else t.detach().contiguous().clone()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        return h[rank].to(t.device)
    out = torch.empty(t[0].numel(), dtype=t.dtype, device=t.device)
    dist.reduce_scatter_tensor(out, t.contiguous().reshape(-1), op=dist.ReduceOp.SUM, group=group)
    return out.reshape(tuple(t.shape[1:]))


def hip_local_poisson(mean, scale, eps, W_pos, V_pos, y, with_lgamma):
    """(loglik, dmean, dscale, dW, dV) of this rank's genes through the fused HIP kernel (the product path)."""
    from . import ops
    return ops.poisson_nsf(mean, scale, eps, W_pos, V_pos, y, with_lgamma)


class _ShardedPoissonLogLik(torch.autograd.Function):
    """Expected Poisson log-likelihood over ALL genes and latents; inputs are this rank's latent block of q(F)'s moments
    and its gene block of the loadings / counts.  The value is the same on every rank."""

    @staticmethod
    def forward(ctx, mean_loc, scale_loc, W_rows, V_pos, y_rows, eps, L, group, comm, local, with_lgamma):
        have_group = dist.is_available() and dist.is_initialized()
        world = comm.world if comm is not None else dist.get_world_size(group) if have_group else 1
        rank = comm.rank if comm is not None else dist.get_rank(group) if have_group else 0
        exchange = comm is not None or have_group      # (a one-rank group still runs the collectives: identity)
        lat = shard_latents(L, world, rank)
        Lb = _padded_block(L, world)
        N = mean_loc.shape[-1]
        if mean_loc.shape[0] != len(lat):
            raise ValueError(f"rank {rank} owns latents {lat.start}..{lat.stop - 1} but was given {mean_loc.shape[0]} rows")

        def pad(t):                                 # equal-size contributions: blocks are padded to ceil(L / world) rows
            if t.shape[0] == Lb:
                return t.detach()
            return torch.cat([t.detach(), torch.ones((Lb - t.shape[0], N), dtype=t.dtype, device=t.device)])

        if exchange:
            both = _dist_allgather(torch.stack([pad(mean_loc), pad(scale_loc)]), group, comm)     # (world, 2, Lb, N)
            rows = torch.cat([torch.arange(r * Lb, r * Lb + len(shard_latents(L, world, r))) for r in range(world)])
            mean = both[:, 0].reshape(world * Lb, N)[rows.to(both.device)]
            scale = both[:, 1].reshape(world * Lb, N)[rows.to(both.device)]
        else:
            mean, scale = mean_loc.detach(), scale_loc.detach()
        ll, dmean, dscale, dW, dV = local(mean, scale, eps, W_rows.detach(), V_pos.detach(), y_rows, with_lgamma)
        ll = ll.detach().to(torch.float64).reshape(())
        if exchange:
            ll = _allreduce_scalar(ll, group, comm)
            g = torch.zeros((world, 2, Lb, N), dtype=dmean.dtype, device=dmean.device)
            for r in range(world):
                lr = shard_latents(L, world, r)
                g[r, 0, :len(lr)] = dmean[lr.start:lr.stop]
                g[r, 1, :len(lr)] = dscale[lr.start:lr.stop]
            mine = _dist_reduce_scatter(g, group, comm)                                             # (2, Lb, N)
            dmean, dscale = mine[0, :len(lat)], mine[1, :len(lat)]
        ctx.save_for_backward(dmean.to(mean_loc.dtype), dscale.to(scale_loc.dtype), dW.to(W_rows.dtype), dV.to(V_pos.dtype))
        return ll.to(mean_loc.dtype)

    @staticmethod
    def backward(ctx, g):
        dmean, dscale, dW, dV = ctx.saved_tensors
        return g * dmean, g * dscale, g * dW, g * dV, None, None, None, None, None, None, None


def sharded_poisson_loglik(mean_loc, scale_loc, W_rows, V_pos, y_rows, eps, L: int, *, group=None,
                           comm: Optional[AbiCommunicator] = None, local: Callable = hip_local_poisson,
                           with_lgamma: bool = True) -> torch.Tensor:
    """E_q[log p(y | F)] of the Poisson factor model summed over ALL genes, evaluated with latents AND genes sharded.

    mean_loc, scale_loc  (L_r, N)  q(F) moments of the latents this rank owns (``shard_latents(L, world, rank)``)
    W_rows               (D_r, L)  positive loadings of the genes this rank owns, all latents
    V_pos                (N,)      positive size factors (replicated: sum their gradient with allreduce_shared_grads)
    y_rows               (D_r, N)  counts of this rank's genes;   eps (E, L, N) the SAME draw on every rank
    Gradients flow to mean_loc / scale_loc (complete, for the owned latents), W_rows (complete) and V_pos (this rank's
    part).  ``comm``: the C-ABI communicator (gpz_allgather / gpz_reduce_scatter_sum_f32) instead of torch.distributed;
    ``local``: the per-rank evaluation (the fused HIP kernel; the CPU tests pass a torch one)."""
    return _ShardedPoissonLogLik.apply(mean_loc, scale_loc, W_rows, V_pos, y_rows, eps, L, group, comm, local, with_lgamma)


def sharded_nsf_step(gp, X, W_rows_raw, V_raw, y_rows, eps, L: int, *, group=None, comm=None,
                     local: Callable = hip_local_poisson, shared_params=(), gp_kwargs=None) -> torch.Tensor:
    """One training step's loss and gradients of an NSF-type model (reference likelihoods.py:74-97 with the minibatch
    objective of utilities.py:600-632) on latent + gene shards: ``gp`` holds this rank's latents, ``W_rows_raw`` /
    ``y_rows`` its genes, ``V_raw`` (N,) and ``shared_params`` (e.g. the inducing points) are replicated.  Returns the
    loss  -(E_q log p(y|F) - sum KL)  of the WHOLE model (identical on all ranks); after the call every parameter's
    ``.grad`` is the single-process gradient (the shards' own, or summed over the ranks for the replicated ones)."""
    import torch.nn.functional as Fn
    qF, qU, pU = gp(X, **(gp_kwargs or {}))
    ll = sharded_poisson_loglik(qF.mean, qF.scale, Fn.softplus(W_rows_raw), Fn.softplus(V_raw), y_rows, eps, L,
                                group=group, comm=comm, local=local)
    kl = torch.distributions.kl_divergence(qU, pU).sum() if pU is not None else _whitened_kl(qU).sum()
    kl_all = kl
    if comm is not None or (dist.is_available() and dist.is_initialized()):
        tot = _allreduce_scalar(kl.detach().to(torch.float64).reshape(()), group, comm).to(kl.dtype)
        kl_all = kl + (tot - kl.detach())          # the value of the whole model, the gradient of this rank's latents
    loss = -(ll - kl_all)
    loss.backward()
    allreduce_shared_grads([V_raw, *shared_params], group, comm=comm)
    return loss.detach()


def _whitened_kl(qU):
    """Per-latent KL(q(U) || N(0, I)) of a whitened GP (reference utilities.py:27-36, batched)."""
    Lz, mz = qU.scale_tril, qU.mean
    M = mz.shape[-1]
    return 0.5 * (-2.0 * torch.log(torch.diagonal(Lz, dim1=-2, dim2=-1)).sum(-1) + (Lz ** 2).sum((-2, -1)) + (mz ** 2).sum(-1) - M)
