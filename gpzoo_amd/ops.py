"""Tensor-level entry points: torch CUDA tensors in, HIP kernels underneath.

PyTorch is used for storage, streams and (in ``parallel.py``) torch.distributed
only; every numeric step of the hot path runs in libgpzoo_hip.so.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import threading
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import GPZ_F32, GPZ_F64, KernelDesc, SvgpProblem


@dataclass
class KernelSpec:
    """Flattened hyper-parameters of one gpzoo kernel object (see kernels.py)."""
    kind: int
    sigma: torch.Tensor            # (L,)
    lengthscale: torch.Tensor      # (L,)
    batched: bool                  # False: scalar parameters -> results drop the latent axis
    group_a: Optional[torch.Tensor] = None   # (L,) effective multiplier of r^2_group
    group_r2: Optional[torch.Tensor] = None  # (G,G)
    group_pow: float = 1.0

    @property
    def L(self) -> int:
        return int(self.sigma.numel())


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return GPZ_F32
    if t.dtype == torch.float64:
        return GPZ_F64
    raise TypeError(f"gpzoo_amd supports float32/float64 tensors, got {t.dtype}")


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gpzoo_amd computes on the GPU only (HIP kernels, no CPU fallback): "
                               "move the model and inputs to a cuda device")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device: torch.device):
    """torch's current stream ON THE TENSORS' DEVICE (not the process-wide current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _same_device(*ts) -> torch.device:
    """All tensor arguments must be CUDA tensors of ONE device; returns it.  The library's launches,
    memsets and event records go to the current HIP device, so every entry point runs under
    ``torch.cuda.device(dev)`` with this device (a model moved with ``.to('cuda:1')`` while the process's
    current device is 0 would otherwise launch on GPU 0 against GPU-1 pointers)."""
    dev = None
    for t in ts:
        if t is None or not isinstance(t, torch.Tensor) or not t.is_floating_point():
            continue                       # integer group ids / index tables are moved by the wrappers
        if not t.is_cuda:
            raise RuntimeError("gpzoo_amd computes on the GPU only (HIP kernels, no CPU fallback): "
                               "move the model and inputs to a cuda device")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"gpzoo_amd: tensor arguments live on different devices ({dev} and {t.device})")
    if dev is None:
        raise RuntimeError("gpzoo_amd: no tensor argument to take the device from")
    return dev


def _on_device(fn):
    """Decorator: validate that every tensor argument shares one CUDA device and make it current for the call."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        flat = [a for a in args if isinstance(a, torch.Tensor)] + [v for v in kwargs.values() if isinstance(v, torch.Tensor)]
        for a in args:
            if isinstance(a, KernelSpec):
                flat += [a.sigma, a.lengthscale, a.group_a, a.group_r2]
        dev = _same_device(*flat)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


_workspaces: dict = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Scratch buffer of at least ``nbytes`` for the library call about to be issued.  One buffer per
    (device, stream): calls on one stream are ordered and may share it, calls on different streams may not."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        _workspaces.pop(key, None)
        ws = None
        ws = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def release_workspaces():
    _workspaces.clear()


def _desc(spec: KernelSpec, dtype: torch.dtype, keep: list) -> KernelDesc:
    def prep(t):
        if t is None:
            return None
        t = t.detach().to(dtype).contiguous()
        keep.append(t)
        return t
    sig, ell = prep(spec.sigma.reshape(-1)), prep(spec.lengthscale.reshape(-1))
    ga, gr2 = prep(spec.group_a), prep(spec.group_r2)
    d = KernelDesc()
    d.kind, d.n_latent, d.dtype = spec.kind, spec.L, _dt(sig)
    d.n_groups = 0 if gr2 is None else int(gr2.shape[0])
    d.sigma, d.lengthscale = sig.data_ptr(), ell.data_ptr()
    d.group_a = None if ga is None else ga.data_ptr()
    d.group_r2 = None if gr2 is None else gr2.data_ptr()
    d.group_pow = float(spec.group_pow)
    return d


@_on_device
def kfill(spec: KernelSpec, A: torch.Tensor, B: torch.Tensor, gA=None, gB=None, jitter: float = 0.0,
          out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """K[l,i,j] = k_l(A_i, B_j): (L,nA,nB), or (nA,nB) for scalar-parameter kernels."""
    _need_cuda(A, B, spec.sigma)
    lib = _lib.load()
    keep: list = []
    A = A.detach().contiguous()
    B = B.detach().to(A.dtype).contiguous()
    if A.dim() != 2 or B.dim() != 2 or A.shape[1] != B.shape[1]:
        raise ValueError(f"expected (n,d) inputs with equal d, got {tuple(A.shape)} and {tuple(B.shape)}")
    d = _desc(spec, A.dtype, keep)
    if spec.kind == _lib.KERNEL_MGGP_RBF:
        gA, gB = _group_ids(gA, A.shape[0], A.device, "groupsX"), _group_ids(gB, B.shape[0], A.device, "groupsZ")
        G = int(spec.group_r2.shape[0])
        # stand-alone call: checked here with ONE device-to-host sync for both id vectors (the fused pass flags it on the
        # device and rides on the sync it needs anyway)
        bad = [((g_ < 0) | (g_ >= G)).any() for g_ in (gA, gB) if g_.numel()]
        if bad and bool(torch.stack(bad).any()):
            raise IndexError("index out of range in self: a group id is outside [0, n_groups)")
    else:
        gA = gB = None
    odt = A.dtype if out_dtype is None else out_dtype
    nA, nB = A.shape[0], B.shape[0]
    K = torch.empty((spec.L, nA, nB), dtype=odt, device=A.device)
    if nA == 0 or nB == 0:      # an empty point set gives an empty matrix, as torch.cdist does for the reference
        return K if spec.batched else K[0]
    rc = lib.gpz_kfill(C.byref(d), _ptr(A), nA, _ptr(B), nB, A.shape[1], _ptr(gA), _ptr(gB), _ptr(K), nB,
                       nA * nB, float(jitter), _dt(K), _stream(A.device))
    _lib.check(rc, "gpz_kfill")
    return K if spec.batched else K[0]


@_on_device
def kgrad(spec: KernelSpec, A: torch.Tensor, B: torch.Tensor, Kbar: torch.Tensor, gA=None, gB=None,
          want_points: bool = True):
    """Backward of ``kfill`` (gpz_kgrad): Kbar = dLoss/dK of shape (L,nA,nB) -> (grad_theta (L,3) fp64 =
    d/d(sigma, lengthscale, effective group multiplier), grad_A (nA,d) fp64 or None)."""
    lib = _lib.load()
    keep: list = []
    A = A.detach().contiguous()
    B = B.detach().to(A.dtype).contiguous()
    d = _desc(spec, A.dtype, keep)
    nA, nB, L = A.shape[0], B.shape[0], spec.L
    Kbar = Kbar.detach().to(A.dtype).reshape(L, nA, nB).contiguous()
    if spec.kind == _lib.KERNEL_MGGP_RBF:
        gA, gB = _group_ids(gA, nA, A.device, "groupsX"), _group_ids(gB, nB, A.device, "groupsZ")
    else:
        gA = gB = None
    gth = torch.empty((L, 4), dtype=torch.float64, device=A.device)
    gpt = torch.empty((nA, 4), dtype=torch.float64, device=A.device) if want_points else None
    if nA == 0 or nB == 0:      # nothing to contract
        return gth.zero_()[:, :3], (gpt.zero_()[:, :A.shape[1]] if want_points else None)
    nbytes = lib.gpz_kgrad_workspace_bytes(nA, L)
    ws = _workspace(A.device, nbytes)
    rc = lib.gpz_kgrad(C.byref(d), _ptr(A), nA, _ptr(B), nB, A.shape[1], _ptr(gA), _ptr(gB), _ptr(Kbar), nB, nA * nB,
                       _ptr(gth), _ptr(gpt), _ptr(ws), ws.numel(), _stream(A.device))
    _lib.check(rc, "gpz_kgrad")
    return gth[:, :3], (gpt[:, :A.shape[1]] if want_points else None)


def pairwise_distance(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Euclidean distances (nA,nB) -- what RBF.forward(return_distance=True) hands back."""
    one = torch.ones(1, dtype=A.dtype, device=A.device)
    return kfill(KernelSpec(_lib.KERNEL_DISTANCE, one, one, False), A, B)


def _group_ids(g, n: int, dev, name: str) -> torch.Tensor:
    """int64 group ids on the device, one per point (reference kernels.py:99-100, 177-178, 209-210 index the
    embedding with them: a wrong length or an id outside [0, n_groups) raises there, and here)."""
    g = g.detach().to(device=dev, dtype=torch.int64).reshape(-1).contiguous()
    if g.numel() != n:
        raise IndexError(f"{name} has {g.numel()} entries for {n} points")
    return g


def _raise_info(info: torch.Tensor, what: str):
    """LAPACK convention: info > 0 = not positive-definite (order of the failing minor), info < 0 = an argument
    was illegal -- here: a group id outside [0, n_groups), flagged by the covariance fill on the device."""
    if bool((info == -7).any()):
        raise RuntimeError(f"{what}: a hand-off inside the one-launch factorisation timed out (info = -7); "
                           "set GPZ_FACTOR_PATH=launches to use the launch-per-step path")
    if bool((info < 0).any()):
        raise IndexError("index out of range in self: a group id (groupsX / groupsZ) is outside [0, n_groups)")
    _raise_not_pd(info, what)


def _raise_not_pd(info: torch.Tensor, what: str):
    bad = torch.nonzero(info)
    b = int(bad[0, 0])
    k = int(info[b])
    prefix = f"(Batch element {b}): " if info.numel() > 1 else ""
    raise torch.linalg.LinAlgError(
        f"{what}: {prefix}The factorization could not be completed because the input is not "
        f"positive-definite (the leading minor of order {k} is not positive-definite).")


class _DeferredInfo:
    """`info` words of the factorisations launched inside one ``deferred_info()`` block, still on the device."""

    def __init__(self):
        self.items = []          # (info tensor, what, cache committed on trust or None)
        self.flags = []          # (0-d bool tensor "arguments valid", callable that raises torch's own error)
        self.checked = False

    def add(self, info: torch.Tensor, what: str, cache=None):
        self.items.append((info, what, cache))
        self.checked = False

    def add_flag(self, ok: torch.Tensor, reraise):
        self.flags.append((ok, reraise))
        self.checked = False

    def any_bad(self) -> Optional[torch.Tensor]:
        """0-d bool device tensor: some registered info word is non-zero or some argument check failed (no sync)."""
        words = [i.reshape(-1) != 0 for i, _, _ in self.items] + [~f.reshape(1) for f, _ in self.flags]
        return torch.cat(words).any() if words else None

    def check(self, keep: bool = False):
        """ONE device-to-host sync for everything registered so far; raises what the eager check would have raised
        (torch.linalg.LinAlgError / IndexError / RuntimeError) for the first failing call, after invalidating every
        factor cache that was committed on trust.  ``keep``: the registrations stay (a captured graph writes the same
        tensors at every replay)."""
        items, flags = self.items, self.flags
        if not keep:
            self.items, self.flags = [], []
        self.checked = True
        if not items and not flags:
            return
        words = [i.reshape(-1) != 0 for i, _, _ in items] + [~f.reshape(1) for f, _ in flags]
        if not bool(torch.cat(words).any()):
            return
        for _, _, cache in items:
            if cache is not None:
                cache.invalidate()
        for info, what, _ in items:        # in call order: a failed factorisation comes before the NaN moments it produced
            if bool(info.any()):
                _raise_info(info, what)
        for ok, reraise in flags:
            if not bool(ok):
                reraise()


_deferred = threading.local()


@contextlib.contextmanager
def deferred_info():
    """Inside the block, forward passes do not stop the launch queue to read their factorisation's ``info`` word: it
    stays on the device and is read ONCE when the block ends (or at ``.check()`` on the yielded object), raising what
    the call itself would have raised -- torch.linalg.LinAlgError for a Kzz that is not positive-definite (gp.py:213,
    270, 360).  A training step wraps forward + ``loss.backward()`` in it and puts the optimiser step behind it: the
    backward pass's launches are queued while the forward still runs (at the notebooks' sizes the eager check costs a
    drained queue per step: 0.4 ms of a 4 ms step at BASELINE configs[1]), a failed factorisation still raises before
    the parameters are touched.  What the block computed from a failed factor is garbage, as it would be unreachable
    eagerly.  Nested blocks register with the innermost one."""
    d = _DeferredInfo()
    stack = _deferred.__dict__.setdefault("stack", [])
    stack.append(d)
    try:
        yield d
    except BaseException:
        stack.pop()
        raise
    stack.pop()
    d.check()


def _deferring():
    stack = _deferred.__dict__.get("stack")
    return stack[-1] if stack else None


def checked_dist(cls, *args, **kwargs):
    """``cls(*args, **kwargs)`` -- a torch distribution -- with its argument validation (a device reduction and a HOST
    SYNC per constrained argument: ``Normal(loc, scale)`` stops the launch queue twice) moved to the end of the enclosing
    ``deferred_info()`` block: the constraint checks are queued as device flags, read with the block's one sync, and a
    violation raises what the constructor itself raises (it is re-run with validation on).  Outside a block: the plain
    constructor."""
    d = _deferring()
    if d is None or kwargs.get("validate_args") is False:
        return cls(*args, **kwargs)
    from torch import distributions as _D
    if not _D.Distribution._validate_args:          # validation switched off globally: nothing to defer
        return cls(*args, **kwargs)
    dist = cls(*args, validate_args=False, **kwargs)
    oks = []
    for name, constraint in dist.arg_constraints.items():
        if getattr(constraint, "is_dependent", False) or name not in dist.__dict__:
            continue
        oks.append(constraint.check(getattr(dist, name)).all())
    if oks:
        d.add_flag(torch.stack(oks).all(), lambda: cls(*args, validate_args=True, **kwargs))
    return dist


def freeze_spec(spec: KernelSpec) -> KernelSpec:
    """The same kernel with private copies of its (tiny) tensors: what a backward pass must hold so that edits of the
    live parameters between forward and backward (``.data`` writes, optimiser steps of another closure) cannot change
    the problem it differentiates."""
    cp = lambda t: None if t is None else t.detach().contiguous().clone()
    return KernelSpec(spec.kind, cp(spec.sigma), cp(spec.lengthscale), spec.batched, cp(spec.group_a), cp(spec.group_r2),
                      spec.group_pow)


@_on_device
def cholesky(A: torch.Tensor) -> torch.Tensor:
    """Lower Cholesky factor of (M,M) or (L,M,M), fp32 or fp64 storage (fp64 arithmetic); raises
    torch.linalg.LinAlgError when a matrix is not positive-definite (what gp.py:213/270/360 callers see)."""
    _need_cuda(A)
    lib = _lib.load()
    M = A.shape[-1]
    W = A.detach().reshape(-1, M, M).contiguous().clone()
    batch = W.shape[0]
    info = torch.empty(batch, dtype=torch.int32, device=A.device)
    nbytes = lib.gpz_potrf_workspace_bytes(M, batch)
    ws = _workspace(A.device, nbytes)
    rc = lib.gpz_potrf_batched(_ptr(W), _dt(W), M, M, M * M, batch, _ptr(info), _ptr(ws), ws.numel(), _stream(A.device))
    _lib.check(rc, "gpz_potrf_batched")
    if bool(info.any()):
        _raise_info(info, "linalg.cholesky")
    return W.reshape(A.shape)


@_on_device
def solve_triangular_lower(Lc: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Lc^{-1} B for lower-triangular (.., M, M) and (.., M, N): blocked forward substitution (gpz_trsm_lln_batched)."""
    _need_cuda(Lc, B)
    lib = _lib.load()
    M, N = B.shape[-2], B.shape[-1]
    Lw = Lc.detach().to(B.dtype).reshape(-1, M, M).contiguous()
    Bw = B.detach().reshape(-1, M, N).contiguous().clone()
    batch = Bw.shape[0]
    if Lw.shape[0] != batch:
        Lw = Lw.expand(batch, M, M).contiguous()
    nbytes = lib.gpz_trsm_workspace_bytes(M, N, batch)
    ws = _workspace(B.device, nbytes)
    rc = lib.gpz_trsm_lln_batched(_ptr(Lw), M, M * M, _ptr(Bw), N, M * N, _dt(Bw), M, N, batch, _ptr(ws), ws.numel(),
                                  _stream(B.device))
    _lib.check(rc, "gpz_trsm_lln_batched")
    return Bw.reshape(B.shape)


class FactorCache:
    """Device buffer for chol(Kzz), its inverse and log-determinant, reused while its inputs are unchanged.

    Validity is decided by CONTENT, not by tensor identity: next to the factor the cache keeps a byte copy
    of everything the factor depends on (Z, sigma, lengthscale, effective group multiplier, the group r^2
    table, groupsZ) and compares it on the device before every use (one concatenation + one ``torch.equal``,
    a few KB), plus a host key (kernel family, L, exponent, jitter, dtype, shapes).  In-place edits through
    ``.data`` (``p.data.fill_()``, ``p.data = ...``), re-created Parameters that land on a freed pointer and
    optimiser steps are therefore all seen (SURVEY §8f "next" #3: the reference refactors Kzz every step even
    when those are frozen)."""

    def __init__(self):
        self.buf = None
        self.key = None
        self.snap = None
        self._pending = None
        # Counts the factorisations written into ``buf`` (and its invalidations).  A forward pass notes the value it
        # committed; its backward pass may skip the content check only while the counter still has that value, i.e.
        # while no other call has refactored into the shared buffer in between.
        self.generation = 0
        # Counts the calls that wrote their q(U) operands (LuE^T, muE, un-whitened LuE) behind the factor -- every call
        # on the buffer but a backward that took them from it.  A backward pass may do that (factor_cache_valid = 3)
        # only while this is still the value its forward left.
        self.qu_generation = 0
        self._rebuilds = False

    @staticmethod
    def fingerprint(tensors) -> torch.Tensor:
        return torch.cat([t.reshape(-1).view(torch.uint8) for t in tensors if t is not None])

    def attach(self, lib, p: "SvgpProblem", key, device, deps, trust: bool = False, trust_qu: bool = False):
        """``trust``: the caller vouches that the buffer still holds the factor of exactly these ``deps`` -- the
        backward pass of the call whose forward committed it, holding private copies of the inputs and having checked
        that ``generation`` is still the value that forward saw -- so the content comparison (a device reduction and a
        host sync) is skipped.  ``trust_qu`` (with ``trust``): no other forward has written its q(U) operands into the
        buffer since, so the backward takes them from there too."""
        nbytes = lib.gpz_svgp_factor_cache_bytes(C.byref(p))
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.key = self.snap = None
            self.generation += 1
        if trust and self.key == key and self.snap is not None:
            p.factor_cache = self.buf.data_ptr()
            p.factor_cache_valid = 3 if trust_qu else 1
            if not trust_qu:
                self.qu_generation += 1
            self._pending = (key, self.snap)
            self._rebuilds = False
            return True
        fp = self.fingerprint(deps)
        valid = (self.key == key and self.snap is not None and self.snap.shape == fp.shape
                 and bool(torch.equal(self.snap, fp)))
        p.factor_cache = self.buf.data_ptr()
        p.factor_cache_valid = int(valid)
        self.qu_generation += 1
        self._pending = (key, fp)
        self._rebuilds = not valid
        return valid

    def invalidate(self):
        self.key = self.snap = None
        self.generation += 1

    def commit(self):
        self.key, self.snap = self._pending
        if self._rebuilds:                 # the launch just wrote a new factor into the buffer
            self.generation += 1
            self._rebuilds = False


def factor_key(spec: "KernelSpec", Z, jitter, dtype) -> tuple:
    """Host half of the cache key; the tensor contents are compared on the device (FactorCache.attach)."""
    G = 0 if spec.group_r2 is None else int(spec.group_r2.shape[0])
    return (spec.kind, spec.L, float(spec.group_pow), float(jitter), str(dtype), tuple(Z.shape), G)


def _problem(spec: KernelSpec, X, Z, mu, Lu_raw, jitter, whitened, gX, gZ, clamp_min, keep: list):
    """Fill a gpz_svgp_problem from tensors (inputs only); returns (problem, (L, M, N, dtype, device, deps)) where
    ``deps`` are the prepared tensors chol(Kzz) depends on (what FactorCache fingerprints)."""
    dt = X.dtype
    if X.dim() != 2 or Z.dim() != 2 or X.shape[1] != Z.shape[1]:
        raise ValueError(f"expected X (N,d) and Z (M,d) with equal d, got {tuple(X.shape)} and {tuple(Z.shape)}")
    X = X.detach().contiguous()
    Z = Z.detach().to(dt).contiguous()
    L = spec.L
    M, N, dim = Z.shape[0], X.shape[0], X.shape[1]
    if mu.numel() != L * M or Lu_raw.numel() != L * M * M:
        raise ValueError(f"mu / Lu of shapes {tuple(mu.shape)} / {tuple(Lu_raw.shape)} do not match L={L} latents "
                         f"and M={M} inducing points")
    mu = mu.detach().to(dt).reshape(L, M).contiguous()
    Lu_raw = Lu_raw.detach().to(dt).reshape(L, M, M).contiguous()
    dev = X.device
    p = SvgpProblem()
    n0 = len(keep)
    p.k = _desc(spec, dt, keep)
    deps = [Z] + keep[n0:]
    p.dtype, p.whitened, p.d = _dt(X), int(whitened), dim
    p.N, p.M = N, M
    p.X, p.Z, p.mu, p.Lu_raw = X.data_ptr(), Z.data_ptr(), mu.data_ptr(), Lu_raw.data_ptr()
    if spec.kind == _lib.KERNEL_MGGP_RBF:
        if gX is None or gZ is None:
            raise ValueError("multi-group kernels need groupsX and groupsZ")
        gX, gZ = _group_ids(gX, N, dev, "groupsX"), _group_ids(gZ, M, dev, "groupsZ")
        p.gX, p.gZ = gX.data_ptr(), gZ.data_ptr()
        deps.append(gZ)
    p.jitter, p.var_clamp_min = float(jitter), float(clamp_min)
    keep.extend([X, Z, mu, Lu_raw, gX, gZ])
    return p, (L, M, N, dt, dev, deps)


@_on_device
def svgp_forward(spec: KernelSpec, X, Z, mu, Lu_raw, jitter: float, whitened: bool, *, gX=None, gZ=None,
                 y=None, noise_sd: Optional[float] = None, clamp_min: float = 1e-6, chunk: int = 0,
                 want_moments: bool = True, want_Lu: bool = True, want_chol: bool = False,
                 check_info: bool = True, cache: Optional[FactorCache] = None,
                 retain_wt: float = 0.0, materialize_kzx: Optional[bool] = None, narrow_tiles: bool = False,
                 panel_products: bool = False) -> dict:
    """One fused forward pass (gpz_svgp_forward).  Returns a dict with mean, scale
    (L,N), Lu (L,M,M), chol (L,M,M), kl (L,), loglik (L,), elbo () -- fp64 scalars.
    ``retain_wt`` > 0: keep Wt of every chunk for ``svgp_backward(wt_cache=out["wt_cache"])`` when it fits
    in that fraction of the free device memory (288 GB HBM: 52 GB at N=200k, M=2048, L=32, fp32).
    ``materialize_kzx``: True = write every Kzx chunk to HBM and run the triangular product on it (the reference's
    structure), False = the product that generates its covariance operand itself (fp32 RBF / Matern-3/2, d <= 2), None =
    the library's choice; ``narrow_tiles``: the 128 x 128-tile kernel of the other precisions for the two big fp32
    products; ``panel_products`` (fp32, M <= 512): both products in one launch on 64-column panels held in LDS.  Same Wt
    bits on every path."""
    _need_cuda(X, Z, mu, Lu_raw)
    if X.dim() == 2 and X.shape[0] == 0 and Z.dim() == 2 and Z.shape[0] > 0:
        # No data points: q(F) is empty and the ELBO is -sum(KL), as the reference's torch code gives for an (0,d) X.
        # The library wants N >= 1: evaluate at one stand-in point (the first inducing point) and drop its column.
        o = svgp_forward(spec, Z[:1].to(X.dtype), Z, mu, Lu_raw, jitter, whitened, gX=None if gZ is None else gZ[:1], gZ=gZ,
                         clamp_min=clamp_min, want_moments=want_moments, want_Lu=want_Lu, want_chol=want_chol,
                         check_info=check_info, cache=cache)
        for k in ("mean", "scale"):
            if k in o:
                o[k] = o[k][:, :0]
        o["loglik"] = torch.zeros_like(o["kl"])
        o["elbo"] = -o["kl"].sum()
        return o
    lib = _lib.load()
    keep: list = []
    p, (L, M, N, dt, dev, deps) = _problem(spec, X, Z, mu, Lu_raw, jitter, whitened, gX, gZ, clamp_min, keep)
    out = {}
    if materialize_kzx is not None:
        p.flags |= _lib.SVGP_MATERIALIZE_KZX if materialize_kzx else _lib.SVGP_GENERATE_KZX
    if narrow_tiles:
        p.flags |= _lib.SVGP_NARROW_TILES
    if panel_products:               # fp32, M <= 512: both products in one launch, panel by panel (csrc/gemmp.hip)
        p.flags |= _lib.SVGP_PANEL_PRODUCTS
    if y is not None:
        y = y.detach().to(dt).reshape(L, N).contiguous()
        p.y, p.noise_sd = y.data_ptr(), float(noise_sd)
    if want_moments:
        out["mean"] = torch.empty((L, N), dtype=dt, device=dev)
        out["scale"] = torch.empty((L, N), dtype=dt, device=dev)
        p.mean, p.scale = out["mean"].data_ptr(), out["scale"].data_ptr()
    if want_Lu:
        out["Lu"] = torch.empty((L, M, M), dtype=dt, device=dev)
        p.Lu = out["Lu"].data_ptr()
    if want_chol:
        out["chol"] = torch.empty((L, M, M), dtype=dt, device=dev)
        p.chol = out["chol"].data_ptr()
    scal = torch.empty(2 * L + 1, dtype=torch.float64, device=dev)     # every entry is written by the pass (reduce / elbo_sum kernels)
    info = torch.empty(L, dtype=torch.int32, device=dev)
    p.kl, p.loglik, p.elbo = scal.data_ptr(), scal.data_ptr() + 8 * L, scal.data_ptr() + 16 * L
    p.info = info.data_ptr()
    if cache is not None:
        cache.attach(lib, p, factor_key(spec, Z, jitter, dt), dev, deps)
        out["qu_generation"] = cache.qu_generation   # this call writes its q(U) operands behind the factor
    if retain_wt > 0:
        need = lib.gpz_svgp_wt_cache_bytes(C.byref(p), int(chunk))
        if 0 < need <= retain_wt * torch.cuda.mem_get_info(dev)[0]:
            out["wt_cache"] = torch.empty(need, dtype=torch.uint8, device=dev)
            p.wt_cache = out["wt_cache"].data_ptr()
    nbytes = lib.gpz_svgp_workspace_bytes(C.byref(p), int(chunk))
    if nbytes == 0:
        _lib.check(-1, "gpz_svgp_workspace_bytes")
    ws = _workspace(dev, nbytes)
    rc = lib.gpz_svgp_forward(C.byref(p), int(chunk), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc, "gpz_svgp_forward")
    out["path"] = lib.gpz_svgp_forward_path(C.byref(p), int(chunk))    # bit 0: wide tiles, bit 1: generated Kzx, 4: panel kernel
    later = _deferring() if check_info else None
    if later is not None:
        # inside deferred_info(): no sync here; the cache is committed on trust and invalidated by the block's check
        # should the factorisation have failed
        later.add(info, "linalg.cholesky", cache)
        check_info = False
    bad = check_info and bool(info.any())       # one device-to-host sync, shared by the cache decision and the raise
    if cache is not None:
        # a factor that failed must not be reused; without the host check the cache stays uncommitted
        if later is not None:
            cache.commit()
        elif not check_info or bad:
            cache.invalidate()
        else:
            cache.commit()
        out["factor_generation"] = cache.generation
    if bad:
        _raise_info(info, "linalg.cholesky")
    out["kl"], out["loglik"], out["elbo"] = scal[:L], scal[L:2 * L], scal[2 * L]
    out["info"] = info
    return out


@_on_device
def svgp_backward(spec: KernelSpec, X, Z, mu, Lu_raw, jitter: float, whitened: bool, g_mean, g_scale, scale, *,
                  gX=None, gZ=None, clamp_min: float = 1e-6, chunk: int = 0, cache: Optional[FactorCache] = None,
                  kernel_grads: bool = False, g_chol=None, wt_cache=None, g_kl=None, trust_cache: bool = False,
                  trust_qu: bool = False, narrow_tiles: bool = False, form: Optional[str] = None):
    """dLoss/dmu (L,M) and dLoss/dLu_raw (L,M,M) (gpz_svgp_backward); with ``kernel_grads`` also
    dLoss/d(sigma, lengthscale, effective group parameter) (L,3) and dLoss/dZ (M,d), both fp64.
    ``wt_cache``: the buffer a forward pass on the same inputs and ``chunk`` returned under "wt_cache".
    ``g_kl`` (L,): upstream gradient of the forward's per-latent ``kl``; its own
    gradient is folded into the results.  ``form``: "algebra" / "classic" / None (the library's choice by N / M): the
    N-sized work as one weighted symmetric accumulation plus M x M products, or as the products autograd would run."""
    _need_cuda(X, Z, mu, Lu_raw, g_mean, g_scale)
    if X.dim() == 2 and X.shape[0] == 0 and Z.dim() == 2 and Z.shape[0] > 0:
        # empty X (see svgp_forward): one stand-in point with zero upstream gradients leaves the KL / factor terms
        zero = torch.zeros((spec.L, 1), dtype=X.dtype, device=X.device)
        return svgp_backward(spec, Z[:1].to(X.dtype), Z, mu, Lu_raw, jitter, whitened, zero, zero, torch.ones_like(zero),
                             gX=None if gZ is None else gZ[:1], gZ=gZ, clamp_min=clamp_min, cache=cache,
                             kernel_grads=kernel_grads, g_chol=g_chol, g_kl=g_kl, trust_cache=trust_cache)
    lib = _lib.load()
    keep: list = []
    p, (L, M, N, dt, dev, deps) = _problem(spec, X, Z, mu, Lu_raw, jitter, whitened, gX, gZ, clamp_min, keep)
    info = torch.empty(L, dtype=torch.int32, device=dev)
    p.info = info.data_ptr()
    if narrow_tiles:                 # the 128 x 128-tile kernel for the fp32 products (csrc/gemm.hip), as in round 2
        p.flags |= _lib.SVGP_NARROW_TILES
    if form is not None:
        p.flags |= {"algebra": _lib.SVGP_BACKWARD_ALGEBRA, "classic": _lib.SVGP_BACKWARD_CLASSIC}[form]
    g = _lib.SvgpGrads()
    gm = g_mean.detach().to(dt).reshape(L, N).contiguous()
    gs = g_scale.detach().to(dt).reshape(L, N).contiguous()
    sc = scale.detach().to(dt).reshape(L, N).contiguous()
    grad_mu = torch.empty((L, M), dtype=dt, device=dev)
    grad_Lu = torch.empty((L, M, M), dtype=dt, device=dev)
    g.g_mean, g.g_scale, g.scale = gm.data_ptr(), gs.data_ptr(), sc.data_ptr()
    g.grad_mu, g.grad_Lu_raw = grad_mu.data_ptr(), grad_Lu.data_ptr()
    if kernel_grads and g_chol is not None:
        gc = g_chol.detach().to(dt).reshape(L, M, M).contiguous()
        keep.append(gc)
        g.g_chol = gc.data_ptr()
    if kernel_grads:
        gth = torch.zeros((L, 4), dtype=torch.float64, device=dev)
        gz = torch.zeros((M, 4), dtype=torch.float64, device=dev)
        g.grad_theta, g.grad_Z = gth.data_ptr(), gz.data_ptr()
    if g_kl is not None:
        gk = g_kl.detach().to(device=dev, dtype=torch.float64).reshape(-1).expand(L).contiguous()
        keep.append(gk)
        g.g_kl = gk.data_ptr()
    if cache is not None:
        cache.attach(lib, p, factor_key(spec, Z, jitter, dt), dev, deps, trust=trust_cache, trust_qu=trust_cache and trust_qu)
    if wt_cache is not None:
        if wt_cache.numel() != lib.gpz_svgp_wt_cache_bytes(C.byref(p), int(chunk)):
            raise ValueError("wt_cache does not belong to this problem / chunking")
        p.wt_cache, p.wt_cache_valid = wt_cache.data_ptr(), 1
    nbytes = lib.gpz_svgp_backward_workspace_bytes(C.byref(p), int(chunk))
    if nbytes == 0:
        _lib.check(-1, "gpz_svgp_backward_workspace_bytes")
    ws = _workspace(dev, nbytes)
    rc = lib.gpz_svgp_backward(C.byref(p), C.byref(g), int(chunk), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc, "gpz_svgp_backward")
    if cache is not None and not (p.factor_cache_valid & 1):
        # the backward refactored (cache miss): keep the result only if the factorisation succeeded
        if bool(info.any()):
            cache.invalidate()
        else:
            cache.commit()
    if kernel_grads:
        return grad_mu, grad_Lu, gth[:, :3], gz[:, :X.shape[1]]
    return grad_mu, grad_Lu


@_on_device
def wsvgp_precomputed(W, sigma, mu, Lu_raw) -> dict:
    """q(F) moments from a caller-supplied W (L,N,M) or (N,M): WSVGP.forward_precomputed."""
    _need_cuda(W, mu, Lu_raw)
    lib = _lib.load()
    dt = W.dtype
    M = W.shape[-1]
    W3 = W.detach().reshape(-1, W.shape[-2], M).contiguous()
    L, N = W3.shape[0], W3.shape[1]
    mu2 = mu.detach().to(dt).reshape(-1, M).expand(L, M).contiguous()
    Lu3 = Lu_raw.detach().to(dt).reshape(-1, M, M).expand(L, M, M).contiguous()
    sig = sigma.detach().to(dt).reshape(-1).expand(L).contiguous()
    out = {"mean": torch.empty((L, N), dtype=dt, device=W.device), "scale": torch.empty((L, N), dtype=dt, device=W.device),
           "Lu": torch.empty((L, M, M), dtype=dt, device=W.device)}
    nbytes = lib.gpz_wsvgp_precomputed_workspace_bytes(L, N, M, _dt(W3))
    ws = _workspace(W.device, nbytes)
    rc = lib.gpz_wsvgp_precomputed(_ptr(W3), _ptr(sig), _ptr(mu2), _ptr(Lu3), L, N, M, _dt(W3), _ptr(out["mean"]),
                                   _ptr(out["scale"]), _ptr(out["Lu"]), _ptr(ws), ws.numel(), _stream(W.device))
    _lib.check(rc, "gpz_wsvgp_precomputed")
    return out


@_on_device
def wsvgp_precomputed_backward(W, sigma, mu, Lu_raw, g_mean, g_scale, scale):
    """dLoss/d(mu (L,M), raw Lu (L,M,M), sigma (L,) fp64) of ``wsvgp_precomputed`` (gpz_wsvgp_precomputed_backward);
    W is a constant of the graph."""
    lib = _lib.load()
    dt = W.dtype
    M = W.shape[-1]
    W3 = W.detach().reshape(-1, W.shape[-2], M).contiguous()
    L, N = W3.shape[0], W3.shape[1]
    mu2 = mu.detach().to(dt).reshape(-1, M).expand(L, M).contiguous()
    Lu3 = Lu_raw.detach().to(dt).reshape(-1, M, M).expand(L, M, M).contiguous()
    sig = sigma.detach().to(dt).reshape(-1).expand(L).contiguous()
    gm = g_mean.detach().to(dt).reshape(L, N).contiguous()
    gs = g_scale.detach().to(dt).reshape(L, N).contiguous()
    sc = scale.detach().to(dt).reshape(L, N).contiguous()
    grad_mu = torch.empty((L, M), dtype=dt, device=W.device)
    grad_Lu = torch.empty((L, M, M), dtype=dt, device=W.device)
    grad_sig = torch.empty(L, dtype=torch.float64, device=W.device)
    nbytes = lib.gpz_wsvgp_precomputed_backward_workspace_bytes(L, N, M, _dt(W3))
    ws = _workspace(W.device, nbytes)
    rc = lib.gpz_wsvgp_precomputed_backward(_ptr(W3), _ptr(sig), _ptr(mu2), _ptr(Lu3), L, N, M, _dt(W3), _ptr(gm), _ptr(gs),
                                            _ptr(sc), _ptr(grad_mu), _ptr(grad_Lu), _ptr(grad_sig), _ptr(ws), ws.numel(),
                                            _stream(W.device))
    _lib.check(rc, "gpz_wsvgp_precomputed_backward")
    return grad_mu, grad_Lu, grad_sig


@_on_device
def knn(X, Z, K: int) -> torch.Tensor:
    """(N,K) int64 indices of the K nearest rows of Z per row of X, ties to the lower index."""
    _need_cuda(X, Z)
    lib = _lib.load()
    X = X.detach().contiguous()
    Z = Z.detach().to(X.dtype).contiguous()
    idx = torch.empty((X.shape[0], K), dtype=torch.int64, device=X.device)
    rc = lib.gpz_knn(_ptr(X), X.shape[0], _ptr(Z), Z.shape[0], X.shape[1], K, _dt(X), _ptr(idx), _stream(X.device))
    _lib.check(rc, "gpz_knn")
    return idx


def morton_order(X: torch.Tensor) -> torch.Tensor:
    """(N,) int64 permutation that sorts the rows of X (N,d), d <= 4, along a Z-order (Morton) curve, 16 bits per
    coordinate, ties in index order: the ``point_order`` of ``vnngp_backward`` -- points that share their nearest
    inducing points become neighbours in the pass's per-point records."""
    Xd = X.detach().double()
    lo = Xd.min(dim=0).values
    span = (Xd.max(dim=0).values - lo).clamp_min(1e-300)
    q = ((Xd - lo) / span * 65535.0).round().to(torch.int64)
    d = X.shape[1]
    key = torch.zeros(X.shape[0], dtype=torch.int64, device=X.device)
    for bit in range(16):
        for k in range(d):
            key |= ((q[:, k] >> bit) & 1) << (bit * d + k)
    return torch.argsort(key, stable=True)


@_on_device
def vnngp_forward(spec: KernelSpec, X, Z, mu, Lu_raw, jitter: float, K: int, clamp_min: float = 5e-2,
                  check_info: bool = True, idx=None, keep_state: bool = False) -> dict:
    """VNNGP forward (gpz_vnngp_forward): mean, scale (L,N), Lu, chol (L,M,M), idx (N,K).  ``keep_state``: also
    "state", the buffer ``vnngp_backward(state=...)`` of the same call takes the factor, S and the KL operands from."""
    _need_cuda(X, Z, mu, Lu_raw)
    lib = _lib.load()
    keep: list = []
    p, (L, M, N, dt, dev, deps) = _problem(spec, X, Z, mu, Lu_raw, jitter, False, None, None, clamp_min, keep)
    out = {"mean": torch.empty((L, N), dtype=dt, device=dev), "scale": torch.empty((L, N), dtype=dt, device=dev),
           "Lu": torch.empty((L, M, M), dtype=dt, device=dev), "chol": torch.empty((L, M, M), dtype=dt, device=dev)}
    info = torch.empty(L, dtype=torch.int32, device=dev)
    p.mean, p.scale, p.Lu, p.chol = (out[k].data_ptr() for k in ("mean", "scale", "Lu", "chol"))
    p.info = info.data_ptr()
    out["kl"] = torch.empty(L, dtype=torch.float64, device=dev)      # KL(qU || pU) per latent
    p.kl = out["kl"].data_ptr()
    out["idx"] = knn(X, Z, K) if idx is None else idx
    if keep_state:
        out["state"] = torch.empty(lib.gpz_vnngp_state_bytes(C.byref(p)), dtype=torch.uint8, device=dev)
        p.factor_cache, p.factor_cache_valid = out["state"].data_ptr(), 0
    nbytes = lib.gpz_vnngp_workspace_bytes(C.byref(p), K)
    if nbytes == 0:
        _lib.check(-1, "gpz_vnngp_workspace_bytes")
    ws = _workspace(dev, nbytes)
    rc = lib.gpz_vnngp_forward(C.byref(p), K, _ptr(out["idx"]), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc, "gpz_vnngp_forward")
    later = _deferring() if check_info else None
    if later is not None:
        later.add(info, "linalg.cholesky")
    elif check_info and bool(info.any()):
        _raise_info(info, "linalg.cholesky")
    return out


@_on_device
def vnngp_backward(spec: KernelSpec, X, Z, mu, Lu_raw, jitter: float, K: int, idx, g_mean, g_scale, *,
                   clamp_min: float = 5e-2, kernel_grads: bool = False, g_chol=None, g_kl=None, state=None,
                   point_order=None):
    """dLoss/dmu (L,M), dLoss/dLu_raw (L,M,M) of VNNGP (gpz_vnngp_backward); with ``kernel_grads`` also
    dLoss/d(sigma, lengthscale) (L,2) and dLoss/dZ (M,d), both fp64.  ``state``: what ``vnngp_forward(keep_state=True)``
    of the SAME inputs returned (consumed: good for one backward pass).  ``point_order`` (N,) int64: a permutation of
    the points the pass lays its records out in (``morton_order(X)``: the gather then reads runs of neighbouring records)."""
    _need_cuda(X, Z, mu, Lu_raw, g_mean, g_scale, idx)
    lib = _lib.load()
    keep: list = []
    p, (L, M, N, dt, dev, deps) = _problem(spec, X, Z, mu, Lu_raw, jitter, False, None, None, clamp_min, keep)
    info = torch.empty(L, dtype=torch.int32, device=dev)
    p.info = info.data_ptr()
    g = _lib.SvgpGrads()
    gm = g_mean.detach().to(dt).reshape(L, N).contiguous()
    gs = g_scale.detach().to(dt).reshape(L, N).contiguous()
    grad_mu = torch.empty((L, M), dtype=dt, device=dev)
    grad_Lu = torch.empty((L, M, M), dtype=dt, device=dev)
    g.g_mean, g.g_scale = gm.data_ptr(), gs.data_ptr()
    g.grad_mu, g.grad_Lu_raw = grad_mu.data_ptr(), grad_Lu.data_ptr()
    if kernel_grads and g_chol is not None:
        gc = g_chol.detach().to(dt).reshape(L, M, M).contiguous()
        keep.append(gc)
        g.g_chol = gc.data_ptr()
    if kernel_grads:
        gth = torch.zeros((L, 4), dtype=torch.float64, device=dev)
        gz = torch.zeros((M, 4), dtype=torch.float64, device=dev)
        g.grad_theta, g.grad_Z = gth.data_ptr(), gz.data_ptr()
    if g_kl is not None:
        gk = g_kl.detach().to(device=dev, dtype=torch.float64).reshape(-1).expand(L).contiguous()
        keep.append(gk)
        g.g_kl = gk.data_ptr()
    idx = idx.contiguous()
    if point_order is not None:
        po = point_order.detach().to(device=dev, dtype=torch.int64).contiguous()
        if po.numel() != N:
            raise ValueError(f"point_order has {po.numel()} entries for {N} points")
        keep.append(po)
        g.point_order = po.data_ptr()
    if state is not None:
        if state.numel() != lib.gpz_vnngp_state_bytes(C.byref(p)):
            raise ValueError("state does not belong to this problem")
        p.factor_cache, p.factor_cache_valid = state.data_ptr(), 5
    nbytes = lib.gpz_vnngp_backward_workspace_bytes(C.byref(p), K)
    if nbytes == 0:
        _lib.check(-1, "gpz_vnngp_backward_workspace_bytes")
    ws = _workspace(dev, nbytes)
    rc = lib.gpz_vnngp_backward(C.byref(p), C.byref(g), K, _ptr(idx), _ptr(ws), ws.numel(), _stream(dev))
    _lib.check(rc, "gpz_vnngp_backward")
    if kernel_grads:
        return grad_mu, grad_Lu, gth[:, :2], gz[:, :X.shape[1]]
    return grad_mu, grad_Lu


@_on_device
def poisson_nsf(mean, scale, eps, W_pos, V_pos, y, with_lgamma: bool = True):
    """Fused Monte-Carlo Poisson log-likelihood of the NSF models and its gradients (gpz_poisson_nsf).

    mean, scale (Lt,N); eps (E,Lt,N); W_pos (D,Lt) and V_pos (N,) positive; y (D,N).  Returns
    (loglik fp64 scalar, dmean, dscale, dW, dV) with the mean over the E samples already applied.
    Up to 32 samples go through one call (y is read once per pass whatever E is); more are split."""
    _need_cuda(mean, scale, eps, W_pos, V_pos, y)
    lib = _lib.load()
    f32 = torch.float32
    mean, scale = mean.detach().to(f32).contiguous(), scale.detach().to(f32).contiguous()
    eps, y = eps.detach().to(f32).contiguous(), y.detach().to(f32).contiguous()
    W_pos, V_pos = W_pos.detach().to(f32).contiguous(), V_pos.detach().to(f32).contiguous()
    Lt, N = mean.shape
    E, D = eps.shape[0], y.shape[0]
    dev = mean.device
    eg = 32                                  # samples per call (PMAXE of csrc/poisson.hip)
    total = torch.zeros((), dtype=torch.float64, device=dev)
    acc = [torch.zeros((Lt, N), dtype=f32, device=dev), torch.zeros((Lt, N), dtype=f32, device=dev),
           torch.zeros((D, Lt), dtype=f32, device=dev), torch.zeros((N,), dtype=f32, device=dev)]
    for e0 in range(0, E, eg):
        ee = min(eg, E - e0)
        ll = torch.empty(2, dtype=torch.float64, device=dev)
        out = [torch.empty_like(t) for t in acc]
        nbytes = lib.gpz_poisson_nsf_workspace_bytes(N, D, Lt, ee)
        ws = _workspace(dev, nbytes)
        rc = lib.gpz_poisson_nsf(_ptr(mean), _ptr(scale), _ptr(eps[e0:e0 + ee]), _ptr(W_pos), _ptr(V_pos), _ptr(y), N, D,
                                 Lt, ee, int(with_lgamma and e0 == 0), _ptr(ll), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]),
                                 _ptr(out[3]), _ptr(ws), ws.numel(), _stream(dev))
        _lib.check(rc, "gpz_poisson_nsf")
        wgt = ee / E
        total += wgt * ll[0]
        if with_lgamma and e0 == 0:
            total -= ll[1]          # the parameter-free lgamma(y+1) term is evaluated once
        for a_, o_ in zip(acc, out):
            a_.add_(o_, alpha=wgt)
    return (total, *acc)


def profile_enable(on: bool = True):
    _lib.check(_lib.load().gpz_profile_enable(int(on)), "gpz_profile_enable")


def profile_read() -> dict:
    n = len(_lib.PROF_SLOTS)
    ms = (C.c_double * n)()
    cnt = (C.c_int32 * n)()
    _lib.check(_lib.load().gpz_profile_read(ms, cnt, n), "gpz_profile_read")
    return {name: (ms[i], cnt[i]) for i, name in enumerate(_lib.PROF_SLOTS) if not name.startswith("_")}
