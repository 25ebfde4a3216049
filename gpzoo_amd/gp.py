"""Sparse variational GP modules with the gpzoo.gp class API on the HIP hot path.

``SVGP`` / ``WSVGP`` / ``MGGP_SVGP`` / ``MGGP_WSVGP`` keep the reference's
constructor signatures, attribute names (``kernel, jitter, Z, Lu, mu, groupsZ,
constraint`` -- the state_dict keys) and return contract
``(qF: Normal, qU: MultivariateNormal, pU: MultivariateNormal | None)``; shapes
of ``mu`` / ``Lu`` / ``Z`` are read at call time because notebooks replace those
parameters after construction.  ``forward`` is ONE call into
``gpz_svgp_forward`` (csrc/svgp.hip): covariance fill, Cholesky, the triangular
solves and the q(F) reductions never round-trip through torch ops, and neither
Kzx nor W is materialised for more than one N-chunk.

Training: ``qF`` (and, un-whitened, ``pU.scale_tril``) is differentiable w.r.t. ``mu``, ``Lu``, ``Z``,
``sigma``, ``lengthscale`` and ``group_diff_param`` through ``gpz_svgp_backward`` (SURVEY.md §8f
"next" #1); parameters with ``requires_grad=False`` cost nothing.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import distributions
from torch.distributions import constraints

from . import ops
from .kernels import _like, kernel_spec


class _QFMoments(torch.autograd.Function):
    """(mean, scale) of q(F) -- and, un-whitened, chol(Kzz) and the per-latent KL(qU || pU) -- as a
    differentiable function of mu, the raw Lu, Z and the kernel hyper-parameters.
    forward = gpz_svgp_forward, backward = gpz_svgp_backward (SURVEY.md §8f "next" #1)."""

    @staticmethod
    def forward(ctx, mu, Lu_raw, Z, sigma, lengthscale, group_param, call):
        out = call["forward"](mu, Lu_raw)
        ctx.call = call
        ctx.save_for_backward(mu, Lu_raw, out["scale"], Z, sigma, lengthscale,
                              group_param if group_param is not None else mu.new_empty(0))
        ctx.has_group = group_param is not None
        ctx.mark_non_differentiable(out["Lu"])
        chol = out.get("chol")
        if chol is None:
            chol = out["Lu"].new_empty(0)
            ctx.mark_non_differentiable(chol)
        return out["mean"], out["scale"], out["Lu"], chol, out["kl"].to(out["mean"].dtype)

    @staticmethod
    def backward(ctx, g_mean, g_scale, _g_lu, _g_chol, g_kl):
        mu, Lu_raw, scale, Z, sigma, lengthscale, group_param = ctx.saved_tensors
        if g_mean is None:
            g_mean = torch.zeros_like(scale)
        if g_scale is None:
            g_scale = torch.zeros_like(scale)
        need_kernel = any(ctx.needs_input_grad[2:6])
        res = ctx.call["backward"](mu, Lu_raw, g_mean, g_scale, scale, need_kernel,
                                   _g_chol if (need_kernel and _g_chol is not None and _g_chol.numel()) else None, g_kl)
        grads = [res[0].reshape(mu.shape), res[1].reshape(Lu_raw.shape), None, None, None, None, None]
        if need_kernel:
            gth, gZ = res[2], res[3]
            grads[2] = gZ.to(Z.dtype)
            grads[3] = _like(gth[:, 0], sigma)
            grads[4] = _like(gth[:, 1], lengthscale)
            if ctx.has_group:
                grads[5] = _like(gth[:, 2], group_param) * ctx.call["group_chain"]
        return tuple(grads)


class _PrecomputedMoments(torch.autograd.Function):
    """(mean, scale) of WSVGP.forward_precomputed as a differentiable function of mu, the raw Lu and sigma
    (reference gp.py:308-322 under autograd); W is the caller's constant."""

    @staticmethod
    def forward(ctx, W, sigma, mu, Lu_raw):
        out = ops.wsvgp_precomputed(W, sigma, mu, Lu_raw)
        ctx.save_for_backward(W, sigma, mu, Lu_raw, out["scale"])
        return out["mean"], out["scale"]

    @staticmethod
    def backward(ctx, g_mean, g_scale):
        W, sigma, mu, Lu_raw, scale = ctx.saved_tensors
        g_mean = torch.zeros_like(scale) if g_mean is None else g_mean
        g_scale = torch.zeros_like(scale) if g_scale is None else g_scale
        gmu, gLu, gsig = ops.wsvgp_precomputed_backward(W, sigma, mu, Lu_raw, g_mean, g_scale, scale)
        L = gmu.shape[0]
        if mu.numel() != gmu.numel():          # one q(U) shared by the L rows of W
            gmu, gLu = gmu.sum(0), gLu.sum(0)
        return None, _like(gsig, sigma) if ctx.needs_input_grad[1] else None, gmu.reshape(mu.shape), gLu.reshape(Lu_raw.shape)


class _FusedQU(distributions.MultivariateNormal):
    """q(U) as returned by the modules while training: a MultivariateNormal that remembers the per-latent KL the fused pass
    already evaluated (differentiable through gpz_svgp_backward) and builds its ``scale_tril`` only when somebody asks.

    The reference forms ``Lu = tril(raw, -1) + diag(exp(diag raw))`` in every forward (gp.py:278, torch's
    LowerCholeskyTransform); its training loops then only use q(U) inside the KL term, which the fused pass has already.
    Built eagerly through torch that matrix is three passes over (L, M, M) per step -- 0.8 ms of a 55 ms minibatch step at
    L = 20, M = 3000 -- for a tensor nobody reads; here ``raw`` is kept and the same torch expression runs on first access
    (``scale_tril``, ``rsample``, ``log_prob``, ``covariance_matrix``, a KL against another distribution ...), so every other
    use still differentiates w.r.t. the raw parameter exactly as before."""
    _gpz_kl = None
    _gpz_pair = None

    def __init__(self, loc, scale_tril=None, raw=None, validate_args=False):
        if raw is None:
            super().__init__(loc, scale_tril=scale_tril, validate_args=validate_args)
            return
        # what MultivariateNormal.__init__ derives from (loc, scale_tril), without the matrix
        batch_shape = torch.broadcast_shapes(raw.shape[:-2], loc.shape[:-1])
        self.loc = loc.expand(batch_shape + (-1,))
        self._gpz_raw, self._gpz_tril = raw, None
        distributions.Distribution.__init__(self, batch_shape, self.loc.shape[-1:], validate_args=False)

    @property
    def _unbroadcasted_scale_tril(self):
        t = self.__dict__.get("_gpz_tril")
        if t is None:
            raw = self._gpz_raw
            t = raw.tril(-1) + torch.diag_embed(torch.diagonal(raw, dim1=-2, dim2=-1).exp())
            self.__dict__["_gpz_tril"] = t
        return t

    @_unbroadcasted_scale_tril.setter
    def _unbroadcasted_scale_tril(self, value):      # (the eager constructor path assigns it)
        self.__dict__["_gpz_tril"] = value

    def expand(self, batch_shape, _instance=None):
        """As MultivariateNormal.expand (a subclass with its own __init__ has to say how): the expanded distribution
        is a plain MultivariateNormal over the materialised factor -- the fused KL belongs to this very q(U)."""
        batch_shape = torch.Size(batch_shape)
        tril = self._unbroadcasted_scale_tril
        return distributions.MultivariateNormal(self.loc.expand(batch_shape + self.event_shape),
                                                scale_tril=tril.expand(batch_shape + tril.shape[-2:]),
                                                validate_args=False)


class _FusedPU(distributions.MultivariateNormal):
    _gpz_pair = None


@distributions.kl.register_kl(_FusedQU, _FusedPU)
def _kl_fused(q, p):
    """kl_divergence(qU, pU) (utilities.py:481) without torch's batched triangular solves: the value comes
    from the forward pass, its gradient is folded into the fused backward.  Any other pairing falls back
    to torch's MVN-MVN formula."""
    if q._gpz_kl is not None and q._gpz_pair is not None and q._gpz_pair is p._gpz_pair:
        return q._gpz_kl
    return distributions.kl._kl_multivariatenormal_multivariatenormal(q, p)


class _FusedGP(nn.Module):
    _whitened = True
    _clamp_min = 1e-6
    _mggp = False

    def _init_common(self, kernel, dim, M, jitter):
        self.kernel = kernel
        self.jitter = jitter
        self.Z = nn.Parameter(torch.randn((M, dim)))
        self.Lu = nn.Parameter(torch.randn((M, M)))
        self.mu = nn.Parameter(torch.zeros((M,)))
        self.constraint = constraints.lower_cholesky

    # -- the reference's small helpers ------------------------------------
    def kernel_forward(self, X, Z, **args):
        return self.kernel(X, Z, **args)

    def forward_kernels(self, X, **args):
        """(Kxx diag, Kzx, Kzz) as separate matrices -- API parity with gp.py:252-258 /
        :392-399; ``forward`` itself never materialises them."""
        if self._mggp:
            gX = args['groupsX']
            return (self.kernel(X, X, gX, gX, diag=True), self.kernel(self.Z, X, self.groupsZ, gX),
                    self.kernel(self.Z, self.Z, self.groupsZ, self.groupsZ).contiguous())
        return (self.kernel(X, X, diag=True), self.kernel(self.Z, X), self.kernel(self.Z, self.Z).contiguous())

    # -- fused evaluation --------------------------------------------------
    def _latents(self) -> int:
        return 1 if self.mu.dim() == 1 else int(self.mu.shape[0])

    def _cache_args(self, spec, X) -> dict:
        """Factor cache validated by the CONTENT of everything chol(Kzz) depends on (Z, the kernel tensors,
        the group table, groupsZ, jitter, dtype -- ``ops.FactorCache``): frozen hyper-parameters => one
        factorisation per model, not one per step; any edit, also through ``.data``,
        refactors.  ``gp.cache_factor = False`` restores the reference's recompute-every-call behaviour."""
        if not getattr(self, "cache_factor", True):
            return {}
        return dict(cache=self.__dict__.setdefault("_factor_cache", ops.FactorCache()))

    def _evaluate(self, X, groupsX=None, y=None, noise_sd=None, want_moments=True, want_Lu=True,
                  want_chol=False, chunk=0):
        spec = kernel_spec(self.kernel, X, self._latents())
        if spec.L != self._latents():
            raise ValueError(f"kernel defines {spec.L} latent GPs but mu has shape {tuple(self.mu.shape)}")
        gk = dict(gX=groupsX, gZ=self.groupsZ) if self._mggp else {}
        return spec, ops.svgp_forward(spec, X, self.Z, self.mu, self.Lu, float(self.jitter), self._whitened,
                                      y=y, noise_sd=noise_sd, clamp_min=self._clamp_min, chunk=chunk,
                                      want_moments=want_moments, want_Lu=want_Lu, want_chol=want_chol,
                                      **self._cache_args(spec, X), **gk)

    def _distributions(self, out):
        single = self.mu.dim() == 1
        pick = (lambda t: t[0]) if single else (lambda t: t)
        qF = ops.checked_dist(distributions.Normal, pick(out["mean"]), pick(out["scale"]))
        qU = _FusedQU(self.mu, scale_tril=pick(out["Lu"]), validate_args=False)
        if "kl" in out:
            qU._gpz_kl = pick(out["kl"].to(out["Lu"].dtype))
        pU = None
        if not self._whitened:
            pU = _FusedPU(torch.zeros_like(self.mu), scale_tril=pick(out["chol"]), validate_args=False)
            qU._gpz_pair = pU._gpz_pair = object()    # a token, not the distribution: no reference cycle
        return qF, qU, pU

    def _forward(self, X, groupsX=None, verbose=False):
        if verbose:
            print('gpz_svgp_forward: kernels, cholesky, solves and moments in one fused pass')
        gparam = getattr(self.kernel, "group_diff_param", None)
        trainable = [self.mu, self.Lu, self.Z, self.kernel.sigma, self.kernel.lengthscale] + ([gparam] if gparam is not None else [])
        if not (torch.is_grad_enabled() and any(t.requires_grad for t in trainable)):
            _, out = self._evaluate(X, groupsX, want_chol=not self._whitened)
            return self._distributions(out)
        # The closures below serve this call's backward pass too: they hold PRIVATE copies of everything small the
        # problem is made of (Z, the kernel's tensors, groupsZ -- a few KB), so that whatever happens to the live
        # parameters between this forward and its backward (`.data` edits, another forward / optimiser step) the
        # gradients are those of the problem this forward evaluated.
        spec = ops.freeze_spec(kernel_spec(self.kernel, X, self._latents()))
        gk = dict(gX=groupsX, gZ=self.groupsZ.detach().clone()) if self._mggp else {}
        args = (spec, X, self.Z.detach().clone())
        factor_deps = [self.Z, self.kernel.sigma, self.kernel.lengthscale] + ([gparam] if gparam is not None else [])
        cargs = {} if any(t.requires_grad for t in factor_deps) else self._cache_args(spec, X)
        if cargs and torch.cuda.is_current_stream_capturing():
            cargs = {}          # a captured step cannot stop for the cross-call cache's content check: it factors per replay
        if not cargs:
            # cache_factor = False, or Z / kernel hyper-parameters are being trained (the optimiser changes them every
            # step: a cross-call cache could never hit and its content check costs a host sync per call): nothing is
            # kept ACROSS calls, but the backward pass of this very call needs the same chol(Kzz) / inverse its forward
            # just produced -- hand it over instead of factoring twice per step
            cargs = dict(cache=ops.FactorCache())
        common = dict(clamp_min=self._clamp_min, **cargs, **gk)

        kept = {}

        def fwd(mu, Lu_raw):
            # Wt of every chunk stays in HBM for the backward pass when it fits (a third of the free memory)
            out = ops.svgp_forward(*args, mu, Lu_raw, float(self.jitter), self._whitened,
                                   want_chol=not self._whitened, retain_wt=getattr(self, "retain_wt", 1.0 / 3), **common)
            kept["wt"] = out.pop("wt_cache", None)
            kept["gen"] = out.get("factor_generation")     # which factorisation the shared buffer holds now
            kept["qu"] = out.get("qu_generation")          # ... and whose q(U) operands
            return out

        def bwd(mu, Lu_raw, g_mean, g_scale, scale, need_kernel, g_chol, g_kl):
            # the content check (a host sync) is skipped only while the buffer still holds THIS call's factor: a
            # forward of other inputs in between refactors into the shared buffer and bumps its generation -- the
            # check then fails on content and the backward refactors from its own copies
            trust = kept.get("gen") is not None and common["cache"].generation == kept["gen"]
            # the q(U) operands behind the factor are this call's only while no other forward has run on the buffer
            trust_qu = trust and kept.get("qu") is not None and common["cache"].qu_generation == kept["qu"]
            return ops.svgp_backward(*args, mu, Lu_raw, float(self.jitter), self._whitened, g_mean, g_scale, scale,
                                     kernel_grads=need_kernel, g_chol=g_chol, wt_cache=kept.pop("wt", None),
                                     g_kl=g_kl, trust_cache=trust, trust_qu=trust_qu, **common)

        call = dict(forward=fwd, backward=bwd)
        if gparam is not None:
            call["group_chain"] = self.kernel._group_a_chain()
        mean, scale, _, chol, kl = _QFMoments.apply(self.mu, self.Lu, self.Z, self.kernel.sigma,
                                                    self.kernel.lengthscale, gparam, call)
        single = self.mu.dim() == 1
        pick = (lambda t: t[0]) if single else (lambda t: t)
        qF = ops.checked_dist(distributions.Normal, pick(mean), pick(scale))
        # q(U)'s scale_tril is the torch expression of the raw parameter (so that other uses of it differentiate w.r.t.
        # it), evaluated when first asked for: the training loops never do (_FusedQU)
        qU = _FusedQU(self.mu, raw=self.Lu)
        qU._gpz_kl = pick(kl)          # whitened: read by the training loops' KL term (utilities._kl_u)
        if self._whitened:
            return qF, qU, None
        # un-whitened: kl_divergence(qU, pU) resolves to the KL the fused pass already holds (_kl_fused)
        pU = _FusedPU(torch.zeros_like(self.mu), scale_tril=pick(chol), validate_args=False)
        qU._gpz_pair = pU._gpz_pair = object()    # a token, not the distribution: no reference cycle
        return qF, qU, pU

    def elbo(self, X, y, noise_sd, groupsX=None, chunk=0):
        """Closed-form Gaussian ELBO (mggp_test_exact.ipynb:157-159 / utilities.py:479-481
        with an exact likelihood): fp64 device scalar, plus per-latent KL and log-lik terms."""
        _, out = self._evaluate(X, groupsX, y=y, noise_sd=float(noise_sd), want_moments=False, want_Lu=False,
                                chunk=chunk)
        return out["elbo"], out["kl"], out["loglik"]


class _VNNMoments(torch.autograd.Function):
    """(mean, scale, chol, kl) of VNNGP as a differentiable function of mu, the raw Lu, Z, sigma and
    lengthscale: forward = gpz_vnngp_forward, backward = gpz_vnngp_backward.  The neighbour table is a
    constant of the graph, exactly as argsort is in the reference's."""

    @staticmethod
    def forward(ctx, mu, Lu_raw, Z, sigma, lengthscale, call):
        out = call["forward"](mu, Lu_raw)
        ctx.call, ctx.idx = call, out["idx"]
        ctx.save_for_backward(mu, Lu_raw, Z, sigma, lengthscale)
        return out["mean"], out["scale"], out["chol"], out["kl"].to(out["mean"].dtype)

    @staticmethod
    def backward(ctx, g_mean, g_scale, g_chol, g_kl):
        mu, Lu_raw, Z, sigma, lengthscale = ctx.saved_tensors
        need_kernel = any(ctx.needs_input_grad[2:5])
        res = ctx.call["backward"](mu, Lu_raw, ctx.idx, g_mean, g_scale, need_kernel, g_chol if need_kernel else None, g_kl)
        grads = [res[0].reshape(mu.shape), res[1].reshape(Lu_raw.shape), None, None, None, None]
        if need_kernel:
            grads[2] = res[3].to(Z.dtype)
            grads[3] = _like(res[2][:, 0], sigma)
            grads[4] = _like(res[2][:, 1], lengthscale)
        return tuple(grads)


class VNNGP(nn.Module):
    """Nearest-neighbour variational GP; reference gp.py:7-122 (RBF-family kernels: the ones with
    ``return_distance``).  The reference's scalar-``RBF`` path raises (gp.py:83 repeats the neighbour
    table N times instead of L); here a scalar kernel returns ``(N,)`` moments.  Differentiable w.r.t.
    ``mu``, ``Lu``, ``Z``, ``sigma`` and ``lengthscale``; ``kl_divergence(qU, pU)`` on the returned pair
    resolves to the KL the fused pass evaluated (see ``_kl_fused``)."""
    _clamp_min = 5e-2

    def __init__(self, kernel, dim=1, M=50, K=3, jitter=1e-4):
        super().__init__()
        self.kernel = kernel
        self.jitter = jitter
        self.K = K
        self.Z = nn.Parameter(torch.randn((M, dim)))
        self.Lu = nn.Parameter(torch.randn((M, M)))
        self.mu = nn.Parameter(torch.zeros((M,)))
        self.constraint = constraints.lower_cholesky

    def _point_order(self, X):
        """Morton order of the data points for the backward pass's records, kept while X is the same tensor (the notebooks
        train on one X for thousands of steps)."""
        key = (X.data_ptr(), X._version, tuple(X.shape), X.device)
        c = self.__dict__.get("_order_cache")
        if c is None or c[0] != key:
            c = (key, ops.morton_order(X))
            self.__dict__["_order_cache"] = c
        return c[1]

    def forward(self, X, verbose=False):
        nlat = 1 if self.mu.dim() == 1 else int(self.mu.shape[0])
        spec = kernel_spec(self.kernel, X, nlat)
        jitter, K = float(self.jitter), int(self.K)
        pick = (lambda t: t[0]) if self.mu.dim() == 1 else (lambda t: t)
        params = (self.mu, self.Lu, self.Z, self.kernel.sigma, self.kernel.lengthscale)
        if not (torch.is_grad_enabled() and any(t.requires_grad for t in params)):
            out = ops.vnngp_forward(spec, X, self.Z, self.mu, self.Lu, jitter, K, self._clamp_min)
            mean, scale, Lu, chol, kl = out["mean"], out["scale"], out["Lu"], out["chol"], out["kl"].to(out["mean"].dtype)
        else:
            # the closures hold private copies of Z and the kernel's tensors (as _FusedGP._forward does): the backward
            # pass differentiates the problem this forward evaluated and may take its factor, S and KL operands over
            spec = ops.freeze_spec(spec)
            Zc = self.Z.detach().clone()
            kept = {}

            def fwd(mu, Lu_raw):
                out = ops.vnngp_forward(spec, X, Zc, mu, Lu_raw, jitter, K, self._clamp_min, keep_state=True)
                kept["state"] = out.pop("state")
                kept["mu"], kept["Lu"] = mu._version, Lu_raw._version
                return out

            def bwd(mu, Lu_raw, idx, g_mean, g_scale, need_kernel, g_chol, g_kl):
                state = kept.pop("state", None)       # good for one backward pass, and only for the tensors it was made from
                if (mu._version, Lu_raw._version) != (kept.get("mu"), kept.get("Lu")):
                    state = None
                return ops.vnngp_backward(spec, X, Zc, mu, Lu_raw, jitter, K, idx, g_mean, g_scale, clamp_min=self._clamp_min,
                                          kernel_grads=need_kernel, g_chol=g_chol, g_kl=g_kl, state=state,
                                          point_order=self._point_order(X))

            mean, scale, chol, kl = _VNNMoments.apply(*params, dict(forward=fwd, backward=bwd))
            # q(U)'s scale_tril through torch so that other uses of it differentiate w.r.t. the raw parameter
            Lu = self.Lu.tril(-1) + torch.diag_embed(torch.diagonal(self.Lu, dim1=-2, dim2=-1).exp())
            Lu = Lu.reshape(-1, Lu.shape[-2], Lu.shape[-1])
        qF = ops.checked_dist(distributions.Normal, pick(mean), pick(scale))
        # valid by construction: skip the O(L M^2) scale_tril validation
        qU = _FusedQU(self.mu, scale_tril=pick(Lu), validate_args=False)
        pU = _FusedPU(torch.zeros_like(self.mu), scale_tril=pick(chol), validate_args=False)
        qU._gpz_kl = pick(kl)
        qU._gpz_pair = pU._gpz_pair = object()    # a token, not the distribution: no reference cycle
        return qF, qU, pU


class GaussianPrior(nn.Module):
    """Mean-field Normal prior for the non-spatial factors; reference gp.py:125-146 (no kernel work)."""

    def __init__(self, y, L=10):
        super().__init__()
        D, N = y.shape
        self.mean = nn.Parameter(torch.randn(size=(L, N)))
        self.scale = nn.Parameter(torch.rand(size=(L, N)))
        self.scale_pf = 1.0

    def _pair(self, mean, raw_scale):
        qF = ops.checked_dist(distributions.Normal, mean, torch.nn.functional.softplus(raw_scale))
        pF = ops.checked_dist(distributions.Normal, torch.zeros_like(qF.mean), self.scale_pf * torch.ones_like(qF.scale))
        return qF, pF

    def forward(self):
        return self._pair(self.mean, self.scale)

    def forward_batched(self, idx):
        return self._pair(self.mean[:, idx], self.scale[:, idx])


class WSVGP(_FusedGP):
    """Whitened SVGP; reference gp.py:235-322."""
    _whitened = True

    def __init__(self, kernel, dim=1, M=50, jitter=1e-4):
        super().__init__()
        self._init_common(kernel, dim, M, jitter)

    def forward(self, X, verbose=False, **args):
        return self._forward(X, args.get('groupsX'), verbose)

    def forward_precomputed(self, W, **args):
        """q(F) from a caller-supplied W (L,N,M) (gp.py:308-322); differentiable w.r.t. mu, Lu and the kernel's
        sigma like the reference's expression (gpz_wsvgp_precomputed_backward).  W itself is a constant here."""
        trainable = (self.mu, self.Lu, self.kernel.sigma)
        if not (torch.is_grad_enabled() and any(t.requires_grad for t in trainable)):
            return self._distributions(ops.wsvgp_precomputed(W, self.kernel.sigma, self.mu, self.Lu))
        if W.requires_grad:
            raise NotImplementedError("forward_precomputed treats W as a constant: detach it (gradients w.r.t. the kernel "
                                      "hyper-parameters and Z flow through forward())")
        mean, scale = _PrecomputedMoments.apply(W, self.kernel.sigma, self.mu, self.Lu)
        Lu = self.Lu.tril(-1) + torch.diag_embed(torch.diagonal(self.Lu, dim1=-2, dim2=-1).exp())
        pick = (lambda t: t[0]) if self.mu.dim() == 1 else (lambda t: t)
        return ops.checked_dist(distributions.Normal, pick(mean), pick(scale)), _FusedQU(self.mu, scale_tril=Lu, validate_args=False), None


class SVGP(_FusedGP):
    """Un-whitened SVGP; reference gp.py:149-232 (pU = N(0, Kzz) is returned for the MVN-MVN KL)."""
    _whitened = False
    _clamp_min = 1e-6

    def __init__(self, kernel, dim=1, M=50, jitter=1e-4):
        super().__init__()
        self._init_common(kernel, dim, M, jitter)
        self.precompute_distance = False

    def forward_kernels(self, X, Z, **args):
        return super().forward_kernels(X, **args)

    def forward(self, X, verbose=False):
        return self._forward(X, None, verbose)


class MGGP_SVGP(_FusedGP):
    """Multi-group un-whitened SVGP; reference gp.py:329-382 (variance clamp 5e-2, gp.py:378)."""
    _whitened = False
    _clamp_min = 5e-2
    _mggp = True

    def __init__(self, kernel, dim=1, M=50, jitter=1e-4, n_groups=2):
        super().__init__()
        self._init_common(kernel, dim, M, jitter)
        self.groupsZ = nn.Parameter(torch.randint(0, n_groups, (M,)).type(torch.LongTensor), requires_grad=False)

    def forward(self, X, groupsX, verbose=False):
        return self._forward(X, groupsX, verbose)


class MGGP_WSVGP(WSVGP):
    """Multi-group whitened SVGP; reference gp.py:385-399 (groupsX is a required kwarg)."""
    _mggp = True

    def __init__(self, kernel, dim=1, M=50, n_groups=2, jitter=1e-4):
        super().__init__(kernel, dim, M, jitter)
        self.groupsZ = nn.Parameter(torch.randint(0, n_groups, (M,)).type(torch.LongTensor), requires_grad=False)

    def forward(self, X, verbose=False, **args):
        return self._forward(X, args['groupsX'], verbose)
