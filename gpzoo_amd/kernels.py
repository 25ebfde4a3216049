"""Covariance kernels with the gpzoo.kernels class API, evaluated by HIP.

Same class names, constructor signatures, parameter names/shapes (they are
state_dict keys) and ``forward`` contracts as the reference's gpzoo/kernels.py;
each class only flattens its hyper-parameters into a ``KernelSpec`` and hands the
pairwise work to ``gpz_kfill`` (csrc/kfill.hip).  Results are ``(N, M)`` for
scalar parameters and ``(L, N, M)`` for per-latent ones.

Deviations from the reference at HEAD, all documented in SURVEY.md §8a:
  * ``diag=True`` of the vmap kernels returns sigma^2 broadcast to ``(N,)`` /
    ``(L, N)`` (the intended contract, reference kernels.py:26/54; the HEAD
    expression at kernels.py:55/96 is shape-broken);
  * ``NSF_RBF(L=1).forward(diag=True)`` returns ``(1, N)`` instead of raising;
  * squared distances are formed by direct differencing (more accurate than
    cdist's / _squared_dist's matmul expansion in fp32).
``forward`` is differentiable like the reference's traced modules (kernels.py:114-130, 139-155, 176-228):
``kernel(X, Z).sum().backward()`` sends gradients to sigma, lengthscale, group_diff_param, X and Z through
``gpz_kgrad`` (csrc/kgrad.hip), so notebook code that builds its own model from kernel matrices (the
inline ExactGP of exact_mggp.ipynb) trains.  The GP modules do not use this route: their fused
backward (gp.py) differentiates the whole pass.  The distance matrix of ``return_distance=True`` is a
constant (the reference only sorts it, gp.py:64).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import KernelSpec
from .utilities import _embed_distance_matrix


def _sq(v: torch.Tensor, n: int) -> torch.Tensor:
    """sigma^2 broadcast over n points: (n,) for a scalar parameter, (L, n) otherwise (differentiable)."""
    s2 = v.reshape(-1) ** 2
    if v.dim() == 0:
        return s2.expand(n)
    return s2[:, None].expand(-1, n)


def _like(grad_l: torch.Tensor, param: torch.Tensor) -> torch.Tensor:
    """Per-latent gradient (L,) -> the parameter's own shape: (), (L,) or (L,1,1)."""
    if param.numel() == 1:
        return grad_l.sum().reshape(param.shape).to(param.dtype)
    return grad_l.reshape(param.shape).to(param.dtype)


class _KernelMatrix(torch.autograd.Function):
    """K = kernel(A, B) as a differentiable function of A, B, sigma, lengthscale and the group parameter:
    forward = gpz_kfill, backward = gpz_kgrad (one contraction per point set that needs a gradient)."""

    @staticmethod
    def forward(ctx, A, B, sigma, lengthscale, group_param, meta):
        ctx.meta = meta
        ctx.save_for_backward(A, B, sigma, lengthscale, group_param if group_param is not None else A.new_empty(0))
        return ops.kfill(meta["spec"], A, B, gA=meta.get("gA"), gB=meta.get("gB"))

    @staticmethod
    def backward(ctx, gK):
        A, B, sigma, lengthscale, group_param = ctx.saved_tensors
        m = ctx.meta
        spec, gA, gB = m["spec"], m.get("gA"), m.get("gB")
        gK3 = gK.reshape(spec.L, A.shape[0], B.shape[0])
        need_A, need_B = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gth, gpa = ops.kgrad(spec, A, B, gK3, gA, gB, want_points=need_A)
        gpb = None
        if need_B:
            _, gpb = ops.kgrad(spec, B.to(A.dtype), A, gK3.transpose(-1, -2), gB, gA)
        grads = [gpa.to(A.dtype) if need_A else None, gpb.to(B.dtype) if need_B else None,
                 _like(gth[:, 0], sigma) if ctx.needs_input_grad[2] else None,
                 _like(gth[:, 1], lengthscale) if ctx.needs_input_grad[3] else None, None, None]
        if m.get("chain") is not None and ctx.needs_input_grad[4]:
            grads[4] = _like(gth[:, 2], group_param) * m["chain"]
        return tuple(grads)


def _kernel_matrix(spec: KernelSpec, A, B, sigma, lengthscale, group_param=None, chain=None, gA=None, gB=None):
    ts = [A, B, sigma, lengthscale] + ([group_param] if group_param is not None else [])
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        return _KernelMatrix.apply(A, B, sigma, lengthscale, group_param, dict(spec=spec, gA=gA, gB=gB, chain=chain))
    return ops.kfill(spec, A, B, gA=gA, gB=gB)


class _HipKernel(nn.Module):
    _kind = _lib.KERNEL_RBF

    def _check_covariance(self):
        """The vmap kernels of the reference evaluate ``self.covariance`` (kernels.py:14-20, 42-47, 75-91), which a
        user may override; only the shipped closed forms have HIP kernels, so an overridden one is refused instead
        of being silently replaced by the built-in formula."""
        mine = getattr(type(self), "covariance", None)
        if mine is None:
            return
        shipped = any(mine is getattr(c, "covariance", None) for c in (batched_RBF, batched_Matern32, batched_MGGP_RBF))
        if not shipped or "covariance" in self.__dict__:
            raise NotImplementedError(
                f"{type(self).__name__}.covariance is user-defined: gpzoo_amd evaluates the closed-form RBF / Matern-3/2 / "
                "multi-group RBF covariances in HIP and has no generic (vmap) path; use the reference package for custom kernels")

    def _spec(self, n_latent: int | None = None) -> KernelSpec:
        sig, ell = self.sigma.detach().reshape(-1), self.lengthscale.detach().reshape(-1)
        batched = self.sigma.dim() > 0 or self.lengthscale.dim() > 0
        L = max(sig.numel(), ell.numel(), n_latent or 1)
        return KernelSpec(self._kind, sig.expand(L), ell.expand(L), batched or (n_latent or 1) > 1)

    def forward(self, X, Z, diag=False, return_distance=False):
        self._check_covariance()
        if diag:
            return _sq(self.sigma, X.size(0))
        K = _kernel_matrix(self._spec(), X, Z, self.sigma, self.lengthscale)
        if return_distance:
            return K, ops.pairwise_distance(X, Z)
        return K


class RBF(_HipKernel):
    """sigma^2 exp(-d^2 / (2 l^2)); reference kernels.py:106-130."""

    def __init__(self, sigma=1.0, lengthscale=2.0):
        super().__init__()
        self.sigma = nn.Parameter(torch.tensor(sigma))
        self.lengthscale = nn.Parameter(torch.tensor(lengthscale))
        self.input_dim = 2

    def forward_distance(self, distance_squared):
        # element-wise on a caller-supplied tensor (kernels.py:128-130); not on the fused path
        return (self.sigma ** 2) * torch.exp(-0.5 * distance_squared / (self.lengthscale ** 2))


class NSF_RBF(RBF):
    """L independent RBFs with (L,1,1) parameters; reference kernels.py:133-155."""

    def __init__(self, sigma=1.0, lengthscale=2.0, L=10):
        super().__init__(sigma=sigma, lengthscale=lengthscale)
        self.L = L
        self.sigma = nn.Parameter(sigma * torch.ones((L, 1, 1)))
        self.lengthscale = nn.Parameter(lengthscale * torch.ones((L, 1, 1)))


class batched_RBF(_HipKernel):
    """vmap-style RBF: scalar or length-L vector parameters; reference kernels.py:34-59."""

    def __init__(self, sigma=1.0, lengthscale=2.0):
        super().__init__()
        self.sigma = nn.Parameter(torch.tensor(sigma))
        self.lengthscale = nn.Parameter(torch.tensor(lengthscale))

    def covariance(self, x1, x2):
        d2 = ((x1 - x2) ** 2).sum()
        return (self.sigma ** 2) * torch.exp(-0.5 * d2 / (self.lengthscale ** 2))

    def forward(self, X, Z, diag=False):
        return super().forward(X, Z, diag=diag)


class batched_Matern32(_HipKernel):
    """sigma^2 (1 + sqrt3 r / l) exp(-sqrt3 r / l); reference kernels.py:6-30."""
    _kind = _lib.KERNEL_MATERN32

    def __init__(self, sigma=1.0, lengthscale=2.0):
        super().__init__()
        self.sigma = nn.Parameter(torch.tensor(sigma))
        self.lengthscale = nn.Parameter(torch.tensor(lengthscale))

    def covariance(self, x1, x2):
        v = (3 ** 0.5) * torch.sqrt(((x1 - x2) ** 2).sum()) / self.lengthscale
        return (self.sigma ** 2) * (1 + v) * torch.exp(-v)

    def forward(self, X, Z, diag=False):
        return super().forward(X, Z, diag=diag)


class _MGGPMixin:
    """Multi-group RBF: sigma^2 exp(-d^2 / (2 l^2 den)) den^(-p/2), den = a' r_g^2 + 1,
    r_g^2 the squared distance between group embeddings (bit-exact int64 gather)."""
    _kind = _lib.KERNEL_MGGP_RBF

    def _group_a(self) -> torch.Tensor:
        raise NotImplementedError

    def _group_a_chain(self) -> torch.Tensor:
        """d(effective multiplier)/d(group_diff_param), element-wise."""
        raise NotImplementedError

    def _group_pow(self, X) -> float:
        return 0.5 * self.input_dim

    def _mggp_spec(self, X, n_latent=None) -> KernelSpec:
        base = _HipKernel._spec(self, n_latent)
        emb = self.embedding.detach().to(device=X.device, dtype=X.dtype)
        r2 = ((emb[:, None, :] - emb[None, :, :]) ** 2).sum(-1)   # (G,G) table, G <= a few dozen
        a = self._group_a().detach().reshape(-1)
        batched = base.batched or self.group_diff_param.dim() > 0
        L = max(base.L, a.numel())
        return KernelSpec(self._kind, base.sigma.expand(L), base.lengthscale.expand(L), batched,
                          a.expand(L), r2, self._group_pow(X))

    def _mggp_forward(self, X, Z, groupsX, groupsZ, diag=False):
        _HipKernel._check_covariance(self)
        if diag:
            return _sq(self.sigma, X.size(0))
        return _kernel_matrix(self._mggp_spec(X), X, Z, self.sigma, self.lengthscale, self.group_diff_param,
                              self._group_a_chain(), gA=groupsX, gB=groupsZ)


class MGGP_RBF(_MGGPMixin, RBF):
    """Scalar-parameter multi-group RBF, a un-squared; reference kernels.py:158-191."""

    def __init__(self, sigma=1.0, lengthscale=2.0, group_diff_param=1.0, n_groups=2, device='cpu'):
        RBF.__init__(self, sigma, lengthscale)
        self.group_diff_param = nn.Parameter(torch.tensor(group_diff_param))
        group_distances = torch.ones(n_groups) - torch.eye(n_groups)
        self.embedding = _embed_distance_matrix(group_distances).to(device)  # plain tensor, as in the reference

    def set_group_distances(self, group_distances):
        self.embedding = _embed_distance_matrix(group_distances)

    def _group_a(self):
        return self.group_diff_param

    def _group_a_chain(self):
        return torch.ones_like(self.group_diff_param)

    def forward(self, X, Z, groupsX, groupsZ, diag=False):
        return self._mggp_forward(X, Z, groupsX, groupsZ, diag)


class MGGP_NSF_RBF(_MGGPMixin, NSF_RBF):
    """Per-latent multi-group RBF, a squared; reference kernels.py:194-228."""

    def __init__(self, sigma=1.0, lengthscale=2.0, group_diff_param=1.0, n_groups=2, L=10, device='cpu'):
        NSF_RBF.__init__(self, sigma, lengthscale, L)
        self.group_diff_param = nn.Parameter(group_diff_param * torch.ones((L, 1, 1)))
        group_distances = torch.ones(n_groups) - torch.eye(n_groups)
        self.embedding = nn.Parameter(_embed_distance_matrix(group_distances), requires_grad=False)

    def set_group_distances(self, group_distances):
        self.embedding = nn.Parameter(_embed_distance_matrix(group_distances), requires_grad=False)

    def _group_a(self):
        return torch.square(self.group_diff_param)

    def _group_a_chain(self):
        return 2 * self.group_diff_param.detach()

    def forward(self, X, Z, groupsX, groupsZ, diag=False):
        return self._mggp_forward(X, Z, groupsX, groupsZ, diag)


class batched_MGGP_RBF(_MGGPMixin, batched_RBF):
    """vmap-style multi-group RBF, |a| and p = true input dim; reference kernels.py:62-104."""

    def __init__(self, sigma=1.0, lengthscale=1.0, group_diff_param=1.0, n_groups=10):
        batched_RBF.__init__(self, sigma, lengthscale)
        self.group_diff_param = nn.Parameter(torch.tensor(group_diff_param))
        group_distances = torch.ones(n_groups) - torch.eye(n_groups)
        self.embedding = nn.Parameter(_embed_distance_matrix(group_distances), requires_grad=False)

    def set_group_distances(self, group_distances):
        self.embedding = nn.Parameter(_embed_distance_matrix(group_distances), requires_grad=False)

    def covariance(self, x1, x2, group_embedding1, group_embedding2):
        d2 = ((x1 - x2) ** 2).sum() / (self.lengthscale ** 2)
        p = x1.unsqueeze(0).shape[-1]
        val = 1 / (torch.abs(self.group_diff_param) * torch.sum((group_embedding1 - group_embedding2) ** 2) + 1)
        return (self.sigma ** 2) * torch.exp(-0.5 * d2 * val) * (val ** (0.5 * p))

    def _group_a(self):
        return torch.abs(self.group_diff_param)

    def _group_a_chain(self):
        return torch.sign(self.group_diff_param.detach())

    def _group_pow(self, X) -> float:
        return 0.5 * X.shape[-1]

    def forward(self, X, Z, groupsX, groupsZ, diag=False):
        return self._mggp_forward(X, Z, groupsX, groupsZ, diag)


def kernel_spec(kernel: nn.Module, X: torch.Tensor, n_latent: int | None = None) -> KernelSpec:
    """KernelSpec of any kernel object above (what the GP classes pass to the fused forward)."""
    if isinstance(kernel, _HipKernel):
        kernel._check_covariance()
    if isinstance(kernel, _MGGPMixin):
        return kernel._mggp_spec(X, n_latent)
    if isinstance(kernel, _HipKernel):
        return kernel._spec(n_latent)
    raise TypeError(f"{type(kernel).__name__} has no HIP implementation: gpzoo_amd supports the closed-form "
                    "kernels of gpzoo.kernels (RBF, Matern-3/2 and multi-group RBF families)")
