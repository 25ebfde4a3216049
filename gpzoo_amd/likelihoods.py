"""Gaussian likelihood wrappers with the gpzoo.likelihoods class API.

Thin consumers of the hot path's outputs (they stay torch):
``forward`` calls the GP once -- one fused HIP pass -- and wraps the result in
torch distributions exactly like reference likelihoods.py:7-36.  The Poisson
factor models (SURVEY §8f "next" #2) follow below: same classes and return tuples
as the reference, plus ``expected_loglik`` -- the fused fp32 training-step form
(``gpz_poisson_nsf``; fp64 models are evaluated in fp32 there and cast back).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import distributions


def _dist(cls, *args, **kwargs):
    """A torch distribution whose argument validation (one host sync per constrained argument) moves to the end of an
    enclosing ``ops.deferred_info()`` block -- the training loops' steps; the plain constructor elsewhere."""
    from .ops import checked_dist
    return checked_dist(cls, *args, **kwargs)


class GaussianLikelihood(nn.Module):
    """pY = Normal(F, softplus(noise)) with F ~ qF.rsample((E,)); reference likelihoods.py:7-20."""

    def __init__(self, gp, noise=0.1):
        super().__init__()
        self.gp = gp
        self.noise = nn.Parameter(torch.tensor(noise))

    def forward(self, X, E=1, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X, verbose=verbose, **kwargs)
        F = qF.rsample((E,))
        pY = _dist(distributions.Normal, F, torch.nn.functional.softplus(self.noise))
        return pY, qF, qU, pU


class ExactLikelihood(nn.Module):
    """pY = Normal(qF.mean, softplus(noise)); reference likelihoods.py:23-36.  ``elbo`` evaluates
    the closed-form objective of mggp_test_exact.ipynb:157-159 in the same fused pass."""

    def __init__(self, gp, noise=0.1):
        super().__init__()
        self.gp = gp
        self.noise = nn.Parameter(torch.tensor(noise))

    def forward(self, X, E=1, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X, verbose=verbose, **kwargs)
        pY = _dist(distributions.Normal, qF.mean, torch.nn.functional.softplus(self.noise))
        return pY, qF, qU, pU

    def elbo(self, X, y, **kwargs):
        sd = float(torch.nn.functional.softplus(self.noise.detach()))
        return self.gp.elbo(X, y, sd, groupsX=kwargs.get('groupsX'))[0]


# --------------------------------------------------------------------------------------------
# Poisson factor models (SURVEY.md §8f "next" #2).  The classes keep the reference's names,
# constructor signatures, parameter names and return tuples (pY is a real torch Poisson over the
# (E,D,N) rate, because callers use pY.log_prob / pY.rate); `expected_loglik` is the fused
# MI355X path for the training step: same number as pY.log_prob(y).mean(0).sum(), same gradients,
# without materialising the rate.
# --------------------------------------------------------------------------------------------

class _PoissonLogLik(torch.autograd.Function):
    """(1/E) sum_e sum_dn log Poisson(y | V Z_e) through gpz_poisson_nsf (forward and gradients in
    one fused pass; backward only scales them)."""

    @staticmethod
    def forward(ctx, mean, scale, W_pos, V_pos, eps, y, with_lgamma):
        from . import ops
        ll, dmean, dscale, dW, dV = ops.poisson_nsf(mean, scale, eps, W_pos, V_pos, y, with_lgamma)
        ctx.save_for_backward(dmean.to(mean.dtype), dscale.to(scale.dtype), dW.to(W_pos.dtype), dV.to(V_pos.dtype))
        return ll.to(mean.dtype)

    @staticmethod
    def backward(ctx, g):
        dmean, dscale, dW, dV = ctx.saved_tensors
        return g * dmean, g * dscale, g * dW, g * dV, None, None, None


def poisson_expected_loglik(qF_list, W_list, V_pos, y, E=10, with_lgamma=True, eps=None):
    """Fused Monte-Carlo E_q[log p(y | F)] for rate = V * sum_k W_k exp(F_k), F_k ~ qF_k.

    qF_list: Normal distributions over (L_k, N) factors; W_list: positive (D, L_k) loadings."""
    mean = torch.cat([q.mean for q in qF_list], dim=0)
    scale = torch.cat([q.scale for q in qF_list], dim=0)
    W = torch.cat(list(W_list), dim=1)
    if eps is None:   # one draw per factor set, in order, through the function Normal.rsample itself uses
        import torch.distributions.normal as tdn
        eps = torch.cat([tdn._standard_normal((E,) + tuple(q.mean.shape), dtype=mean.dtype, device=mean.device)
                         for q in qF_list], dim=1)
    return _PoissonLogLik.apply(mean, scale, W, V_pos, eps, y, with_lgamma)


class PoissonFactorization(nn.Module):
    """softplus(W) @ exp(F): base of PNMF / NSF2 / the hybrids; reference likelihoods.py:39-53."""

    def __init__(self, prior, y, L=10):
        super().__init__()
        D, N = y.shape
        self.prior = prior
        self.W = nn.Parameter(torch.rand((D, L)))

    def get_rate(self, prior_samples):
        return torch.matmul(torch.nn.functional.softplus(self.W), torch.exp(prior_samples))   # (E, D, N)


class PNMF(PoissonFactorization):
    """Non-spatial Poisson NMF over a GaussianPrior; reference likelihoods.py:56-72."""

    def __init__(self, prior, y, L=10):
        super().__init__(prior=prior, y=y, L=L)
        D, N = y.shape
        self.V = nn.Parameter(torch.ones((N,)))
        self.X = nn.Parameter(torch.zeros((N, 2)), requires_grad=False)

    def forward(self, E=10, **kwargs):
        qF, pF = self.prior()
        Z = self.get_rate(qF.rsample((E,)))
        pY = _dist(distributions.Poisson, torch.nn.functional.softplus(self.V) * Z)
        return pY, qF, pF


class NSF2(PoissonFactorization):
    """Non-negative spatial factorisation over an SVGP prior; reference likelihoods.py:74-97."""

    def __init__(self, gp, y, L=10):
        super().__init__(prior=gp, y=y, L=L)
        D, N = y.shape
        self.V = nn.Parameter(torch.ones((N,)))

    def forward(self, X, E=10, verbose=False, **kwargs):
        qF, qU, pU = self.prior(X=X, verbose=verbose, **kwargs)
        Z = self.get_rate(qF.rsample((E,)))
        pY = _dist(distributions.Poisson, torch.nn.functional.softplus(self.V) * Z)
        return pY, qF, qU, pU

    def forward_batched(self, X, idx, E=10, verbose=False, **kwargs):
        qF, qU, pU = self.prior(X=X[idx], verbose=verbose, **kwargs)
        Z = self.get_rate(qF.rsample((E,)))
        pY = _dist(distributions.Poisson, torch.nn.functional.softplus(self.V[idx]) * Z)
        return pY, qF, qU, pU

    def expected_loglik(self, X, y, idx=None, E=10, eps=None, with_lgamma=True, **kwargs):
        """Fused training-step form: (E_q[log p(y | F)] as a scalar, qF, qU, pU)."""
        Xb = X if idx is None else X[idx]
        V = self.V if idx is None else self.V[idx]
        qF, qU, pU = self.prior(X=Xb, **kwargs)
        ll = poisson_expected_loglik([qF], [torch.nn.functional.softplus(self.W)],
                                     torch.nn.functional.softplus(V), y, E=E, with_lgamma=with_lgamma, eps=eps)
        return ll, qF, qU, pU


class NSF(nn.Module):
    """Same model with W, V held directly; reference likelihoods.py:227-268."""

    def __init__(self, gp, y, L=10):
        super().__init__()
        D, N = y.shape
        self.gp = gp
        self.W = nn.Parameter(torch.rand((D, L)))
        self.V = nn.Parameter(torch.ones((N,)))

    def _rate(self, qF, V, E):
        F = torch.exp(qF.rsample((E,)))
        return V * torch.matmul(torch.nn.functional.softplus(self.W), F)

    def forward(self, X, E=10, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X=X, verbose=verbose, **kwargs)
        return _dist(distributions.Poisson, self._rate(qF, torch.nn.functional.softplus(self.V), E)), qF, qU, pU

    def forward_batched(self, X, idx, E=10, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X=X[idx], verbose=verbose, **kwargs)
        return _dist(distributions.Poisson, self._rate(qF, torch.nn.functional.softplus(self.V)[idx], E)), qF, qU, pU

    def expected_loglik(self, X, y, idx=None, E=10, eps=None, with_lgamma=True, **kwargs):
        Xb = X if idx is None else X[idx]
        V = torch.nn.functional.softplus(self.V)
        qF, qU, pU = self.gp(X=Xb, **kwargs)
        ll = poisson_expected_loglik([qF], [torch.nn.functional.softplus(self.W)], V if idx is None else V[idx], y,
                                     E=E, with_lgamma=with_lgamma, eps=eps)
        return ll, qF, qU, pU


class MGGP_NSF(NSF):
    """NSF over a multi-group GP (groupsX is positional); reference likelihoods.py:341-374."""

    def _gp(self, X, groupsX, verbose):
        if getattr(self.gp, "_whitened", False):
            return self.gp(X, verbose=verbose, groupsX=groupsX)     # MGGP_WSVGP takes groupsX as a keyword
        return self.gp(X, groupsX, verbose)

    def forward(self, X, groupsX, E=10, verbose=False):
        qF, qU, pU = self._gp(X, groupsX, verbose)
        return _dist(distributions.Poisson, self._rate(qF, torch.nn.functional.softplus(self.V), E)), qF, qU, pU

    def forward_batched(self, X, groupsX, idx, E=10, verbose=False):
        qF, qU, pU = self._gp(X[idx], groupsX[idx], verbose)
        return _dist(distributions.Poisson, self._rate(qF, torch.nn.functional.softplus(self.V)[idx], E)), qF, qU, pU

    def expected_loglik(self, X, y, groupsX=None, idx=None, E=10, eps=None, with_lgamma=True, **kwargs):
        """Fused training-step form; the group ids follow the sampled spots (``groupsX[idx]``, as in
        ``forward_batched`` / reference likelihoods.py:364-366)."""
        if groupsX is None:
            raise TypeError("MGGP_NSF.expected_loglik needs groupsX")
        Xb, gb = (X, groupsX) if idx is None else (X[idx], groupsX[idx])
        V = torch.nn.functional.softplus(self.V)
        qF, qU, pU = self._gp(Xb, gb, False)
        ll = poisson_expected_loglik([qF], [torch.nn.functional.softplus(self.W)], V if idx is None else V[idx], y,
                                     E=E, with_lgamma=with_lgamma, eps=eps)
        return ll, qF, qU, pU


class Hybrid_NSF2(nn.Module):
    """Spatial (GP) + non-spatial (GaussianPrior) factors; reference likelihoods.py:100-164."""

    def __init__(self, gp, prior, y, L=10, T=10):
        super().__init__()
        D, N = y.shape
        self.sf = PoissonFactorization(prior=gp, y=y, L=L)
        self.cf = PoissonFactorization(prior=prior, y=y, L=T)
        self.V = nn.Parameter(torch.ones((N,)))

    def _pY(self, qF1, qF2, V, E):
        Z = self.sf.get_rate(qF1.rsample((E,))) + self.cf.get_rate(qF2.rsample((E,)))
        return _dist(distributions.Poisson, V * Z)

    def forward(self, X, E=10, verbose=False, **kwargs):
        qF1, qU, pU = self.sf.prior(X=X, verbose=verbose, **kwargs)
        qF2, pF2 = self.cf.prior()
        return self._pY(qF1, qF2, torch.nn.functional.softplus(self.V), E), qF1, qU, pU, qF2, pF2

    def forward_batched(self, X, idx, E=10, verbose=False, **kwargs):
        qF1, qU, pU = self.sf.prior(X=X[idx], verbose=verbose, **kwargs)
        qF2, pF2 = self.cf.prior.forward_batched(idx)
        return self._pY(qF1, qF2, torch.nn.functional.softplus(self.V[idx]), E), qF1, qU, pU, qF2, pF2

    def forward_precomputed(self, W, idx, E=10, verbose=False, **kwargs):
        qF1, qU, pU = self.sf.prior.forward_precomputed(W, verbose=verbose, **kwargs)
        qF2, pF2 = self.cf.prior.forward_batched(idx)
        return self._pY(qF1, qF2, torch.nn.functional.softplus(self.V[idx]), E), qF1, qU, pU, qF2, pF2

    def expected_loglik(self, X, y, idx=None, E=10, eps=None, with_lgamma=True, **kwargs):
        Xb = X if idx is None else X[idx]
        V = self.V if idx is None else self.V[idx]
        qF1, qU, pU = self.sf.prior(X=Xb, **kwargs)
        qF2, pF2 = self.cf.prior() if idx is None else self.cf.prior.forward_batched(idx)
        sp = torch.nn.functional.softplus
        ll = poisson_expected_loglik([qF1, qF2], [sp(self.sf.W), sp(self.cf.W)], sp(V), y, E=E,
                                     with_lgamma=with_lgamma, eps=eps)
        return ll, qF1, qU, pU, qF2, pF2


class Hybrid_NSF_Exact(Hybrid_NSF2):
    """Hybrid model with the log-normal mean exp(m + s^2/2) in place of sampling; reference
    likelihoods.py:167-222.  Nothing is sampled, so the rate has NO sample axis: ``pY.rate`` is (D,N) and ``E`` is
    ignored, as in the reference."""

    def _pY(self, qF1, qF2, V, E):
        Z = self.sf.get_rate(qF1.mean + 0.5 * qF1.scale ** 2) + self.cf.get_rate(qF2.mean + 0.5 * qF2.scale ** 2)
        return _dist(distributions.Poisson, V * Z)

    def expected_loglik(self, X, y, idx=None, E=10, eps=None, with_lgamma=True, **kwargs):
        """The closed-form objective of THIS class as the reference's loops evaluate it -- not the sampled one of
        Hybrid_NSF2.  With a (D,N) rate, ``pY.log_prob(y).mean(axis=0).sum()`` (utilities.py:537) and
        ``(y log r - r).mean(axis=0).sum()`` (utilities.py:508-510) average over the GENE axis: the value is
        (1/D) sum_dn log Poisson(y | V (W1 exp(m1 + s1^2/2) + W2 exp(m2 + s2^2/2))).  Evaluated by the fused kernel as
        one noise-free "sample" of F = m + s^2/2 (gpz_poisson_nsf; the chain to m and s runs through torch);
        ``E`` and ``eps`` are accepted and unused."""
        Xb = X if idx is None else X[idx]
        V = self.V if idx is None else self.V[idx]
        qF1, qU, pU = self.sf.prior(X=Xb, **kwargs)
        qF2, pF2 = self.cf.prior() if idx is None else self.cf.prior.forward_batched(idx)
        sp = torch.nn.functional.softplus
        m = torch.cat([qF1.mean + 0.5 * qF1.scale ** 2, qF2.mean + 0.5 * qF2.scale ** 2], dim=0)
        zero = torch.zeros_like(m)
        ll = _PoissonLogLik.apply(m, zero, torch.cat([sp(self.sf.W), sp(self.cf.W)], dim=1), sp(V), zero[None], y,
                                  with_lgamma)
        return ll / y.shape[0], qF1, qU, pU, qF2, pF2


class Hybrid_NSF(NSF):
    """NSF plus mean-field non-spatial factors held in the module (raw, un-soft-plussed loadings);
    reference likelihoods.py:271-338."""

    def __init__(self, gp, y, L=10, non_spatial_factors=10):
        super().__init__(gp=gp, y=y, L=L)
        D, N = y.shape
        self.W2 = nn.Parameter(torch.rand((D, non_spatial_factors)))
        self.mF = nn.Parameter(torch.zeros((non_spatial_factors, N)))
        self.scale_qF = nn.Parameter(1e-1 * torch.rand((non_spatial_factors, N)))

    def _hybrid(self, qF, mF, raw_scale, V, E):
        scale2 = torch.nn.functional.softplus(raw_scale)
        qF2 = _dist(distributions.Normal, mF, scale2)
        F = torch.exp(torch.cat((qF.rsample((E,)), qF2.rsample((E,))), dim=1))
        Z = torch.matmul(torch.cat((self.W, self.W2), dim=1), F)
        pF2 = _dist(distributions.Normal, torch.zeros_like(mF), torch.ones_like(scale2))
        return _dist(distributions.Poisson, V * Z), qF2, pF2

    def forward(self, X, E=10, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X=X, verbose=verbose, **kwargs)
        pY, qF2, pF2 = self._hybrid(qF, self.mF, self.scale_qF, torch.nn.functional.softplus(self.V), E)
        return pY, qF, qU, pU, qF2, pF2

    def forward_batched(self, X, idx, E=10, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X=X[idx], verbose=verbose, **kwargs)
        pY, qF2, pF2 = self._hybrid(qF, self.mF[:, idx], self.scale_qF[:, idx],
                                    torch.nn.functional.softplus(self.V)[idx], E)
        return pY, qF, qU, pU, qF2, pF2

    def expected_loglik(self, X, y, idx=None, E=10, eps=None, with_lgamma=True, **kwargs):
        """Fused training-step form over both factor sets (raw loadings W, W2 as in ``_hybrid``)."""
        Xb = X if idx is None else X[idx]
        mF, rs = (self.mF, self.scale_qF) if idx is None else (self.mF[:, idx], self.scale_qF[:, idx])
        V = torch.nn.functional.softplus(self.V)
        qF, qU, pU = self.gp(X=Xb, **kwargs)
        scale2 = torch.nn.functional.softplus(rs)
        qF2 = _dist(distributions.Normal, mF, scale2)
        pF2 = _dist(distributions.Normal, torch.zeros_like(mF), torch.ones_like(scale2))
        ll = poisson_expected_loglik([qF, qF2], [self.W, self.W2], V if idx is None else V[idx], y, E=E,
                                     with_lgamma=with_lgamma, eps=eps)
        return ll, qF, qU, pU, qF2, pF2
