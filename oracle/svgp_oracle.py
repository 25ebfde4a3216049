"""CPU oracle for the SVGP / WSVGP hot path of luisdiaz1997/GPzoo.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (plain torch CPU tensor
ops, no nn.Module; the gradient tests differentiate through it with torch autograd) of the arithmetic the reference performs on its hot
path.  It is the *checker* for the HIP kernels: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``gpzoo_amd/`` imports it and the product path never
falls back to it.

Pinning: the reference has no tests / golden vectors of its own (SURVEY.md §4,
§8c), so this oracle is pinned against outputs of the reference itself, imported
in the build container from ``/root/reference`` by ``tests/golden/make_golden.py``
and committed as ``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` checks
every stored tensor).  Third-party arithmetic (cholesky, solve_triangular,
cdist) is torch 2.10.0's CPU implementation on both sides.

Every function cites the reference lines (``/root/reference/gpzoo/...``) it
restates.  Shapes: L latents, M inducing points, N data points, d input dims.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

# --------------------------------------------------------------------------
# covariance functions
# --------------------------------------------------------------------------

KINDS = ("rbf", "nsf_rbf", "batched_rbf", "matern32",
         "mggp_rbf", "mggp_nsf_rbf", "batched_mggp_rbf")


def sqdist_expansion(X: torch.Tensor, Z: torch.Tensor) -> torch.Tensor:
    """||x||^2 - 2 x.z + ||z||^2, clamped at 0 (utilities.py:399-405)."""
    xx = (X * X).sum(1, keepdim=True)
    zz = (Z * Z).sum(1, keepdim=True)
    r2 = xx - 2.0 * (X @ Z.t()) + zz.t()
    return r2.clamp(min=0)


def sqdist_direct(X: torch.Tensor, Z: torch.Tensor) -> torch.Tensor:
    """sum_k (x_k - z_k)^2 per pair: what the vmap kernels evaluate
    (kernels.py:14-16, 44-47)."""
    diff = X[:, None, :] - Z[None, :, :]
    return (diff * diff).sum(-1)


def embed_group_distances(D: torch.Tensor) -> torch.Tensor:
    """Classical-MDS embedding of a (G,G) distance matrix
    (utilities.py:450-469): B = -1/2 C D^2 C, eigh, negative eigenvalues
    clipped, Q diag(sqrt(lambda + 1e-6))."""
    G = D.shape[0]
    C = torch.eye(G) - torch.ones(G, G) / G
    B = -0.5 * (C @ (D ** 2) @ C)
    lam, Q = torch.linalg.eigh(B)
    lam = torch.where(lam < 0, torch.zeros_like(lam), lam)
    return Q @ torch.diag((lam + 1e-6).sqrt())


def _per_latent(v: torch.Tensor) -> torch.Tensor:
    """(L,), (L,1,1) or scalar parameter -> (L,1,1) or 0-d."""
    if v.dim() == 0:
        return v
    return v.reshape(-1, 1, 1)


def kernel_matrix(kind: str, A: torch.Tensor, B: torch.Tensor,
                  sigma: torch.Tensor, lengthscale: torch.Tensor, *,
                  gA: Optional[torch.Tensor] = None,
                  gB: Optional[torch.Tensor] = None,
                  embedding: Optional[torch.Tensor] = None,
                  group_diff: Optional[torch.Tensor] = None,
                  input_dim: int = 2) -> torch.Tensor:
    """K(A, B): (a,b) for scalar parameters, (L,a,b) for per-latent ones.

    rbf / nsf_rbf     kernels.py:114-130, 139-155  (torch.cdist, then **2)
    batched_rbf       kernels.py:42-47, 57-58       (direct difference)
    matern32          kernels.py:14-20, 29-30       (direct difference, sqrt)
    mggp_rbf          kernels.py:171-191            (a un-squared)
    mggp_nsf_rbf      kernels.py:206-228            (a squared, input_dim)
    batched_mggp_rbf  kernels.py:75-89, 98-104      (|a|, p = true input dim)
    """
    s = _per_latent(sigma)
    ell = _per_latent(lengthscale)
    if kind in ("rbf", "nsf_rbf"):
        d2 = torch.cdist(A, B) ** 2
        return s ** 2 * torch.exp(-0.5 * d2 / ell ** 2)
    if kind == "batched_rbf":
        d2 = sqdist_direct(A, B)
        return s ** 2 * torch.exp(-0.5 * d2 / ell ** 2)
    if kind == "matern32":
        r = torch.sqrt(sqdist_direct(A, B))
        v = math.sqrt(3.0) * r / ell
        return s ** 2 * (1.0 + v) * torch.exp(-v)
    if kind in ("mggp_rbf", "mggp_nsf_rbf", "batched_mggp_rbf"):
        a = _per_latent(group_diff)
        eA, eB = embedding[gA], embedding[gB]
        if kind == "batched_mggp_rbf":
            g2 = sqdist_direct(eA, eB)
            d2 = sqdist_direct(A, B)
            den = a.abs() * g2 + 1.0
            p = A.shape[-1]
        else:
            g2 = sqdist_expansion(eA, eB)
            d2 = sqdist_expansion(A, B)
            den = (a ** 2 if kind == "mggp_nsf_rbf" else a) * g2 + 1.0
            p = input_dim
        return s ** 2 * torch.exp(-0.5 * (d2 / ell ** 2) / den) / den ** (0.5 * p)
    raise ValueError(kind)


def kernel_diag(sigma: torch.Tensor, n: int) -> torch.Tensor:
    """k(x,x) = sigma^2 for every kernel on the path: (n,) for scalar sigma,
    (L,n) per latent (kernels.py:26, 115-116, 143-144, 172-173, 207-208; the
    vmap kernels' intended contract is the commented kernels.py:54)."""
    s2 = sigma.reshape(-1) ** 2
    if sigma.dim() == 0:
        return s2.expand(n).clone()
    return s2[:, None].expand(-1, n).clone()


# --------------------------------------------------------------------------
# GP forward passes
# --------------------------------------------------------------------------

def add_jitter_(K: torch.Tensor, jitter: float) -> torch.Tensor:
    """In-place diagonal jitter on (M,M) or (L,M,M) (utilities.py:407-418)."""
    K.diagonal(dim1=-2, dim2=-1).add_(jitter)
    return K


def lower_cholesky_param(Lu_raw: torch.Tensor) -> torch.Tensor:
    """transform_to(constraints.lower_cholesky): strict lower triangle kept,
    diagonal exponentiated (gp.py:220, 278; torch LowerCholeskyTransform)."""
    d = torch.diagonal(Lu_raw, dim1=-2, dim2=-1).exp()
    return Lu_raw.tril(-1) + torch.diag_embed(d)


def wsvgp_moments(Kxx, Kzx, Kzz_jit, mu, Lu_raw):
    """Whitened q(F) moments (gp.py:270-296).  Kzz_jit already jittered.
    Returns mean, scale, Lu, chol."""
    chol = torch.linalg.cholesky(Kzz_jit)
    Wt = torch.linalg.solve_triangular(chol, Kzx, upper=False)
    W = Wt.transpose(-2, -1)
    Lu = lower_cholesky_param(Lu_raw)
    var = (Kxx - (W ** 2).sum(-1)).clamp(min=0.0) + ((W @ Lu) ** 2).sum(-1)
    mean = (W @ mu.unsqueeze(-1)).squeeze(-1)
    return mean, var ** 0.5, Lu, chol


def svgp_moments(Kxx, Kzx, Kzz_jit, mu, Lu_raw, clamp_min=1e-6):
    """Un-whitened q(F) moments (gp.py:213-228 with utilities.py:382-397;
    MGGP_SVGP uses clamp_min=5e-2, gp.py:378)."""
    chol = torch.linalg.cholesky(Kzz_jit)
    W = torch.cholesky_solve(Kzx, chol).transpose(-2, -1)
    Lu = lower_cholesky_param(Lu_raw)
    S = Lu @ Lu.transpose(-2, -1)
    mean = (W @ mu.unsqueeze(-1)).squeeze(-1)
    var = Kxx + ((W @ (S - Kzz_jit)) * W).sum(-1)
    return mean, var.clamp(min=clamp_min) ** 0.5, Lu, chol


# --------------------------------------------------------------------------
# KL terms and the closed-form Gaussian ELBO
# --------------------------------------------------------------------------

def whitened_kl(mu: torch.Tensor, Lu: torch.Tensor) -> torch.Tensor:
    """KL(N(mu, Lu Lu^T) || N(0, I)) per latent (utilities.py:27-36, applied
    per GP as the notebooks do; batched here over leading dims)."""
    M = mu.shape[-1]
    logdiag = torch.diagonal(Lu, dim1=-2, dim2=-1).log().sum(-1)
    return 0.5 * (-2.0 * logdiag + (Lu ** 2).sum((-2, -1)) + (mu ** 2).sum(-1) - M)


def mvn_kl(mu: torch.Tensor, Lu: torch.Tensor, chol: torch.Tensor) -> torch.Tensor:
    """KL(N(mu, Lu Lu^T) || N(0, chol chol^T)) per latent -- what
    distributions.kl_divergence(qU, pU) evaluates at utilities.py:481
    (torch/distributions/kl.py MVN-MVN closed form)."""
    M = mu.shape[-1]
    half_logdet_p = torch.diagonal(chol, dim1=-2, dim2=-1).log().sum(-1)
    half_logdet_q = torch.diagonal(Lu, dim1=-2, dim2=-1).log().sum(-1)
    A = torch.linalg.solve_triangular(chol, Lu, upper=False)
    b = torch.linalg.solve_triangular(chol, mu.unsqueeze(-1), upper=False)
    return half_logdet_p - half_logdet_q + 0.5 * ((A ** 2).sum((-2, -1)) + (b ** 2).sum((-2, -1)) - M)


def gaussian_elbo(y, mean, scale, noise_sd: float, kl) -> torch.Tensor:
    """Closed-form Gaussian ELBO of mggp_test_exact.ipynb:157-159:
    sum log N(y; mean, s^2) - sum scale^2 / (2 s^2) - sum KL, in fp64."""
    y, mean, scale = y.double(), mean.double(), scale.double()
    s2 = float(noise_sd) ** 2
    loglik = (-0.5 * math.log(2.0 * math.pi * s2) - (y - mean) ** 2 / (2.0 * s2)).sum()
    return loglik - (scale ** 2).sum() / (2.0 * s2) - kl.double().sum()


# --------------------------------------------------------------------------
# one complete evaluation (what bench.py times as the CPU baseline)
# --------------------------------------------------------------------------

def elbo_eval(kind: str, whitened: bool, X, y, Z, sigma, lengthscale, mu, Lu_raw,
              jitter: float, noise_sd: float, *, gX=None, gZ=None, embedding=None,
              group_diff=None, input_dim=2, clamp_min=1e-6):
    """kernel build -> Cholesky -> solves -> moments -> KL -> scalar ELBO,
    in the reference's op order.  Returns
    (elbo fp64 scalar, mean, scale)."""
    kw = dict(embedding=embedding, group_diff=group_diff, input_dim=input_dim)
    Kxx = kernel_diag(sigma, X.shape[0])
    Kzx = kernel_matrix(kind, Z, X, sigma, lengthscale, gA=gZ, gB=gX, **kw)
    Kzz = kernel_matrix(kind, Z, Z, sigma, lengthscale, gA=gZ, gB=gZ, **kw).contiguous()
    add_jitter_(Kzz, jitter)
    if whitened:
        mean, scale, Lu, chol = wsvgp_moments(Kxx, Kzx, Kzz, mu, Lu_raw)
        kl = whitened_kl(mu, Lu)
    else:
        mean, scale, Lu, chol = svgp_moments(Kxx, Kzx, Kzz, mu, Lu_raw, clamp_min)
        kl = mvn_kl(mu, Lu, chol)
    return gaussian_elbo(y, mean, scale, noise_sd, kl), mean, scale


# --------------------------------------------------------------------------
# VNNGP: K nearest inducing points per datum (gp.py:7-122)
# --------------------------------------------------------------------------

def vnngp_moments(X, Z, sigma, lengthscale, mu, Lu_raw, jitter: float, K: int, idx=None):
    """q(F) of the nearest-neighbour variational GP (gp.py:21-122): for every x the K nearest
    inducing points (argsort of torch.cdist, gp.py:31,64), the K x K blocks of L L^T = Kzz + jitter I
    (jittered once more in place, gp.py:68-77), W = k_xz[idx] inv(block), S block from Lu[idx],
    then svgp_forward (utilities.py:382-397) and clamp(cov, 5e-2) (gp.py:117).
    ``idx``: a neighbour table to use instead of the argsort (every later line of gp.py:66-122 is a gather on it).
    Returns mean, scale (L,N) or (N,), the neighbour indices (N,K), Lu, chol."""
    batched = sigma.dim() > 0
    s = sigma.reshape(-1, 1, 1)
    ell = lengthscale.reshape(-1, 1, 1)
    dist = torch.cdist(X, Z)
    Kxz = s ** 2 * torch.exp(-0.5 * dist ** 2 / ell ** 2)                      # (L,N,M)
    Kzz = s ** 2 * torch.exp(-0.5 * torch.cdist(Z, Z) ** 2 / ell ** 2)
    Lq = lower_cholesky_param(Lu_raw).reshape(-1, Z.shape[0], Z.shape[0])
    chol = torch.linalg.cholesky(add_jitter_(Kzz.contiguous(), jitter))
    if idx is None:
        idx = torch.argsort(dist, dim=1)[:, :K]                                 # (N,K)
    lL = chol[:, idx]                                                           # (L,N,K,M)
    lK = lL @ lL.transpose(-2, -1)
    lK = add_jitter_(lK.reshape(-1, K, K).contiguous(), jitter).reshape(lK.shape)
    W = (torch.gather(Kxz, 2, idx.expand(Kxz.shape[0], -1, -1))[:, :, None, :] @ torch.inverse(lK))  # (L,N,1,K)
    lmu = mu.reshape(-1, Z.shape[0])[:, idx]                                    # (L,N,K)
    lLu = Lq[:, idx]
    lS = lLu @ lLu.transpose(-2, -1)
    mean = (W @ lmu[..., None]).squeeze(-1).squeeze(-1)
    cov = s.reshape(-1, 1) ** 2 + ((W @ (lS - lK)) * W).sum(-1).squeeze(-1)
    scale = cov.clamp(min=5e-2) ** 0.5
    if not batched:
        mean, scale = mean[0], scale[0]
    return mean, scale, idx, lower_cholesky_param(Lu_raw), chol if batched else chol[0]


# --------------------------------------------------------------------------
# Poisson factor models: the step right after the path (likelihoods.py:39-53, 74-97, 100-222)
# --------------------------------------------------------------------------

def poisson_expected_loglik(mean, scale, eps, W_pos, V_pos, y, with_lgamma: bool = True) -> torch.Tensor:
    """Monte-Carlo objective of NSF2 / Hybrid_NSF2 as the reference's loops evaluate it: F = qF.rsample((E,)) =
    mean + scale * eps (E,Lt,N); rate = V * (softplus(W) @ exp(F)) (likelihoods.py:49-53, 81-86; the hybrids add their
    two factor sets' rates, :111-121 -- here both sets are rows of one (Lt,N) / columns of one (D,Lt) operand);
    pY.log_prob(y).mean(axis=0).sum() (utilities.py:479, 537, 612) or, without the log y! term,
    (y log rate - rate).mean(axis=0).sum() (utilities.py:508-510).  W_pos, V_pos: after softplus."""
    F = mean[None] + scale[None] * eps
    rate = V_pos * torch.matmul(W_pos, torch.exp(F))
    ll = y * torch.log(rate) - rate
    if with_lgamma:
        ll = ll - torch.lgamma(y + 1)
    return ll.mean(dim=0).sum()


def hybrid_exact_loglik(mean1, scale1, mean2, scale2, W1_pos, W2_pos, V_pos, y, with_lgamma: bool = True):
    """Hybrid_NSF_Exact (likelihoods.py:167-222): rate = V * (W1 exp(m1 + s1^2/2) + W2 exp(m2 + s2^2/2)), a (D,N)
    matrix -- nothing is sampled, so the loops' `.mean(axis=0)` (utilities.py:510, 537) averages over GENES.
    Returns (objective, rate)."""
    rate = V_pos * (W1_pos @ torch.exp(mean1 + 0.5 * scale1 ** 2) + W2_pos @ torch.exp(mean2 + 0.5 * scale2 ** 2))
    ll = y * torch.log(rate) - rate
    if with_lgamma:
        ll = ll - torch.lgamma(y + 1)
    return ll.mean(dim=0).sum(), rate
