"""Numeric helpers importable under the reference's names (gpzoo/utilities.py).

On the fused HIP path none of these is called: jitter, the whitened KL and the
svgp_forward moments are produced inside ``gpz_svgp_forward``.  They exist so
notebook code that calls them directly on its own tensors keeps working; they
are element-wise / tiny operations expressed with torch on whatever device the
caller's tensors live on.
"""
from __future__ import annotations

import torch


def _torch_sqrt(x, eps=1e-12):
    """sqrt(x + eps): avoids the NaN gradient of sqrt at 0 (reference utilities.py:450-456)."""
    return (x + eps).sqrt()


def _embed_distance_matrix(distance_matrix):
    """Classical-MDS embedding of a (G,G) group-distance matrix (reference
    utilities.py:459-469).  G is a handful of tissue groups: stays in torch, runs once."""
    G = len(distance_matrix)
    centre = torch.eye(G) - torch.full((G, G), 1.0 / G)
    gram = -0.5 * (centre @ (distance_matrix ** 2) @ centre)
    evals, evecs = torch.linalg.eigh(gram)
    evals = evals.clamp(min=0)
    return evecs @ torch.diag(_torch_sqrt(evals, 1e-6))


def _squared_dist(X, Z):
    """Pairwise squared distances, clamped at zero (reference utilities.py:399-405)."""
    r2 = (X ** 2).sum(1, keepdim=True) - 2 * X.matmul(Z.t()) + (Z ** 2).sum(1, keepdim=True).t()
    return r2.clamp(min=0)


def add_jitter(K, jitter=1e-3):
    """IN-PLACE diagonal jitter on (M,M) or (L,M,M); returns the same tensor, and
    None for other ranks like the reference (utilities.py:407-418)."""
    if K.dim() in (2, 3):
        K.diagonal(dim1=-2, dim2=-1).add_(jitter)
        return K
    return None


def reshape_param(param):
    return param.view(-1, param.shape[-2], param.shape[-1])


def whitened_KL(mz, Lz):
    """KL(N(mz, Lz Lz^T) || N(0, I)) for ONE GP: mz (M,), Lz (M,M) -- the reference's
    2-D-only contract (utilities.py:27-36).  Use ``whitened_KL_batched`` for (L,M,M)."""
    M = len(mz)
    return 0.5 * (-2 * torch.log(torch.diagonal(Lz)).sum() + (Lz ** 2).sum() + (mz ** 2).sum() - M)


def whitened_KL_batched(mz, Lz):
    """Per-latent whitened KL for mz (..., M), Lz (..., M, M)."""
    M = mz.shape[-1]
    logdiag = torch.log(torch.diagonal(Lz, dim1=-2, dim2=-1)).sum(-1)
    return 0.5 * (-2 * logdiag + (Lz ** 2).sum((-2, -1)) + (mz ** 2).sum(-1) - M)


def svgp_forward(Kxx, Kzz, W, inducing_mean, inducing_cov):
    """mean = W mu (L,N,1); cov = Kxx + sum((W (S - Kzz)) * W, -1) (L,N) for caller-supplied
    matrices (reference utilities.py:382-397).  SVGP.forward does not call this: the fused HIP
    path evaluates the same moments from Linv without forming W or S."""
    mean = W @ inducing_mean.unsqueeze(-1)
    cov = Kxx + ((W @ (inducing_cov - Kzz)) * W).sum(-1)
    return mean, cov


def _elbo_terms(model, X, y, E, **kwargs):
    """-ELBO of one step in the reference's Monte-Carlo form (utilities.py:479-481):
    mean over E samples of log p(y | F), minus KL(qU || pU) (whitened KL when pU is None)."""
    from torch import distributions
    pY, _, qU, pU = model(X=X, E=E, **kwargs)
    loglik = pY.log_prob(y).mean(dim=0).sum() if pY.loc.dim() > y.dim() else pY.log_prob(y).sum()
    kl = whitened_KL_batched(qU.mean, qU.scale_tril).sum() if pU is None else distributions.kl_divergence(qU, pU).sum()
    return -(loglik - kl)


def train(model, optimizer, X, y, device=None, steps=200, E=20, **kwargs):
    """Full-batch optimisation loop with the reference's signature (utilities.py:471-493).  The
    forward and the mu / Lu gradients run on the fused HIP path; the optimiser step is torch's.
    Returns the list of losses (one host sync per step, as in the reference)."""
    losses = []
    for _ in range(steps):
        optimizer.zero_grad()
        loss = _elbo_terms(model, X, y, E, **kwargs)
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    return losses


def train_batched(model, optimizer, X, y, device=None, steps=200, E=20, batch_size=1000, **kwargs):
    """Mini-batched variant (reference utilities.py:600-632 shape): a fresh random subset of
    `batch_size` spots per step, log-likelihood rescaled by N / batch_size is NOT applied, like
    the reference."""
    losses = []
    N = X.shape[0]
    for _ in range(steps):
        idx = torch.multinomial(torch.ones(N, device=X.device), min(batch_size, N), replacement=False)
        optimizer.zero_grad()
        kw = dict(kwargs)
        if "groupsX" in kw:
            kw["groupsX"] = kw["groupsX"][idx]
        loss = _elbo_terms(model, X[idx], y[..., idx], E, **kw)
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
    return losses
