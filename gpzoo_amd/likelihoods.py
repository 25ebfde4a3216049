"""Gaussian likelihood wrappers with the gpzoo.likelihoods class API.

Thin consumers of the hot path's outputs (they stay torch):
``forward`` calls the GP once -- one fused HIP pass -- and wraps the result in
torch distributions exactly like reference likelihoods.py:7-36.  The Poisson
factor models of the reference are outside the path (SURVEY §8f "next" #2).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import distributions


class GaussianLikelihood(nn.Module):
    """pY = Normal(F, softplus(noise)) with F ~ qF.rsample((E,)); reference likelihoods.py:7-20."""

    def __init__(self, gp, noise=0.1):
        super().__init__()
        self.gp = gp
        self.noise = nn.Parameter(torch.tensor(noise))

    def forward(self, X, E=1, verbose=False, **kwargs):
        qF, qU, pU = self.gp(X, verbose=verbose, **kwargs)
        F = qF.rsample((E,))
        pY = distributions.Normal(F, torch.nn.functional.softplus(self.noise))
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
        pY = distributions.Normal(qF.mean, torch.nn.functional.softplus(self.noise))
        return pY, qF, qU, pU

    def elbo(self, X, y, **kwargs):
        sd = float(torch.nn.functional.softplus(self.noise.detach()))
        return self.gp.elbo(X, y, sd, groupsX=kwargs.get('groupsX'))[0]
