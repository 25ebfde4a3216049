#!/usr/bin/env python3
"""Time the notebooks' minibatch training step (Slideseq_NSF_newest_version.ipynb:361-364, 571:
N_b=7000 spots per step, M=3000 inducing points, L=20 latents, fp32, frozen Z / sigma / lengthscale)
through the gpzoo modules: forward + loss.backward() + Adam step, with and without the factor cache."""
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo.gp import SVGP, WSVGP  # noqa: E402
from gpzoo.kernels import NSF_RBF  # noqa: E402
from gpzoo.likelihoods import GaussianLikelihood  # noqa: E402
from gpzoo.utilities import _elbo_terms  # noqa: E402
from gpzoo_amd.ops import deferred_info  # noqa: E402


def main():
    torch.manual_seed(0)
    N, Nb, M, L = 40000, 7000, 3000, 20
    dev = torch.device("cuda")
    X = (torch.rand(N, 2) * 200 - 100).to(dev)
    y = torch.randn(L, N).to(dev)
    for cls in (WSVGP, SVGP):
        gp = cls(NSF_RBF(sigma=1.0, lengthscale=8.0, L=L), dim=2, M=M, jitter=1e-1)
        gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu(), requires_grad=False)
        gp.mu = nn.Parameter(torch.zeros(L, M))
        gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
        for t in gp.kernel.parameters():
            t.requires_grad_(False)
        model = GaussianLikelihood(gp, noise=0.5).to(dev)
        opt = torch.optim.Adam([gp.mu, gp.Lu], lr=1e-2, fused=os.environ.get("GPZ_FUSED_ADAM", "0") == "1")
        for cache in (False, True):
            gp.cache_factor = cache
            times = []
            for it in range(6):
                idx = torch.randperm(N, device=dev)[:Nb]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                opt.zero_grad()
                with deferred_info():        # as gpzoo.utilities.train* run a step: Kzz's info word is read once, behind the backward's launches
                    loss = _elbo_terms(model, X[idx], y[:, idx], 1)
                    loss.backward()
                opt.step()
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            print(f"{cls.__name__:6s} cache={cache!s:5s} step = {1e3 * min(times[1:]):7.2f} ms  (loss {loss.item():.1f})")


def nsf_main():
    """NSF2 (Poisson) minibatch step: Slideseq_NSF_newest_version.ipynb shape -- D=17702 genes, N_b=7000,
    M=3000, L=20, E=3 -- through the fused GP pass and the fused Poisson expected log-likelihood."""
    from gpzoo.likelihoods import NSF2
    from gpzoo.utilities import whitened_KL_batched
    torch.manual_seed(0)
    N, Nb, M, L, D, E = 20000, 7000, 3000, 20, 17702, 3
    dev = torch.device("cuda")
    X = (torch.rand(N, 2) * 200 - 100).to(dev)
    y = torch.poisson(2.0 * torch.rand(D, N)).to(dev)
    gp = WSVGP(NSF_RBF(sigma=1.0, lengthscale=8.0, L=L), dim=2, M=M, jitter=1e-1)
    gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu(), requires_grad=False)
    gp.mu = nn.Parameter(torch.zeros(L, M))
    gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    for t in gp.kernel.parameters():
        t.requires_grad_(False)
    model = NSF2(gp, y.cpu()[:, :8], L=L)
    model.V = nn.Parameter(torch.ones(N))
    model = model.to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    for fused in (False, True):
        times = []
        for it in range(5):
            idx = torch.randperm(N, device=dev)[:Nb]
            yb = y[:, idx]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            opt.zero_grad()
            if fused:
                ll, qF, qU, pU = model.expected_loglik(X, yb, idx=idx, E=E)
            else:
                pY, qF, qU, pU = model.forward_batched(X, idx, E=E)
                ll = pY.log_prob(yb).mean(0).sum()
            loss = -(ll - whitened_KL_batched(qU.mean, qU.scale_tril).sum())
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        print(f"NSF2 Poisson step, fused log-lik={fused!s:5s}: {1e3 * min(times[1:]):7.2f} ms  (loss {loss.item():.4g})")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nsf":
        nsf_main()
    else:
        main()
