#!/usr/bin/env python3
"""BASELINE.json's own configurations evaluated by the REFERENCE itself (build container only).

    MPLBACKEND=Agg python tests/golden/make_baseline_golden.py

Imports luisdiaz1997/GPzoo from /root/reference (read-only, never copied, never shipped) and runs its WSVGP on the
synthetic inputs gpzoo_amd/synthetic.py draws for
  * configs[1] as stated: N = 50 000, M = 512, L = 8, NSF_RBF, fp32 (and in fp64, the "truth" the element-wise fp32
    comparisons use);
  * configs[2] (N = 200 000, M = 2048, L = 32, Matern-3/2) on its first 8192 spots -- the slice bench.py's cpu_baseline leg
    evaluates; the reference cannot hold the whole configuration (157 GB) and is used on minibatches of this size in its
    notebooks (utilities.py:605-609);
  * configs[4] (MGGP: 4 groups x 50 000 spots, M = 2048, L = 32, MGGP_NSF_RBF, fp64) on every 24th spot, 8192 of them.
Stored (data only, < 1 MB per file): the closed-form ELBO (mggp_test_exact.ipynb:157-159), its log-likelihood and KL
parts, and q(F)'s mean / scale at 4096 (configs[1]) / 1024 (configs[2]) seeded spot indices.  The tests regenerate the inputs from the same seeds.
"""
import os
import sys
import time

os.environ.setdefault("MPLBACKEND", "Agg")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # the repo (gpzoo_amd.synthetic: seeded inputs, no HIP)
sys.path.insert(0, "/root/reference")                            # ... but `gpzoo` is the reference

import numpy as np
import torch
import torch.nn as nn

import gpzoo.gp as rgp          # noqa: E402  (the reference)
import gpzoo.kernels as rk      # noqa: E402
import gpzoo.likelihoods as rl  # noqa: E402
from gpzoo.utilities import whitened_KL  # noqa: E402

assert rgp.__file__.startswith("/root/reference"), rgp.__file__
from gpzoo_amd.synthetic import make_config  # noqa: E402


def inv_softplus(v):
    return float(np.log(np.expm1(v)))


def run(cfg, dtype, n_spots=None, nidx=4096, stride=1, **kw):
    c = make_config(cfg, **kw)
    L, M = c["mu"].shape
    mggp = c["kind"] == "mggp_nsf_rbf"
    if mggp:
        G = c["n_groups"]
        k = rk.MGGP_NSF_RBF(n_groups=G, L=L)
        k.sigma = nn.Parameter(c["sigma"].reshape(L, 1, 1).double().clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].reshape(L, 1, 1).double().clone())
        k.group_diff_param = nn.Parameter(c["group_diff"].reshape(L, 1, 1).double().clone())
    elif c["kind"] == "nsf_rbf":
        k = rk.NSF_RBF(L=L)
        k.sigma = nn.Parameter(c["sigma"].reshape(L, 1, 1).double().clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].reshape(L, 1, 1).double().clone())
    else:
        k = rk.batched_Matern32()
        k.sigma = nn.Parameter(c["sigma"].double().clone())
        k.lengthscale = nn.Parameter(c["lengthscale"].double().clone())
    gp = rgp.MGGP_WSVGP(k, dim=2, M=M, n_groups=c["n_groups"], jitter=c["jitter"]) if mggp else rgp.WSVGP(k, dim=2, M=M, jitter=c["jitter"])
    if mggp:
        gp.groupsZ = nn.Parameter(c["gZ"].clone(), requires_grad=False)
    gp.Z = nn.Parameter(c["Z"].double().clone())
    gp.mu = nn.Parameter(c["mu"].double().clone())
    gp.Lu = nn.Parameter(c["Lu_raw"].double().clone())
    model = rl.ExactLikelihood(gp, noise=inv_softplus(c["noise_sd"]))
    model = model.double() if dtype == torch.float64 else model.float()
    if mggp and not isinstance(k.embedding, nn.Parameter):
        k.embedding = k.embedding.to(dtype)          # a plain tensor attribute: .double() does not move it
    n = c["X"].shape[0] if n_spots is None else n_spots
    sel = torch.arange(0, n * stride, stride)        # every stride-th spot (configs[4]'s groups come in blocks of 50 000)
    X, y = c["X"][sel].to(dtype), c["y"][:, sel].to(dtype)
    fkw = dict(groupsX=c["gX"][sel]) if mggp else {}
    t0 = time.time()
    with torch.no_grad():
        pY, qF, qU, pU = model(X=X, E=1, **fkw)
        s = torch.nn.functional.softplus(model.noise).double()
        kl = torch.stack([whitened_KL(qU.mean[l], qU.scale_tril[l]) for l in range(L)]).double()
        loglik = (pY.log_prob(y).double() - (qF.scale.double() ** 2) / (2 * s ** 2)).sum(-1)      # (L,)
        elbo = loglik.sum() - kl.sum()
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(777))[:nidx].sort().values
    print(f"cfg {cfg} {dtype} n={n}: elbo {float(elbo):.6f}  ({time.time() - t0:.1f} s)", flush=True)
    return dict(elbo=np.float64(float(elbo)), loglik=loglik.numpy(), kl=kl.numpy(), idx=idx.numpy(),
                mean=qF.mean[:, idx].numpy(), scale=qF.scale[:, idx].numpy(), n=np.int64(n),
                noise_sd=np.float64(float(s)))      # softplus of the (float32-initialised) noise parameter, as the model used it


def run_grads(cfg, dtype, **kw):
    """Gradients of -ELBO w.r.t. every parameter through the reference's own autograd graph (loss.backward(), utilities.py:485)."""
    c = make_config(cfg, **kw)
    L, M = c["mu"].shape
    k = rk.NSF_RBF(L=L)
    k.sigma = nn.Parameter(c["sigma"].reshape(L, 1, 1).double().clone())
    k.lengthscale = nn.Parameter(c["lengthscale"].reshape(L, 1, 1).double().clone())
    gp = rgp.WSVGP(k, dim=2, M=M, jitter=c["jitter"])
    gp.Z = nn.Parameter(c["Z"].double().clone())
    gp.mu = nn.Parameter(c["mu"].double().clone())
    gp.Lu = nn.Parameter(c["Lu_raw"].double().clone())
    model = rl.ExactLikelihood(gp, noise=inv_softplus(c["noise_sd"]))
    model = model.double() if dtype == torch.float64 else model.float()
    X, y = c["X"].to(dtype), c["y"].to(dtype)
    t0 = time.time()
    pY, qF, qU, pU = model(X=X, E=1)
    s = torch.nn.functional.softplus(model.noise)
    kl = torch.stack([whitened_KL(qU.mean[l], qU.scale_tril[l]) for l in range(L)]).sum()
    loss = -(pY.log_prob(y).sum() - (qF.scale ** 2).sum() / (2 * s ** 2) - kl)
    loss.backward()
    gLu = gp.Lu.grad.detach()
    print(f"cfg {cfg} grads {dtype}: loss {float(loss):.6f}  ({time.time() - t0:.1f} s)", flush=True)
    return dict(loss=np.float64(float(loss)), noise_sd=np.float64(float(s)), grad_mu=gp.mu.grad.numpy(),
                grad_Z=gp.Z.grad.numpy(), grad_sigma=k.sigma.grad.reshape(-1).numpy(),
                grad_lengthscale=k.lengthscale.grad.reshape(-1).numpy(), grad_noise=np.float64(float(model.noise.grad)),
                grad_Lu_rowsum=gLu.sum(-1).numpy(), grad_Lu_colsum=gLu.sum(-2).numpy(),
                grad_Lu_diag=torch.diagonal(gLu, dim1=-2, dim2=-1).numpy().copy(), grad_Lu_absmax=np.float64(float(gLu.abs().max())))


def save(name, f64, f32):
    out = {}
    for tag, r in (("f64", f64), ("f32", f32)):
        out.update({f"{tag}_{k}": v for k, v in r.items()})
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if os.environ.get("GPZ_BASELINE_ONLY", "") in ("", "2"):
        save("baseline_cfg2.npz", run(2, torch.float64), run(2, torch.float32))
    if os.environ.get("GPZ_BASELINE_ONLY", "") in ("", "2g"):    # configs[1]: the training step's gradients, all parameters
        save("baseline_cfg2_grads.npz", run_grads(2, torch.float64), run_grads(2, torch.float32))
    if os.environ.get("GPZ_BASELINE_ONLY", "") in ("", "5"):     # configs[4]: MGGP, fp64 (its stated precision) and fp32, every 24th spot
        save("baseline_cfg5_slice.npz", run(5, torch.float64, n_spots=8192, nidx=1024, stride=24),
             run(5, torch.float32, n_spots=8192, nidx=1024, stride=24))
    if os.environ.get("GPZ_BASELINE_ONLY", "") not in ("", "3"):
        sys.exit(0)
    save("baseline_cfg3_slice.npz", run(3, torch.float64, n_spots=8192, nidx=1024), run(3, torch.float32, n_spots=8192, nidx=1024))
