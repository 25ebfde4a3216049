#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (build container only).

    MPLBACKEND=Agg python tests/golden/make_golden.py

Imports luisdiaz1997/GPzoo from /root/reference (read-only, never copied, never
shipped), evaluates every working (GP class x kernel class) cell of SURVEY.md
§8a's compatibility matrix on small seeded inputs in fp64 and fp32, and stores
inputs + outputs as .npz next to this script.  The fixtures are data only.

Stored per case: inputs (X, y, Z, groups, sigma, lengthscale, a, embedding, mu,
Lu_raw, jitter, noise_sd) and reference outputs (Kxx, Kzx, Kzz_jit, chol, mean,
scale, Lu, kl, elbo, and the gradients of -ELBO w.r.t. mu and Lu from the reference's
own autograd graph).  The reference has no tests of its own, so these pin the
oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_hip_golden.py)
to torch 2.10.0's CPU arithmetic run through the reference's code.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn
from torch import distributions

import gpzoo.gp as rgp          # noqa: E402  (the reference)
import gpzoo.kernels as rk      # noqa: E402
import gpzoo.likelihoods as rl  # noqa: E402
from gpzoo.utilities import whitened_KL  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def inv_softplus(v):
    return float(np.log(np.expm1(v)))


sys.path.insert(0, HERE)
from inputs import gen, make_inputs  # noqa: E402  (shared with the tests, which regenerate the larger cases' inputs)


def run_case(name, gp_cls, kern, inp, dtype, jitter, noise_sd, whitened, mggp):
    """Evaluate the reference model and collect everything the parity tests use."""
    M, d = inp["Z"].shape
    kw = dict(kernel=kern, dim=d, M=M, jitter=jitter)
    if mggp:
        kw["n_groups"] = int(kern.embedding.shape[0])
    gp = gp_cls(**kw)
    gp.Z = nn.Parameter(inp["Z"].clone())
    gp.mu = nn.Parameter(inp["mu"].clone())
    gp.Lu = nn.Parameter(inp["Lu_raw"].clone())
    if mggp:
        gp.groupsZ = nn.Parameter(inp["gZ"].clone(), requires_grad=False)
    model = rl.ExactLikelihood(gp, noise=inv_softplus(noise_sd))
    model = model.double() if dtype == torch.float64 else model.float()
    if mggp and not isinstance(kern.embedding, nn.Parameter):
        kern.embedding = kern.embedding.to(dtype)  # plain tensor attr: .double() does not move it
    X = inp["X"].to(dtype)
    y = inp["y"].to(dtype)
    fkw = dict(groupsX=inp["gX"]) if mggp else {}
    with torch.no_grad():
        pY, qF, qU, pU = model(X=X, E=1, **fkw)
        # intermediate matrices through the same kernel object
        if mggp:
            Kzx = kern(gp.Z, X, gp.groupsZ, inp["gX"])
            Kzz = kern(gp.Z, gp.Z, gp.groupsZ, gp.groupsZ).contiguous()
            Kxx = kern(X, X, inp["gX"], inp["gX"], diag=True)
        else:
            Kzx = kern(gp.Z, X)
            Kzz = kern(gp.Z, gp.Z).contiguous()
            Kxx = kern(X, X, diag=True)
        Kzz_jit = Kzz.clone()
        Kzz_jit.diagonal(dim1=-2, dim2=-1).add_(jitter)
        chol = torch.linalg.cholesky(Kzz_jit)
        s = torch.nn.functional.softplus(model.noise)
        if whitened:
            if qU.mean.dim() == 1:
                kl = whitened_KL(qU.mean, qU.scale_tril)
            else:  # whitened_KL is 2-D only: apply per latent as the notebooks do
                kl = torch.stack([whitened_KL(qU.mean[l], qU.scale_tril[l]) for l in range(qU.mean.shape[0])])
        else:
            kl = distributions.kl_divergence(qU, pU)
        elbo = pY.log_prob(y).double().sum() - (qF.scale.double() ** 2).sum() / (2 * s.double() ** 2) - kl.double().sum()
    # gradients of -ELBO w.r.t. mu and Lu through the reference's own autograd graph (loss.backward(),
    # utilities.py:485): goldens for the backward pass
    model.zero_grad()
    pY_g, qF_g, qU_g, pU_g = model(X=X, E=1, **fkw)
    s_g = torch.nn.functional.softplus(model.noise)
    if whitened:
        kl_g = (whitened_KL(qU_g.mean, qU_g.scale_tril) if qU_g.mean.dim() == 1 else
                torch.stack([whitened_KL(qU_g.mean[l], qU_g.scale_tril[l]) for l in range(qU_g.mean.shape[0])]).sum())
    else:
        kl_g = distributions.kl_divergence(qU_g, pU_g).sum()
    loss = -(pY_g.log_prob(y).sum() - (qF_g.scale ** 2).sum() / (2 * s_g ** 2) - kl_g)
    loss.backward()
    grad_mu, grad_Lu = gp.mu.grad.detach().clone(), gp.Lu.grad.detach().clone()
    out = {k: v.to(dtype).numpy() if v.is_floating_point() else v.numpy() for k, v in inp.items()}
    out.update(grad_mu=grad_mu.numpy(), grad_Lu=grad_Lu.numpy(), grad_Z=gp.Z.grad.detach().numpy(),
               grad_sigma=kern.sigma.grad.detach().numpy(), grad_lengthscale=kern.lengthscale.grad.detach().numpy())
    if mggp:
        out["grad_group_diff"] = kern.group_diff_param.grad.detach().numpy()
    out.update(
        sigma=kern.sigma.detach().numpy(), lengthscale=kern.lengthscale.detach().numpy(),
        jitter=np.float64(jitter), noise_sd=np.float64(float(s)),
        Kxx=Kxx.numpy(), Kzx=Kzx.numpy(), Kzz_jit=Kzz_jit.numpy(), chol=chol.numpy(),
        mean=qF.mean.numpy(), scale=qF.scale.numpy(), Lu=qU.scale_tril.numpy(),
        kl=kl.numpy(), elbo=np.float64(float(elbo)),
    )
    if mggp:
        out["embedding"] = kern.embedding.detach().numpy()
        out["group_diff"] = kern.group_diff_param.detach().numpy()
        out["input_dim"] = np.int64(kern.input_dim)
    return out


def per_latent(vals, shape3=False):
    t = torch.tensor(vals, dtype=torch.float64)
    return t.reshape(-1, 1, 1) if shape3 else t


def build_kernel(kind, L):
    sig = [1.0, 0.8, 1.3][:max(L, 1)]
    ell = [2.5, 4.0, 6.0][:max(L, 1)]
    a = [0.7, 0.4, 1.1][:max(L, 1)]
    if kind == "rbf":
        return rk.RBF(sigma=1.2, lengthscale=3.0)
    if kind == "nsf_rbf":
        k = rk.NSF_RBF(L=L)
        k.sigma = nn.Parameter(per_latent(sig, True)); k.lengthscale = nn.Parameter(per_latent(ell, True))
        return k
    if kind == "matern32":
        k = rk.batched_Matern32()
        k.sigma = nn.Parameter(per_latent(sig)); k.lengthscale = nn.Parameter(per_latent(ell))
        return k
    if kind == "mggp_rbf":
        return rk.MGGP_RBF(sigma=1.1, lengthscale=3.5, group_diff_param=0.6, n_groups=3)
    if kind == "mggp_nsf_rbf":
        k = rk.MGGP_NSF_RBF(n_groups=3, L=L)
        k.sigma = nn.Parameter(per_latent(sig, True)); k.lengthscale = nn.Parameter(per_latent(ell, True))
        k.group_diff_param = nn.Parameter(per_latent(a, True))
        return k
    raise ValueError(kind)


CASES = [
    # name, gp class, kernel kind, L (0 = single GP), whitened, mggp, jitter
    ("wsvgp_rbf", "WSVGP", "rbf", 0, True, False, 1e-2),
    ("wsvgp_nsf_rbf", "WSVGP", "nsf_rbf", 3, True, False, 1e-2),
    ("wsvgp_matern32", "WSVGP", "matern32", 3, True, False, 1e-2),
    ("svgp_rbf", "SVGP", "rbf", 0, False, False, 1e-2),
    ("svgp_nsf_rbf", "SVGP", "nsf_rbf", 3, False, False, 1e-2),
    ("svgp_matern32", "SVGP", "matern32", 3, False, False, 1e-2),
    ("mggp_wsvgp_mggp_rbf", "MGGP_WSVGP", "mggp_rbf", 0, True, True, 1e-2),
    ("mggp_wsvgp_mggp_nsf_rbf", "MGGP_WSVGP", "mggp_nsf_rbf", 3, True, True, 1e-2),
    ("mggp_svgp_mggp_rbf", "MGGP_SVGP", "mggp_rbf", 0, False, True, 1e-2),
    ("mggp_svgp_mggp_nsf_rbf", "MGGP_SVGP", "mggp_nsf_rbf", 3, False, True, 1e-2),
]


def kernel_only_cases():
    """Kernel matrices of the vmap kernels whose `diag` branch is broken at HEAD
    (SURVEY §8a a3/a7): the full-matrix branch works and is pinned here."""
    out = {}
    inp = make_inputs(77, 96, 24, 2, 3, n_groups=3)
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        X, Z = inp["X"].to(dtype), inp["Z"].to(dtype)
        k = rk.batched_RBF()
        k.sigma = nn.Parameter(torch.tensor([1.0, 0.8, 1.3], dtype=dtype))
        k.lengthscale = nn.Parameter(torch.tensor([2.5, 4.0, 6.0], dtype=dtype))
        ks = rk.batched_RBF(sigma=1.2, lengthscale=3.0).to(dtype)
        km = rk.batched_MGGP_RBF(sigma=1.1, lengthscale=3.5, group_diff_param=-0.6, n_groups=3).to(dtype)
        kms = rk.batched_Matern32(sigma=0.9, lengthscale=2.0).to(dtype)
        with torch.no_grad():
            out[f"{tag}_batched_rbf_vec"] = k(Z, X).numpy()
            out[f"{tag}_batched_rbf_scalar"] = ks(Z, X).numpy()
            out[f"{tag}_batched_mggp_rbf_scalar"] = km(Z, X, inp["gZ"], inp["gX"]).numpy()
            out[f"{tag}_matern32_scalar"] = kms(Z, X).numpy()
            out[f"{tag}_matern32_zz"] = kms(Z, Z).numpy()
        out[f"{tag}_X"], out[f"{tag}_Z"] = X.numpy(), Z.numpy()
        out[f"{tag}_embedding"] = km.embedding.detach().numpy()
    out["gX"], out["gZ"] = inp["gX"].numpy(), inp["gZ"].numpy()
    return out


def cfg1_case():
    """BASELINE.json configs[0]: 1-D regression, N=1000, M=64, single latent,
    RBF, fp64, un-whitened SVGP + ExactLikelihood (inputs per SURVEY §8d)."""
    from gpzoo_amd.synthetic import make_config  # the build's own generator
    c = make_config(1)
    kern = rk.RBF(sigma=1.0, lengthscale=1.0)
    inp = dict(X=c["X"], Z=c["Z"], mu=c["mu"], Lu_raw=c["Lu_raw"], y=c["y"])
    return run_case("cfg1", rgp.SVGP, kern, inp, torch.float64, c["jitter"], c["noise_sd"], False, False)


def main():
    torch.manual_seed(0)
    for i, (name, gpc, kind, L, whitened, mggp, jitter) in enumerate(CASES):
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            inp = make_inputs(100 + 10 * i, N=160, M=36, d=2, L=L, n_groups=3 if mggp else 0)
            kern = build_kernel(kind, L)
            out = run_case(name, getattr(rgp, gpc), kern, inp, dtype, jitter, 0.5, whitened, mggp)
            out["kind"] = np.array(kind); out["whitened"] = np.array(whitened)
            np.savez_compressed(os.path.join(HERE, f"{name}_{tag}.npz"), **out)
            print(f"{name}_{tag}: elbo={float(out['elbo']):.10f}")
    np.savez_compressed(os.path.join(HERE, "kernels_only.npz"), **kernel_only_cases())
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    out = cfg1_case()
    out["kind"] = np.array("rbf"); out["whitened"] = np.array(False)
    np.savez_compressed(os.path.join(HERE, "cfg1_f64.npz"), **out)
    print(f"cfg1_f64: elbo={float(out['elbo']):.10f}")


if __name__ == "__main__" and not os.environ.get("GPZ_GOLDEN_ONLY"):
    main()


def poisson_cases():
    """NSF2 / Hybrid_NSF2 (reference likelihoods.py) on a tiny count matrix: the Monte-Carlo expected
    log-likelihood with KNOWN rsample noise (the reference's _standard_normal is patched to hand out
    stored eps) and the gradients of -ELBO w.r.t. W, V, mu, Lu (and the non-spatial prior)."""
    import torch.distributions.normal as tdn
    out = {}
    N, M, L, T, D, E = 160, 36, 3, 2, 25, 3
    inp = make_inputs(900, N=N, M=M, d=2, L=L)
    g = torch.Generator().manual_seed(901)
    y = torch.poisson(3.0 * torch.rand(D, N, generator=g), generator=g).float()
    eps1 = torch.randn(E, L, N, generator=g)
    eps2 = torch.randn(E, T, N, generator=g)
    orig = tdn._standard_normal
    for name in ("nsf2", "hybrid_nsf2"):
        kern = build_kernel("nsf_rbf", L).float()
        gp = rgp.WSVGP(kern, dim=2, M=M, jitter=1e-2)
        gp.Z = nn.Parameter(inp["Z"].float(), requires_grad=False)
        gp.mu = nn.Parameter(0.2 * inp["mu"].float())
        gp.Lu = nn.Parameter(inp["Lu_raw"].float())
        for t in kern.parameters():
            t.requires_grad_(False)
        torch.manual_seed(7)
        if name == "nsf2":
            model = rl.NSF2(gp, y, L=L)
            queue = [eps1]
        else:
            prior = rgp.GaussianPrior(y, L=T)
            model = rl.Hybrid_NSF2(gp, prior, y, L=L, T=T)
            queue = [eps1, eps2]
        it = iter(queue)
        tdn._standard_normal = lambda shape, dtype, device: next(it).to(dtype)
        try:
            res = model(X=inp["X"].float(), E=E)
        finally:
            tdn._standard_normal = orig
        pY, qF, qU = res[0], res[1], res[2]
        loglik = pY.log_prob(y).mean(0).sum()
        kl = torch.stack([whitened_KL(qU.mean[l], qU.scale_tril[l]) for l in range(L)]).sum()
        loss = -(loglik - kl)
        if name == "hybrid_nsf2":
            loss = loss + distributions.kl_divergence(res[4], res[5]).sum()
        loss.backward()
        Wp = model.W if name == "nsf2" else model.sf.W
        rec = dict(X=inp["X"].float().numpy(), Z=inp["Z"].float().numpy(), mu=gp.mu.detach().numpy(),
                   Lu_raw=gp.Lu.detach().numpy(), y=y.numpy(), eps1=eps1.numpy(), eps2=eps2.numpy(),
                   sigma=kern.sigma.detach().numpy(), lengthscale=kern.lengthscale.detach().numpy(),
                   W=Wp.detach().numpy(), V=model.V.detach().numpy(), loglik=np.float64(float(loglik)),
                   loss=np.float64(float(loss)), grad_W=Wp.grad.numpy(), grad_V=model.V.grad.numpy(),
                   grad_mu=gp.mu.grad.numpy(), grad_Lu=gp.Lu.grad.numpy(), rate_sum=np.float64(float(pY.rate.double().sum())))
        if name == "hybrid_nsf2":
            rec.update(W2=model.cf.W.detach().numpy(), mean2=prior.mean.detach().numpy(), scale2=prior.scale.detach().numpy(),
                       grad_W2=model.cf.W.grad.numpy(), grad_mean2=prior.mean.grad.numpy(), grad_scale2=prior.scale.grad.numpy())
        np.savez_compressed(os.path.join(HERE, f"poisson_{name}_f32.npz"), **rec)
        print(f"poisson_{name}_f32: loglik={float(loglik):.6f}")


def vnngp_cases():
    """VNNGP (reference gp.py:7-122; its unconditional prints are swallowed) with NSF_RBF.  With a
    scalar-parameter RBF the reference raises (indexes.repeat(Kxx_shape[0], 1) uses N for L, gp.py:83),
    so only the per-latent kernel has reference vectors."""
    import contextlib
    import io
    for kind, L in (("nsf_rbf", 3), ("nsf_rbf", 2)):
        for dtype, tag in ((torch.float64, f"L{L}_f64"), (torch.float32, f"L{L}_f32")):
            inp = make_inputs(700 + L, N=160, M=36, d=2, L=L)
            kern = build_kernel(kind, L)
            gp = rgp.VNNGP(kern, dim=2, M=36, K=5, jitter=1e-2)
            gp.Z = nn.Parameter(inp["Z"].clone())
            gp.mu = nn.Parameter(inp["mu"].clone())
            gp.Lu = nn.Parameter(inp["Lu_raw"].clone())
            gp = gp.double() if dtype == torch.float64 else gp.float()
            X = inp["X"].to(dtype)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                qF, qU, pU = gp(X)
                _, dist = kern(X, gp.Z, return_distance=True)
            idx = torch.argsort(dist, dim=1)[:, :5]
            # gradients of the negative Gaussian ELBO through the reference's own autograd graph
            y, noise_sd = inp["y"].to(dtype), 0.5
            with contextlib.redirect_stdout(io.StringIO()):
                qF_g, qU_g, pU_g = gp(X)
            loss = -(distributions.Normal(qF_g.mean, noise_sd).log_prob(y).sum()
                     - (qF_g.scale ** 2).sum() / (2 * noise_sd ** 2) - distributions.kl_divergence(qU_g, pU_g).sum())
            loss.backward()
            grads = dict(y=y.numpy(), noise_sd=np.float64(noise_sd), loss=np.float64(float(loss)),
                         grad_mu=gp.mu.grad.numpy(), grad_Lu=gp.Lu.grad.numpy(), grad_Z=gp.Z.grad.numpy(),
                         grad_sigma=kern.sigma.grad.numpy(), grad_lengthscale=kern.lengthscale.grad.numpy(),
                         n_clamped=np.int64(int((qF.scale ** 2 <= 5e-2 * (1 + 1e-9)).sum())))
            rec = dict(**grads, X=X.numpy(), Z=gp.Z.detach().numpy(), mu=gp.mu.detach().numpy(), Lu_raw=gp.Lu.detach().numpy(),
                       sigma=kern.sigma.detach().numpy(), lengthscale=kern.lengthscale.detach().numpy(),
                       mean=qF.mean.numpy(), scale=qF.scale.numpy(), idx=idx.numpy(), Lu=qU.scale_tril.numpy(),
                       chol=pU.scale_tril.numpy(), jitter=np.float64(1e-2), K=np.int64(5))
            np.savez_compressed(os.path.join(HERE, f"vnngp_{kind}_{tag}.npz"), **rec)
            print(f"vnngp_{kind}_{tag}: mean[0..2]={qF.mean.reshape(-1)[:3].tolist()} clamped={int(grads['n_clamped'])}"
                  f" loss={float(loss):.6f}")


def state_dict_cases():
    """state_dict keys + shapes of every reference model class (checkpoint-name compatibility,
    SURVEY.md §8f #4): data only -- names, shapes, dtypes."""
    import json
    y = torch.ones(7, 11)
    def gp_of(cls, kern, **kw):
        return cls(kern, dim=2, M=6, **kw)
    builders = {
        "RBF": lambda: rk.RBF(), "NSF_RBF": lambda: rk.NSF_RBF(L=3), "batched_RBF": lambda: rk.batched_RBF(),
        "batched_Matern32": lambda: rk.batched_Matern32(), "MGGP_RBF": lambda: rk.MGGP_RBF(n_groups=3),
        "MGGP_NSF_RBF": lambda: rk.MGGP_NSF_RBF(L=3, n_groups=3), "batched_MGGP_RBF": lambda: rk.batched_MGGP_RBF(n_groups=3),
        "VNNGP": lambda: gp_of(rgp.VNNGP, rk.NSF_RBF(L=3), K=2), "SVGP": lambda: gp_of(rgp.SVGP, rk.RBF()),
        "WSVGP": lambda: gp_of(rgp.WSVGP, rk.NSF_RBF(L=3)),
        "MGGP_SVGP": lambda: gp_of(rgp.MGGP_SVGP, rk.MGGP_RBF(n_groups=3), n_groups=3),
        "MGGP_WSVGP": lambda: gp_of(rgp.MGGP_WSVGP, rk.MGGP_NSF_RBF(L=3, n_groups=3), n_groups=3),
        "GaussianPrior": lambda: rgp.GaussianPrior(y, L=3),
        "GaussianLikelihood": lambda: rl.GaussianLikelihood(gp_of(rgp.SVGP, rk.RBF())),
        "ExactLikelihood": lambda: rl.ExactLikelihood(gp_of(rgp.WSVGP, rk.RBF())),
        "PNMF": lambda: rl.PNMF(rgp.GaussianPrior(y, L=3), y, L=3),
        "NSF2": lambda: rl.NSF2(gp_of(rgp.WSVGP, rk.NSF_RBF(L=3)), y, L=3),
        "NSF": lambda: rl.NSF(gp_of(rgp.WSVGP, rk.NSF_RBF(L=3)), y, L=3),
        "MGGP_NSF": lambda: rl.MGGP_NSF(gp_of(rgp.MGGP_WSVGP, rk.MGGP_NSF_RBF(L=3, n_groups=3), n_groups=3), y, L=3),
        "Hybrid_NSF2": lambda: rl.Hybrid_NSF2(gp_of(rgp.WSVGP, rk.NSF_RBF(L=3)), rgp.GaussianPrior(y, L=2), y, L=3, T=2),
        "Hybrid_NSF_Exact": lambda: rl.Hybrid_NSF_Exact(gp_of(rgp.WSVGP, rk.NSF_RBF(L=3)), rgp.GaussianPrior(y, L=2), y, L=3, T=2),
        "Hybrid_NSF": lambda: rl.Hybrid_NSF(gp_of(rgp.WSVGP, rk.NSF_RBF(L=3)), y, L=3, non_spatial_factors=2),
    }
    table = {}
    for name, make in builders.items():
        torch.manual_seed(0)
        m = make()
        table[name] = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
        print(f"state_dict {name}: {list(table[name])}")
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    # call signatures (parameter names, order, defaults) of the public API the notebooks use
    import inspect
    import gpzoo.utilities as ru

    def sig(f):
        out = []
        for n, q in inspect.signature(f).parameters.items():
            d = None if q.default is inspect.Parameter.empty else repr(q.default)
            out.append([n, str(q.kind), d])
        return out
    sigs = {}
    for mod, names in ((rk, ["RBF", "NSF_RBF", "batched_RBF", "batched_Matern32", "MGGP_RBF", "MGGP_NSF_RBF", "batched_MGGP_RBF"]),
                       (rgp, ["VNNGP", "SVGP", "WSVGP", "MGGP_SVGP", "MGGP_WSVGP", "GaussianPrior"]),
                       (rl, ["GaussianLikelihood", "ExactLikelihood", "PoissonFactorization", "PNMF", "NSF2", "NSF", "MGGP_NSF",
                             "Hybrid_NSF2", "Hybrid_NSF_Exact", "Hybrid_NSF"])):
        for n in names:
            cls = getattr(mod, n)
            sigs[f"{mod.__name__}.{n}.__init__"] = sig(cls.__init__)
            for meth in ("forward", "forward_batched", "forward_precomputed", "forward_kernels", "kernel_forward",
                         "forward_distance", "covariance", "set_group_distances", "get_rate"):
                if meth in vars(cls) or any(meth in vars(b) for b in cls.__mro__[1:-2]):
                    sigs[f"{mod.__name__}.{n}.{meth}"] = sig(getattr(cls, meth))
    for n in ("add_jitter", "whitened_KL", "svgp_forward", "_squared_dist", "_embed_distance_matrix", "reshape_param",
              "train", "train_batched", "train_hybrid", "train_hybrid_batched", "train_closure_batched"):
        sigs[f"gpzoo.utilities.{n}"] = sig(getattr(ru, n))
    with open(os.path.join(HERE, "api_signatures.json"), "w") as f:
        json.dump(sigs, f, indent=1, sort_keys=True)
    print(f"{len(sigs)} signatures")
    # a checkpoint written by the reference (plain state_dict of tensors; loads with weights_only=True)
    torch.manual_seed(5)
    L, M, N = 3, 24, 90
    gp = rgp.WSVGP(rk.NSF_RBF(sigma=1.1, lengthscale=2.5, L=L), dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(4 * torch.randn(M, 2)); gp.mu = nn.Parameter(torch.randn(L, M)); gp.Lu = nn.Parameter(0.1 * torch.randn(L, M, M))
    model = rl.ExactLikelihood(gp, noise=0.3).double()
    X = 4 * torch.randn(N, 2, dtype=torch.float64)
    with torch.no_grad():
        pY, qF, qU, _ = model(X=X)
    torch.save(model.state_dict(), os.path.join(HERE, "ref_checkpoint_exact_wsvgp.pt"))
    np.savez_compressed(os.path.join(HERE, "ref_checkpoint_exact_wsvgp_out.npz"), X=X.numpy(), mean=qF.mean.numpy(),
                        scale=qF.scale.numpy(), pY_scale=pY.scale.numpy(), jitter=np.float64(1e-2))


def trajectory_case():
    """Twelve Adam steps of the reference's own `utilities.train` (utilities.py:471-493) on SVGP +
    NSF_RBF + GaussianLikelihood with every parameter trainable and the rsample noise replayed from
    stored eps: losses per step and the final parameters (end-to-end forward + backward + optimiser)."""
    import contextlib
    import io
    import torch.distributions.normal as tdn
    from gpzoo.utilities import train as ref_train
    steps, E, L, N, M = 12, 4, 3, 120, 20
    inp = make_inputs(900, N=N, M=M, d=2, L=L)
    eps = torch.randn(steps, E, L, N, generator=torch.Generator().manual_seed(901), dtype=torch.float64)
    kern = rk.NSF_RBF(sigma=1.2, lengthscale=2.5, L=L)
    gp = rgp.SVGP(kern, dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(inp["Z"].clone()); gp.mu = nn.Parameter(inp["mu"].clone()); gp.Lu = nn.Parameter(inp["Lu_raw"].clone())
    model = rl.GaussianLikelihood(gp, noise=0.3).double()
    init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
    queue = [e for e in eps]
    orig = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: queue.pop(0).to(dtype)
    try:
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        with contextlib.redirect_stderr(io.StringIO()):
            losses = ref_train(model, opt, inp["X"], inp["y"], torch.device("cpu"), steps=steps, E=E)
    finally:
        tdn._standard_normal = orig
    rec = {"init." + k: v for k, v in init.items()}
    rec.update({"final." + k: v.detach().numpy() for k, v in model.state_dict().items()})
    np.savez_compressed(os.path.join(HERE, "ref_trajectory_svgp_f64.npz"), X=inp["X"].numpy(), y=inp["y"].numpy(),
                        eps=eps.numpy(), losses=np.array(losses), lr=np.float64(1e-2), jitter=np.float64(1e-2),
                        noise0=np.float64(0.3), **rec)
    print("trajectory losses:", [round(v, 4) for v in losses])


def poisson_trajectory_cases():
    """The reference's mini-batch drivers end to end: `train_batched` (utilities.py:600-632) on NSF2 and
    `train_hybrid_batched` (utilities.py:497-527) on Hybrid_NSF, both over an SVGP + NSF_RBF prior, every
    parameter trainable, with the index draws (torch.multinomial) and the rsample noise replayed from the
    fixture: losses per step and final parameters."""
    import contextlib
    import io
    import torch.distributions.normal as tdn
    import gpzoo.utilities as ru
    steps, E, L, T, N, Nb, M, D = 8, 3, 3, 2, 150, 60, 16, 12
    inp = make_inputs(950, N=N, M=M, d=2, L=L)
    g = torch.Generator().manual_seed(951)
    y = torch.poisson(3.0 * torch.rand(D, N, generator=g, dtype=torch.float64), generator=g)
    idxs = torch.stack([torch.randperm(N, generator=g)[:Nb] for _ in range(steps)])
    eps1 = torch.randn(steps, E, L, Nb, generator=g, dtype=torch.float64)
    eps2 = torch.randn(steps, E, T, Nb, generator=g, dtype=torch.float64)
    for name in ("nsf2", "hybrid_nsf"):
        torch.manual_seed(7)
        kern = rk.NSF_RBF(sigma=1.0, lengthscale=3.0, L=L)
        gp = rgp.SVGP(kern, dim=2, M=M, jitter=1e-2)
        gp.Z = nn.Parameter(inp["Z"].clone()); gp.mu = nn.Parameter(inp["mu"].clone()); gp.Lu = nn.Parameter(inp["Lu_raw"].clone())
        if name == "nsf2":
            model = rl.NSF2(gp, y, L=L).double()
            queue = [e for e in eps1]
            loop = ru.train_batched
        else:
            model = rl.Hybrid_NSF(gp, y, L=L, non_spatial_factors=T).double()
            queue = [e for pair in zip(eps1, eps2) for e in pair]
            loop = ru.train_hybrid_batched
        init = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
        iq = [i for i in idxs]
        o_norm, o_multi = tdn._standard_normal, torch.multinomial
        tdn._standard_normal = lambda shape, dtype, device: queue.pop(0).to(dtype)
        torch.multinomial = lambda *a, **k: iq.pop(0)
        try:
            opt = torch.optim.Adam(model.parameters(), lr=1e-2)
            with contextlib.redirect_stderr(io.StringIO()):
                losses = loop(model, opt, inp["X"], y, torch.device("cpu"), steps=steps, E=E, batch_size=Nb)
        finally:
            tdn._standard_normal, torch.multinomial = o_norm, o_multi
        rec = {"init." + k: v for k, v in init.items()}
        rec.update({"final." + k: v.detach().numpy() for k, v in model.state_dict().items()})
        np.savez_compressed(os.path.join(HERE, f"ref_trajectory_{name}_f64.npz"), X=inp["X"].numpy(), y=y.numpy(),
                            idx=idxs.numpy(), eps1=eps1.numpy(), eps2=eps2.numpy(), losses=np.array(losses),
                            lr=np.float64(1e-2), jitter=np.float64(1e-2), **rec)
        print(f"trajectory {name} losses:", [round(v, 3) for v in losses])


def scalar_kernel_batched_q_cases():
    """A scalar-parameter RBF shared by L variational posteriors (mu (L,M), Lu (L,M,M) assigned after
    construction): the reference broadcasts the single kernel matrix over the latents."""
    inp = make_inputs(970, N=70, M=18, d=2, L=3)
    for cls_name in ("WSVGP", "SVGP"):
        gp = getattr(rgp, cls_name)(rk.RBF(sigma=1.2, lengthscale=2.5), dim=2, M=18, jitter=1e-2)
        gp.Z = nn.Parameter(inp["Z"].clone()); gp.mu = nn.Parameter(inp["mu"].clone()); gp.Lu = nn.Parameter(inp["Lu_raw"].clone())
        gp = gp.double()
        qF, qU, pU = gp(inp["X"])
        loss = (qF.mean * inp["y"]).sum() + (qF.scale ** 2).sum()
        loss.backward()
        np.savez_compressed(os.path.join(HERE, f"extra_scalar_rbf_batched_{cls_name.lower()}_f64.npz"),
                            X=inp["X"].numpy(), y=inp["y"].numpy(), Z=inp["Z"].numpy(), mu=inp["mu"].numpy(), Lu_raw=inp["Lu_raw"].numpy(),
                            mean=qF.mean.detach().numpy(), scale=qF.scale.detach().numpy(), grad_mu=gp.mu.grad.numpy(),
                            grad_Lu=gp.Lu.grad.numpy(), grad_Z=gp.Z.grad.numpy(), grad_sigma=gp.kernel.sigma.grad.numpy(),
                            grad_lengthscale=gp.kernel.lengthscale.grad.numpy(), jitter=np.float64(1e-2))
        print(cls_name, "scalar RBF, batched q:", tuple(qF.mean.shape), float(loss))


if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "extra"):
    scalar_kernel_batched_q_cases()

if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "trajectory"):
    trajectory_case()
    poisson_trajectory_cases()

if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "state_dict"):
    state_dict_cases()

if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_POISSON", "1") == "1" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "poisson", "vnngp"):
    if os.environ.get("GPZ_GOLDEN_ONLY", "") != "vnngp":
        poisson_cases()
    if os.environ.get("GPZ_GOLDEN_ONLY", "") != "poisson":
        vnngp_cases()


def kernel_grad_cases():
    """Stand-alone kernel calls differentiated by the reference's own autograd (kernels are ordinary traced
    modules there, kernels.py:114-130, 139-155, 176-228): K = kernel(X, Z[, groups]), loss = sum(K * R) for a
    fixed R, gradients w.r.t. sigma, lengthscale, group_diff_param, X and Z; plus K(X, X) with the same tensor
    in both slots.  -> kernel_grads.npz"""
    out = {}
    inp = make_inputs(311, N=50, M=14, d=2, L=3, n_groups=3)
    R = gen(312, 3, 50, 14)
    R2 = gen(313, 3, 50, 50)
    out["gX"], out["gZ"] = inp["gX"].numpy(), inp["gZ"].numpy()
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        out[f"{tag}.X"], out[f"{tag}.Z"] = inp["X"].to(dtype).numpy(), inp["Z"].to(dtype).numpy()
        out[f"{tag}.R"], out[f"{tag}.R2"] = R.to(dtype).numpy(), R2.to(dtype).numpy()

        def vec(v):
            return nn.Parameter(torch.tensor(v, dtype=dtype))

        def make(kind):
            if kind == "rbf":
                return rk.RBF(sigma=1.2, lengthscale=3.0).to(dtype), False
            if kind == "nsf_rbf":
                return build_kernel("nsf_rbf", 3).to(dtype), False
            if kind == "batched_rbf_vec":
                k = rk.batched_RBF(); k.sigma, k.lengthscale = vec([1.0, 0.8, 1.3]), vec([2.5, 4.0, 6.0])
                return k, False
            if kind == "batched_rbf_scalar":
                return rk.batched_RBF(sigma=1.2, lengthscale=3.0).to(dtype), False
            if kind == "matern32_vec":
                k = rk.batched_Matern32(); k.sigma, k.lengthscale = vec([1.0, 0.8, 1.3]), vec([2.5, 4.0, 6.0])
                return k, False
            if kind == "matern32_scalar":
                return rk.batched_Matern32(sigma=0.9, lengthscale=2.0).to(dtype), False
            if kind == "mggp_rbf":
                k = rk.MGGP_RBF(sigma=1.1, lengthscale=3.5, group_diff_param=0.6, n_groups=3).to(dtype)
                k.embedding = k.embedding.to(dtype)
                return k, True
            if kind == "mggp_nsf_rbf":
                return build_kernel("mggp_nsf_rbf", 3).to(dtype), True
            if kind == "batched_mggp_rbf":
                return rk.batched_MGGP_RBF(sigma=1.1, lengthscale=3.5, group_diff_param=-0.6, n_groups=3).to(dtype), True
            raise ValueError(kind)

        for kind in ("rbf", "nsf_rbf", "batched_rbf_vec", "batched_rbf_scalar", "matern32_vec", "matern32_scalar",
                     "mggp_rbf", "mggp_nsf_rbf", "batched_mggp_rbf"):
            k, mggp = make(kind)
            X = inp["X"].to(dtype).clone().requires_grad_(True)
            Z = inp["Z"].to(dtype).clone().requires_grad_(True)
            K = k(X, Z, inp["gX"], inp["gZ"]) if mggp else k(X, Z)
            Rk = R.to(dtype) if K.dim() == 3 else R.to(dtype)[0]
            (K * Rk).sum().backward()
            pre = f"{tag}.{kind}."
            out[pre + "K"] = K.detach().numpy()
            out[pre + "grad_X"], out[pre + "grad_Z"] = X.grad.numpy(), Z.grad.numpy()
            out[pre + "grad_sigma"], out[pre + "grad_lengthscale"] = k.sigma.grad.numpy(), k.lengthscale.grad.numpy()
            if mggp:
                out[pre + "grad_group_diff_param"] = k.group_diff_param.grad.numpy()
                out[pre + "embedding"] = k.embedding.detach().numpy()
            if "matern" in kind:
                continue        # k(X, X) has r = 0 on the diagonal: NaN gradients in the reference
            k2, _ = make(kind)
            X2 = inp["X"].to(dtype).clone().requires_grad_(True)
            K2 = k2(X2, X2, inp["gX"], inp["gX"]) if mggp else k2(X2, X2)
            Rk2 = R2.to(dtype) if K2.dim() == 3 else R2.to(dtype)[0]
            (K2 * Rk2).sum().backward()
            out[pre + "xx.K"] = K2.detach().numpy()
            out[pre + "xx.grad_X"] = X2.grad.numpy()
            out[pre + "xx.grad_sigma"], out[pre + "xx.grad_lengthscale"] = k2.sigma.grad.numpy(), k2.lengthscale.grad.numpy()
            if mggp:
                out[pre + "xx.grad_group_diff_param"] = k2.group_diff_param.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "kernel_grads.npz"), **out)
    print(f"kernel_grads.npz: {len(out)} arrays")


def exact_gp_case():
    """The inline ExactGP of exact_mggp.ipynb (forward: MultivariateNormal(0, kernel(X, X, g, g) + noise^2 I);
    train: loss = -log_prob(y), loss.backward()) with MGGP_RBF, one step: loss and the gradients the
    reference's autograd sends to sigma, lengthscale, group_diff_param and the noise.  -> exact_mggp_step_f64.npz"""
    inp = make_inputs(411, N=40, M=4, d=1, L=0, n_groups=2, span=5.0)
    X, gX = inp["X"], inp["gX"]
    y = torch.sin(X[:, 0]) + 0.3 * gX.double() + 0.1 * gen(412, 40)
    kernel = rk.MGGP_RBF(sigma=1.3, lengthscale=1.7, group_diff_param=0.8, n_groups=2).double()
    kernel.embedding = kernel.embedding.double()
    kernel.input_dim = 1                                   # the 1-D notebooks overwrite it (mggp_test.ipynb:54)
    noise = nn.Parameter(torch.tensor(0.4, dtype=torch.float64))
    Kxx = kernel.forward(X, X, gX, gX)
    pY = distributions.MultivariateNormal(torch.zeros(40, dtype=torch.float64), Kxx + (noise ** 2) * torch.eye(40, dtype=torch.float64))
    loss = -pY.log_prob(y).sum()
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "exact_mggp_step_f64.npz"), X=X.numpy(), gX=gX.numpy(), y=y.numpy(),
                        embedding=kernel.embedding.numpy(), Kxx=Kxx.detach().numpy(), loss=np.float64(float(loss)),
                        grad_sigma=kernel.sigma.grad.numpy(), grad_lengthscale=kernel.lengthscale.grad.numpy(),
                        grad_group_diff_param=kernel.group_diff_param.grad.numpy(), grad_noise=noise.grad.numpy())
    print("exact_mggp_step_f64: loss", float(loss))


def vnngp_scale_case(tag="f32", N=4000, M=500):
    """VNNGP neighbour bookkeeping at Slide-seq-like coordinates (|x| <= 100, K=8): the reference orders neighbours by
    torch.cdist's matmul-expansion distances (kernels.py:118, gp.py:31, 64; either side > 25 rows), whose fp32 error
    reaches 0.06 there.  Stores the reference's neighbour table and moments.
    -> vnngp_scale_f32.npz (N=4000, M=500) and vnngp_scale_f64.npz (N=1500, M=120: the fp64 instantiation of the
    matmul-expansion ranking)"""
    import contextlib
    import io
    L, K = 2, 8
    dt = torch.float32 if tag == "f32" else torch.float64
    inp = make_inputs(811, N=N, M=M, d=2, L=L, span=100.0)
    kern = rk.NSF_RBF(L=L)
    kern.sigma = nn.Parameter(per_latent([1.0, 0.8], True)); kern.lengthscale = nn.Parameter(per_latent([6.0, 9.0], True))
    gp = rgp.VNNGP(kern, dim=2, M=M, K=K, jitter=1e-2)
    gp.Z = nn.Parameter(inp["Z"].clone()); gp.mu = nn.Parameter(inp["mu"].clone()); gp.Lu = nn.Parameter(inp["Lu_raw"].clone())
    gp = gp.to(dt)
    X = inp["X"].to(dt)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        qF, qU, pU = gp(X)
        _, dist = kern(X, gp.Z, return_distance=True)
    idx = torch.argsort(dist, dim=1)[:, :K]
    np.savez_compressed(os.path.join(HERE, f"vnngp_scale_{tag}.npz"), X=X.numpy(), Z=gp.Z.detach().numpy(),
                        mu=gp.mu.detach().numpy(), Lu_raw=gp.Lu.detach().numpy(), sigma=kern.sigma.detach().numpy(),
                        lengthscale=kern.lengthscale.detach().numpy(), idx=idx.numpy().astype(np.int16),
                        dist_k=torch.gather(dist, 1, idx).numpy(), mean=qF.mean.numpy(), scale=qF.scale.numpy(),
                        jitter=np.float64(1e-2), K=np.int64(K))
    print(f"vnngp_scale_{tag}: idx", tuple(idx.shape), "mean[0,:3]", qF.mean[0, :3].tolist())


if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "kernel_grads"):
    kernel_grad_cases()
    exact_gp_case()

if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "vnngp_scale"):
    vnngp_scale_case()
    vnngp_scale_case("f64", N=1500, M=120)


def multiblock_cases():
    """The reference itself beyond one 128-block: M = 300 (three blocks: panel solves, a trailing update, one level of the
    triangular inverse, partial wide tiles), N = 2000, L = 3.  The inputs are regenerated from their seed by the tests
    (tests/golden/inputs.py); stored are (L,N) / (L,M) outputs and scalars only (~100 KB per case)."""
    for prefix, meta0 in (("multiblock", dict(seed=4242, N=2000, M=300, d=2, L=3, span=25.0)),
                          # nine blocks, an odd count: four levels of the triangular inverse with a ragged tail segment, four
                          # paired trailing updates + a last single panel, odd row-tile counts in every product kernel
                          ("multiblock9", dict(seed=4343, N=1500, M=1100, d=2, L=2, span=40.0))):
      cases = [("wsvgp_matern32", "WSVGP", "matern32", True, 0), ("svgp_nsf_rbf", "SVGP", "nsf_rbf", False, 0)]
      if prefix == "multiblock":       # the int64 group gather and the (latent, group-pair) table beyond one block
          cases.append(("mggp_wsvgp_mggp_nsf_rbf", "MGGP_WSVGP", "mggp_nsf_rbf", True, 3))
      for name, gpc, kind, whitened, n_groups in cases:
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            meta = dict(meta0, n_groups=n_groups)
            inp = make_inputs(meta["seed"], N=meta["N"], M=meta["M"], d=meta["d"], L=meta["L"], n_groups=n_groups, span=meta["span"])
            kern = build_kernel(kind, meta["L"])
            out = run_case(name, getattr(rgp, gpc), kern, inp, dtype, 1e-2, 0.5, whitened, n_groups > 0)
            keep = {k: out[k] for k in ("mean", "scale", "kl", "elbo", "grad_mu", "sigma", "lengthscale", "jitter", "noise_sd")}
            if n_groups:
                keep.update({k: out[k] for k in ("embedding", "group_diff", "input_dim", "grad_group_diff")})
            keep["chol_diag"] = np.diagonal(out["chol"], axis1=-2, axis2=-1).copy()
            keep["chol_rowsum"] = out["chol"].sum(-1)
            keep["grad_Lu_rowsum"] = out["grad_Lu"].sum(-1)
            keep["grad_Lu_diag"] = np.diagonal(out["grad_Lu"], axis1=-2, axis2=-1).copy()
            keep["grad_Lu_absmax"] = np.float64(np.abs(out["grad_Lu"]).max())
            keep.update({k: np.float64(v) if isinstance(v, float) else np.int64(v) for k, v in meta.items()})
            keep["kind"] = np.array(kind); keep["whitened"] = np.array(whitened)
            np.savez_compressed(os.path.join(HERE, f"{prefix}_{name}_{tag}.npz"), **keep)
            print(f"{prefix}_{name}_{tag}: elbo={float(out['elbo']):.10f}")


if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "multiblock"):
    multiblock_cases()


def hybrid_exact_case():
    """Hybrid_NSF_Exact (reference likelihoods.py:167-222): the rate is built from the log-normal mean
    exp(m + s^2/2) -- no sampling, hence NO sample axis: pY.rate is (D,N) and the objective of the reference's
    own loops, `pY.log_prob(y).mean(axis=0).sum()` (utilities.py:537) / `(y log r - r).mean(axis=0).sum()`
    (utilities.py:510), averages over the GENE axis.  Stored: both objectives as those lines compute them and the
    gradients of the first one's loss (train_hybrid's, with both KL terms) through the reference's autograd."""
    N, M, L, T, D = 160, 36, 3, 2, 25
    inp = make_inputs(900, N=N, M=M, d=2, L=L)
    g = torch.Generator().manual_seed(911)
    y = torch.poisson(3.0 * torch.rand(D, N, generator=g), generator=g).float()
    kern = build_kernel("nsf_rbf", L).float()
    gp = rgp.WSVGP(kern, dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(inp["Z"].float(), requires_grad=False)
    gp.mu = nn.Parameter(0.2 * inp["mu"].float())
    gp.Lu = nn.Parameter(inp["Lu_raw"].float())
    for t in kern.parameters():
        t.requires_grad_(False)
    torch.manual_seed(11)
    prior = rgp.GaussianPrior(y, L=T)
    model = rl.Hybrid_NSF_Exact(gp, prior, y, L=L, T=T)
    with torch.no_grad():
        model.V.copy_(0.3 + torch.rand(N, generator=g))
        prior.mean.copy_(0.3 * torch.randn(T, N, generator=g))
    X = inp["X"].float()
    pY, qF1, qU, pU, qF2, pF2 = model(X=X, E=7)
    assert pY.rate.shape == (D, N) and pU is None
    loglik = pY.log_prob(y).mean(axis=0).sum()                     # utilities.py:537
    kl = torch.stack([whitened_KL(qU.mean[l], qU.scale_tril[l]) for l in range(L)]).sum()
    loss = -(loglik - kl - distributions.kl_divergence(qF2, pF2).sum())
    loss.backward()
    idx = torch.arange(0, N, 3)
    with torch.no_grad():
        pYb = model.forward_batched(X=X, idx=idx, E=7)[0]
        loglik_b = (y[:, idx] * torch.log(pYb.rate) - pYb.rate).mean(axis=0).sum()   # utilities.py:508-510
    rec = dict(X=X.numpy(), Z=inp["Z"].float().numpy(), mu=gp.mu.detach().numpy(), Lu_raw=gp.Lu.detach().numpy(),
               y=y.numpy(), sigma=kern.sigma.detach().numpy(), lengthscale=kern.lengthscale.detach().numpy(),
               W=model.sf.W.detach().numpy(), W2=model.cf.W.detach().numpy(), V=model.V.detach().numpy(),
               mean2=prior.mean.detach().numpy(), scale2=prior.scale.detach().numpy(),
               loglik=np.float64(float(loglik)), loss=np.float64(float(loss)), rate=pY.rate.detach().numpy(),
               idx_b=idx.numpy(), loglik_b=np.float64(float(loglik_b)),
               grad_W=model.sf.W.grad.numpy(), grad_W2=model.cf.W.grad.numpy(), grad_V=model.V.grad.numpy(),
               grad_mu=gp.mu.grad.numpy(), grad_Lu=gp.Lu.grad.numpy(), grad_mean2=prior.mean.grad.numpy(),
               grad_scale2=prior.scale.grad.numpy())
    np.savez_compressed(os.path.join(HERE, "poisson_hybrid_nsf_exact_f32.npz"), **rec)
    print(f"poisson_hybrid_nsf_exact_f32: loglik={float(loglik):.6f} loglik_b={float(loglik_b):.6f}")


if __name__ == "__main__" and os.environ.get("GPZ_GOLDEN_ONLY", "") in ("", "hybrid_exact"):
    hybrid_exact_case()
