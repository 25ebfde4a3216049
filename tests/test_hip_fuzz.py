"""GPU: seeded random sweep over shapes, kernel families, GP variants and precisions -- forward
moments, ELBO and the mu / Lu (and, whitened, kernel hyper-parameter) gradients against the CPU oracle
and torch autograd through it.  Ragged extents (N, M not multiples of 128, single points, single
latents, 1-D to 3-D inputs) are drawn on purpose."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

KINDS = ["rbf_scalar", "nsf_rbf", "matern32", "mggp_nsf_rbf"]


def draw(seed):
    g = torch.Generator().manual_seed(10_000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))   # noqa: E731
    kind = KINDS[seed % len(KINDS)]
    N = [1, 5, 127, 128, 129, 300, 517, 700, 2100, 4200][ri(0, 9)]   # 4200: 33 column tiles = two strips of 9 x 2 and 8 x 2 - 1
    M = [1, 2, 31, 64, 128, 129, 200, 257][ri(0, 7)]
    L = 1 if kind == "rbf_scalar" else ri(1, 5)
    d = ri(1, 3)
    f64 = seed % 3 != 0
    whitened = bool(ri(0, 1))
    dt = torch.float64
    X = (torch.rand(N, d, generator=g, dtype=dt) - 0.5) * 20
    Z = (torch.rand(M, d, generator=g, dtype=dt) - 0.5) * 20
    c = dict(kind=kind, N=N, M=M, L=L, d=d, f64=f64, whitened=whitened, X=X, Z=Z,
             sigma=0.6 + torch.rand(L, generator=g, dtype=dt), lengthscale=1.5 + 4 * torch.rand(L, generator=g, dtype=dt),
             mu=torch.randn(L, M, generator=g, dtype=dt), Lu_raw=0.1 * torch.randn(L, M, M, generator=g, dtype=dt),
             y=torch.randn(L, N, generator=g, dtype=dt), jitter=[1e-2, 1e-1][ri(0, 1)], noise_sd=0.4)
    if kind == "mggp_nsf_rbf":
        G = ri(2, 4)
        c.update(G=G, gX=torch.randint(0, G, (N,), generator=g), gZ=torch.randint(0, G, (M,), generator=g),
                 group_diff=0.3 + torch.rand(L, generator=g, dtype=dt))
    return c


def oracle_parts(c, leaf):
    from oracle import svgp_oracle as O
    okind = {"rbf_scalar": "nsf_rbf"}.get(c["kind"], c["kind"])
    kw = {}
    if c["kind"] == "mggp_nsf_rbf":
        G = c["G"]
        kw = dict(embedding=O.embed_group_distances(torch.ones(G, G) - torch.eye(G)).double(), group_diff=leaf["group_diff"])
    K = lambda A, B, gA=None, gB=None: O.kernel_matrix(okind, A, B, leaf["sigma"], leaf["lengthscale"], gA=gA, gB=gB, **kw)  # noqa: E731
    Kzx = K(leaf["Z"], c["X"], c.get("gZ"), c.get("gX"))
    Kzz = K(leaf["Z"], leaf["Z"], c.get("gZ"), c.get("gZ")) + c["jitter"] * torch.eye(c["M"], dtype=torch.float64)
    Kxx = (leaf["sigma"] ** 2)[:, None].expand(-1, c["N"])
    if c["whitened"]:
        mean, scale, Lu, chol = O.wsvgp_moments(Kxx, Kzx, Kzz, leaf["mu"], leaf["Lu_raw"])
        kl = O.whitened_kl(leaf["mu"], Lu)
    else:
        mean, scale, Lu, chol = O.svgp_moments(Kxx, Kzx, Kzz, leaf["mu"], leaf["Lu_raw"], 1e-6)
        kl = O.mvn_kl(leaf["mu"], Lu, chol)
    return mean, scale, kl, O.gaussian_elbo(c["y"], mean, scale, c["noise_sd"], kl)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GPZ_FUZZ_SEEDS", "64"))))
def test_random_case(seed):
    from gpzoo_amd import _lib, ops
    from gpzoo_amd.ops import KernelSpec
    from oracle import svgp_oracle as O
    c = draw(seed)
    dt = torch.float64 if c["f64"] else torch.float32
    cu = lambda t: t.to(dt).cuda()   # noqa: E731
    kid = {"rbf_scalar": _lib.KERNEL_RBF, "nsf_rbf": _lib.KERNEL_RBF, "matern32": _lib.KERNEL_MATERN32,
           "mggp_nsf_rbf": _lib.KERNEL_MGGP_RBF}[c["kind"]]
    extra = {}
    if c["kind"] == "mggp_nsf_rbf":
        emb = O.embed_group_distances(torch.ones(c["G"], c["G"]) - torch.eye(c["G"])).double()
        r2 = ((emb[:, None, :] - emb[None, :, :]) ** 2).sum(-1)
        spec = KernelSpec(kid, cu(c["sigma"]), cu(c["lengthscale"]), True, cu(c["group_diff"] ** 2), cu(r2), 1.0)
        extra = dict(gX=c["gX"].cuda(), gZ=c["gZ"].cuda())
    else:
        spec = KernelSpec(kid, cu(c["sigma"]), cu(c["lengthscale"]), c["kind"] != "rbf_scalar")
    args = (spec, cu(c["X"]), cu(c["Z"]), cu(c["mu"]), cu(c["Lu_raw"]), c["jitter"], c["whitened"])

    matern_grad = c["kind"] == "matern32"      # sqrt at r = 0: the oracle's autograd gives NaN w.r.t. Z there
    names = ["mu", "Lu_raw"] + ([] if matern_grad else ["Z", "sigma", "lengthscale"]) + \
            (["group_diff"] if c["kind"] == "mggp_nsf_rbf" else [])
    leaf = {k: (c[k].clone().requires_grad_(k in names)) for k in ("Z", "sigma", "lengthscale", "mu", "Lu_raw")}
    if "group_diff" in c:
        leaf["group_diff"] = c["group_diff"].clone().requires_grad_(True)
    try:
        mean, scale, kl, elbo = oracle_parts(c, leaf)
    except torch.linalg.LinAlgError as e:
        # e.g. the multi-group kernel with input_dim = 2 on 3-D inputs is not a valid covariance: the HIP
        # factorisation must fail the way torch's does, naming the same matrix and leading minor
        with pytest.raises(torch.linalg.LinAlgError) as mine:
            ops.svgp_forward(*args, y=cu(c["y"]), noise_sd=c["noise_sd"], **extra)
        if c["f64"]:
            assert str(mine.value) == str(e)
        return
    out = ops.svgp_forward(*args, y=cu(c["y"]), noise_sd=c["noise_sd"], **extra)

    rt = 1e-5 if c["f64"] else 1e-3
    tag = {k: c[k] for k in ("kind", "N", "M", "L", "d", "f64", "whitened", "jitter")}
    # scale and kl are strictly positive: PURE rtol (atol = 0); everything that changes sign is norm-wise (rt * max|.|)
    close = lambda a, b, what: torch.testing.assert_close(  # noqa: E731
        a.double().cpu().reshape(b.shape), b.detach(), rtol=rt,
        atol=0.0 if what in ("scale", "kl") else rt * max(float(b.detach().abs().max()), 1e-30),
        msg=lambda m: f"{what} {tag}: {m}")
    close(out["mean"], mean, "mean")
    close(out["scale"], scale, "scale")
    close(out["kl"], kl, "kl")
    assert float(out["elbo"]) == pytest.approx(float(elbo.detach()), rel=rt), tag

    gen = torch.Generator().manual_seed(seed)
    gm = torch.randn(mean.shape, generator=gen, dtype=torch.float64)
    gs = torch.randn(scale.shape, generator=gen, dtype=torch.float64)
    ((mean * gm).sum() + (scale * gs).sum()).backward()
    kg = c["whitened"] and not matern_grad
    res = ops.svgp_backward(*args, cu(gm), cu(gs), out["scale"], kernel_grads=kg, **extra)
    if not c["f64"]:
        rt = 5e-3 if not c["whitened"] else 2e-3     # fp32 gradients: cancellation in the un-whitened W (S - Kzz) W^T term
    close(res[0], leaf["mu"].grad, "grad_mu")
    close(res[1], leaf["Lu_raw"].grad, "grad_Lu")
    if kg:
        close(res[2][:, 0], leaf["sigma"].grad, "grad_sigma")
        close(res[2][:, 1], leaf["lengthscale"].grad, "grad_lengthscale")
        close(res[3], leaf["Z"].grad, "grad_Z")
        if "group_diff" in c:
            close(res[2][:, 2] * 2 * cu(c["group_diff"]).double(), leaf["group_diff"].grad, "grad_group_diff")
