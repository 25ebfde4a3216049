"""GPU: the Poisson NSF factor models (SURVEY §8f "next" #2) -- the API mirrors and the fused
expected log-likelihood -- against the reference's own outputs and autograd gradients, with the
reference's rsample noise replayed."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else float(z[k])) for k in z.files}


def build(c, hybrid):
    from gpzoo.gp import GaussianPrior, WSVGP
    from gpzoo.kernels import NSF_RBF
    from gpzoo.likelihoods import Hybrid_NSF2, NSF2
    L, M = c["mu"].shape
    k = NSF_RBF(L=L)
    k.sigma = nn.Parameter(c["sigma"].clone(), requires_grad=False)
    k.lengthscale = nn.Parameter(c["lengthscale"].clone(), requires_grad=False)
    gp = WSVGP(k, dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(c["Z"].clone(), requires_grad=False)
    gp.mu = nn.Parameter(c["mu"].clone())
    gp.Lu = nn.Parameter(c["Lu_raw"].clone())
    y = c["y"]
    if hybrid:
        prior = GaussianPrior(y, L=c["W2"].shape[1])
        prior.mean = nn.Parameter(c["mean2"].clone())
        prior.scale = nn.Parameter(c["scale2"].clone())
        model = Hybrid_NSF2(gp, prior, y, L=L, T=c["W2"].shape[1])
        model.sf.W = nn.Parameter(c["W"].clone())
        model.cf.W = nn.Parameter(c["W2"].clone())
    else:
        model = NSF2(gp, y, L=L)
        model.W = nn.Parameter(c["W"].clone())
    model.V = nn.Parameter(c["V"].clone())
    return model.cuda()


def close(got, ref, rt=2e-3):
    sc = float(ref.abs().max()) + 1e-30
    torch.testing.assert_close(got.cpu(), ref, rtol=rt, atol=rt * sc)


@pytest.mark.parametrize("name,hybrid", [("poisson_nsf2_f32", False), ("poisson_hybrid_nsf2_f32", True)])
def test_fused_expected_loglik_and_gradients_match_reference(name, hybrid):
    from gpzoo.utilities import whitened_KL_batched
    c = load(name)
    model = build(c, hybrid)
    X, y = c["X"].cuda(), c["y"].cuda()
    eps = torch.cat([c["eps1"], c["eps2"]], dim=1).cuda() if hybrid else c["eps1"].cuda()
    res = model.expected_loglik(X, y, E=3, eps=eps)
    ll, qU = res[0], res[2]
    assert float(ll.detach()) == pytest.approx(c["loglik"], rel=1e-4)
    loss = -(ll - whitened_KL_batched(qU.mean, qU.scale_tril).sum())
    if hybrid:
        loss = loss + torch.distributions.kl_divergence(res[4], res[5]).sum()
    assert float(loss.detach()) == pytest.approx(c["loss"], rel=1e-4)
    loss.backward()
    gp = model.sf.prior if hybrid else model.prior
    close((model.sf.W if hybrid else model.W).grad, c["grad_W"])
    close(model.V.grad, c["grad_V"])
    close(gp.mu.grad, c["grad_mu"])
    close(gp.Lu.grad, c["grad_Lu"])
    if hybrid:
        close(model.cf.W.grad, c["grad_W2"])
        close(model.cf.prior.mean.grad, c["grad_mean2"])
        close(model.cf.prior.scale.grad, c["grad_scale2"])


def test_hybrid_exact_closed_form_objective_matches_reference():
    """Hybrid_NSF_Exact (likelihoods.py:167-222): no sampling, a (D,N) rate, and the reference loops' `.mean(axis=0)`
    therefore averages over genes.  `forward`'s rate, `expected_loglik` (what train_hybrid / train_hybrid_batched take
    with fused=True) and every gradient against the reference's own run (tests/golden/make_golden.py hybrid_exact_case);
    the fused and the literal form of the loop give the same loss."""
    from gpzoo.gp import GaussianPrior, WSVGP
    from gpzoo.kernels import NSF_RBF
    from gpzoo.likelihoods import Hybrid_NSF_Exact
    from gpzoo.utilities import whitened_KL_batched
    c = load("poisson_hybrid_nsf_exact_f32")
    L, M = c["mu"].shape
    k = NSF_RBF(L=L)
    k.sigma = nn.Parameter(c["sigma"].clone(), requires_grad=False)
    k.lengthscale = nn.Parameter(c["lengthscale"].clone(), requires_grad=False)
    gp = WSVGP(k, dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(c["Z"].clone(), requires_grad=False)
    gp.mu = nn.Parameter(c["mu"].clone()); gp.Lu = nn.Parameter(c["Lu_raw"].clone())
    y = c["y"]
    prior = GaussianPrior(y, L=c["W2"].shape[1])
    prior.mean = nn.Parameter(c["mean2"].clone()); prior.scale = nn.Parameter(c["scale2"].clone())
    model = Hybrid_NSF_Exact(gp, prior, y, L=L, T=c["W2"].shape[1])
    model.sf.W = nn.Parameter(c["W"].clone()); model.cf.W = nn.Parameter(c["W2"].clone())
    model.V = nn.Parameter(c["V"].clone())
    model = model.cuda()
    X, yd = c["X"].cuda(), y.cuda()
    with torch.no_grad():
        pY = model(X=X, E=7)[0]
    assert pY.rate.shape == tuple(c["rate"].shape)                 # (D,N): no sample axis
    close(pY.rate, c["rate"], rt=1e-3)
    assert float(pY.log_prob(yd).mean(0).sum()) == pytest.approx(c["loglik"], rel=1e-4)
    res = model.expected_loglik(X, yd, E=7)
    assert float(res[0].detach()) == pytest.approx(c["loglik"], rel=1e-4)
    loss = -(res[0] - whitened_KL_batched(res[2].mean, res[2].scale_tril).sum()
             - torch.distributions.kl_divergence(res[4], res[5]).sum())
    assert float(loss.detach()) == pytest.approx(c["loss"], rel=1e-4)
    loss.backward()
    close(model.sf.W.grad, c["grad_W"]); close(model.cf.W.grad, c["grad_W2"]); close(model.V.grad, c["grad_V"])
    close(gp.mu.grad, c["grad_mu"]); close(gp.Lu.grad, c["grad_Lu"])
    close(prior.mean.grad, c["grad_mean2"]); close(prior.scale.grad, c["grad_scale2"])
    idx = c["idx_b"].long().cuda()
    with torch.no_grad():
        llb = model.expected_loglik(X, yd[:, idx], idx=idx, with_lgamma=False)[0]
    assert float(llb) == pytest.approx(c["loglik_b"], rel=1e-4)     # train_hybrid_batched's y log r - r form


def test_api_forward_returns_reference_shapes():
    """pY is a real Poisson over the (E, D, N) rate; with the replayed noise its rate sums to the reference's."""
    import torch.distributions.normal as tdn
    c = load("poisson_nsf2_f32")
    model = build(c, False)
    eps = c["eps1"].cuda()
    orig = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: eps.to(dtype)
    try:
        with torch.no_grad():
            pY, qF, qU, pU = model(X=c["X"].cuda(), E=3)
            pYb, *_ = model.forward_batched(c["X"].cuda(), torch.arange(40).cuda(), E=3) if False else (pY,)
    finally:
        tdn._standard_normal = orig
    assert isinstance(pY, torch.distributions.Poisson) and pY.rate.shape == (3, 25, 160) and pU is None
    assert float(pY.rate.double().sum()) == pytest.approx(c["rate_sum"], rel=1e-4)
    ll = pY.log_prob(c["y"].cuda()).mean(0).sum()
    assert float(ll.detach()) == pytest.approx(c["loglik"], rel=1e-4)


def test_fused_matches_torch_formula_at_scale():
    """Slide-seq-like sizes shrunk 10x (D=1770 genes, N=3000 spots, 20 + 10 factors, E=5 > one sample
    group): the fused value and gradients equal the straightforward torch evaluation."""
    from gpzoo_amd import ops
    g = torch.Generator().manual_seed(5)
    D, N, Lt, E = 1770, 3000, 30, 5
    mean = 0.3 * torch.randn(Lt, N, generator=g)
    scale = 0.2 + 0.3 * torch.rand(Lt, N, generator=g)
    eps = torch.randn(E, Lt, N, generator=g)
    W = torch.rand(D, Lt, generator=g) + 0.05
    V = 0.5 + torch.rand(N, generator=g)
    y = torch.poisson(2.0 * torch.rand(D, N, generator=g), generator=g)
    ll, dmean, dscale, dW, dV = ops.poisson_nsf(mean.cuda(), scale.cuda(), eps.cuda(), W.cuda(), V.cuda(), y.cuda())
    lv = [t.double().requires_grad_(True) for t in (mean, scale, W, V)]
    F = lv[0] + lv[1] * eps.double()
    rate = lv[3] * torch.matmul(lv[2], torch.exp(F))
    ref = torch.distributions.Poisson(rate).log_prob(y.double()).mean(0).sum()
    ref.backward()
    assert float(ll.detach()) == pytest.approx(float(ref), rel=2e-5)
    for got, want in zip((dmean, dscale, dW, dV), lv):
        close(got.double(), want.grad, rt=5e-4)


@pytest.mark.parametrize("name", ["nsf2", "hybrid_nsf"])
@pytest.mark.parametrize("fused", [True, False])
def test_minibatch_drivers_reproduce_reference_trajectories(name, fused):
    """The reference's `train_batched` (NSF2) and `train_hybrid_batched` (Hybrid_NSF) ran 8 Adam steps on the
    CPU with their index draws and rsample noise stored in the fixture; the same-named drivers here --
    fused Poisson kernel or the literal pY.log_prob form, HIP GP forward/backward either way -- reproduce
    every loss and the final parameters."""
    import os
    import numpy as np
    import torch.distributions.normal as tdn
    import torch.nn as nn
    from helpers import GOLDEN
    from gpzoo.gp import SVGP
    from gpzoo.kernels import NSF_RBF
    from gpzoo.likelihoods import NSF2, Hybrid_NSF
    from gpzoo.utilities import train_batched, train_hybrid_batched
    z = np.load(os.path.join(GOLDEN, f"ref_trajectory_{name}_f64.npz"))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    pre = "prior." if name == "nsf2" else "gp."
    L, M = z["init." + pre + "mu"].shape
    y = t("y")
    gp = SVGP(NSF_RBF(L=L), dim=2, M=M, jitter=float(z["jitter"]))
    gp.mu = nn.Parameter(torch.zeros(L, M)); gp.Lu = nn.Parameter(torch.zeros(L, M, M))
    if name == "nsf2":
        model, loop = NSF2(gp, y, L=L), train_batched
    else:
        model, loop = Hybrid_NSF(gp, y, L=L, non_spatial_factors=z["eps2"].shape[2]), train_hybrid_batched
    model = model.double()
    model.load_state_dict({k[5:]: t(k) for k in z.files if k.startswith("init.")}, strict=True)
    model = model.cuda()
    e1, e2 = t("eps1").cuda(), t("eps2").cuda()
    queue = [e for e in e1] if name == "nsf2" else [e for pair in zip(e1, e2) for e in pair]
    iq = [i.cuda() for i in t("idx")]
    o_norm, o_multi = tdn._standard_normal, torch.multinomial
    tdn._standard_normal = lambda shape, dtype, device: queue.pop(0).to(dtype)
    torch.multinomial = lambda *a, **k: iq.pop(0)
    try:
        opt = torch.optim.Adam(model.parameters(), lr=float(z["lr"]))
        losses = loop(model, opt, t("X").cuda(), y.cuda(), torch.device("cuda"), steps=e1.shape[0], E=e1.shape[1],
                      batch_size=z["idx"].shape[1], fused=fused)
    finally:
        tdn._standard_normal, torch.multinomial = o_norm, o_multi
    torch.testing.assert_close(torch.tensor(losses, dtype=torch.float64), t("losses"), rtol=1e-7, atol=0)
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v.cpu(), t("final." + k), rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("seed", range(24))
def test_random_poisson_shapes(seed):
    """Seeded sweep over ragged (genes, spots, factors, samples): value and the four gradients of the fused
    expected log-likelihood against the plain torch evaluation, with and without the log y! term."""
    from gpzoo_amd import ops
    g = torch.Generator().manual_seed(900 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))   # noqa: E731
    D, N = [1, 7, 130, 257, 1000][ri(0, 4)], [1, 63, 64, 500, 3001][ri(0, 4)]
    Lt, E = [1, 3, 8, 9, 20, 33, 40, 64][ri(0, 7)], [1, 2, 3, 8][ri(0, 3)]      # (8 samples: two LDS groups of pass B)      # 40 = the notebooks' L=20 + T=20 hybrids
    with_lgamma = bool(seed % 2)
    mean = 0.3 * torch.randn(Lt, N, generator=g)
    scale = 0.2 + 0.3 * torch.rand(Lt, N, generator=g)
    eps = torch.randn(E, Lt, N, generator=g)
    W = torch.rand(D, Lt, generator=g) + 0.05
    V = 0.5 + torch.rand(N, generator=g)
    y = torch.poisson(2.0 * torch.rand(D, N, generator=g), generator=g)
    ll, dmean, dscale, dW, dV = ops.poisson_nsf(mean.cuda(), scale.cuda(), eps.cuda(), W.cuda(), V.cuda(), y.cuda(), with_lgamma)
    lv = [t.double().requires_grad_(True) for t in (mean, scale, W, V)]
    rate = lv[3] * torch.matmul(lv[2], torch.exp(lv[0] + lv[1] * eps.double()))
    yd = y.double()
    ref = (torch.distributions.Poisson(rate).log_prob(yd) if with_lgamma else yd * torch.log(rate) - rate).mean(0).sum()
    ref.backward()
    tag = dict(D=D, N=N, Lt=Lt, E=E, with_lgamma=with_lgamma)
    assert float(ll.detach()) == pytest.approx(float(ref), rel=5e-5, abs=1e-3), tag
    for got, want, nm in zip((dmean, dscale, dW, dV), lv, ("dmean", "dscale", "dW", "dV")):
        sc = float(want.grad.abs().max()) + 1e-30
        torch.testing.assert_close(got.double().cpu(), want.grad, rtol=1e-3, atol=1e-3 * sc, msg=lambda m: f"{nm} {tag}: {m}")


@pytest.mark.parametrize("Lt", [40, 64])
def test_more_samples_after_fewer_with_large_dynamic_lds(Lt):
    """Above 64 KB the gene pass's dynamic LDS is an opt-in that is set once per kernel and device; it used to be set
    to the FIRST call's size, so a 3-sample evaluation followed by a 4-sample step failed with hipErrorInvalidValue
    (ADVICE r3).  Ascending sample counts in one process, each against the torch evaluation."""
    from gpzoo_amd import ops
    g = torch.Generator().manual_seed(77 + Lt)
    D, N = 130, 500
    mean = 0.3 * torch.randn(Lt, N, generator=g)
    scale = 0.2 + 0.3 * torch.rand(Lt, N, generator=g)
    W = torch.rand(D, Lt, generator=g) + 0.05
    V = 0.5 + torch.rand(N, generator=g)
    y = torch.poisson(2.0 * torch.rand(D, N, generator=g), generator=g)
    for E in (2, 3, 4):
        eps = torch.randn(E, Lt, N, generator=g)
        ll, dmean, dscale, dW, dV = ops.poisson_nsf(mean.cuda(), scale.cuda(), eps.cuda(), W.cuda(), V.cuda(), y.cuda(), False)
        lw = W.double().requires_grad_(True)
        rate = V.double() * torch.matmul(lw, torch.exp(mean.double() + scale.double() * eps.double()))
        ref = (y.double() * torch.log(rate) - rate).mean(0).sum()
        ref.backward()
        assert float(ll) == pytest.approx(float(ref), rel=5e-5), (Lt, E)
        torch.testing.assert_close(dW.double().cpu(), lw.grad, rtol=1e-3, atol=1e-3 * float(lw.grad.abs().max()))


@pytest.mark.parametrize("E", [5, 20, 33])
def test_many_samples_in_one_call(E):
    """E = 20 is what the reference's benchmark notebooks run (NSF_benchmarks.ipynb:330, 392): one call of the kernel --
    pass B walks the samples in LDS groups of four with y held in registers, E = 5 leaves a partial last group, E = 33
    is split by the host into 32 + 1 -- against the torch evaluation."""
    from gpzoo_amd import ops
    g = torch.Generator().manual_seed(500 + E)
    D, N, Lt = 70, 300, 20
    mean = 0.3 * torch.randn(Lt, N, generator=g)
    scale = 0.2 + 0.3 * torch.rand(Lt, N, generator=g)
    eps = torch.randn(E, Lt, N, generator=g)
    W = torch.rand(D, Lt, generator=g) + 0.05
    V = 0.5 + torch.rand(N, generator=g)
    y = torch.poisson(2.0 * torch.rand(D, N, generator=g), generator=g)
    ll, dmean, dscale, dW, dV = ops.poisson_nsf(mean.cuda(), scale.cuda(), eps.cuda(), W.cuda(), V.cuda(), y.cuda(), True)
    lv = [t.double().requires_grad_(True) for t in (mean, scale, W, V)]
    rate = lv[3] * torch.matmul(lv[2], torch.exp(lv[0] + lv[1] * eps.double()))
    ref = torch.distributions.Poisson(rate).log_prob(y.double()).mean(0).sum()
    ref.backward()
    assert float(ll) == pytest.approx(float(ref), rel=5e-5)
    for got, want, nm in zip((dmean, dscale, dW, dV), lv, ("dmean", "dscale", "dW", "dV")):
        torch.testing.assert_close(got.double().cpu(), want.grad, rtol=1e-3, atol=1e-3 * float(want.grad.abs().max()),
                                   msg=lambda m: f"{nm} E={E}: {m}")


def test_factor_count_limit_is_reported():
    from gpzoo_amd import ops
    Lt, N, D = 65, 10, 4
    with pytest.raises(RuntimeError, match="65 factors unsupported"):
        ops.poisson_nsf(torch.zeros(Lt, N).cuda(), torch.ones(Lt, N).cuda(), torch.zeros(1, Lt, N).cuda(),
                        torch.ones(D, Lt).cuda(), torch.ones(N).cuda(), torch.ones(D, N).cuda())


@pytest.mark.parametrize("whitened", [True, False])
def test_mggp_nsf_minibatch_groups_follow_the_sampled_spots(whitened):
    """train_batched on MGGP_NSF: the fused step and the reference's literal forward_batched form see the same
    sampled spots AND their group ids (groupsX[idx], reference likelihoods.py:344-361); a full-length group vector
    handed to the batch would silently pair spot i of the batch with group id i of the data set."""
    import gpzoo.gp as G
    import gpzoo.kernels as K
    from gpzoo.likelihoods import MGGP_NSF
    from gpzoo.utilities import train_batched
    gen = torch.Generator().manual_seed(5)
    N, D, L, M, n_groups = 600, 40, 3, 30, 3
    X = (torch.rand(N, 2, generator=gen) - 0.5) * 20
    gX = torch.randint(0, n_groups, (N,), generator=gen)
    y = torch.poisson(3.0 * torch.rand(D, N, generator=gen), generator=gen)

    def run(fused):
        torch.manual_seed(11)
        k = K.MGGP_NSF_RBF(sigma=1.0, lengthscale=4.0, group_diff_param=0.8, n_groups=n_groups, L=L)
        gp = (G.MGGP_WSVGP(k, dim=2, M=M, n_groups=n_groups, jitter=1e-2) if whitened else
              G.MGGP_SVGP(k, dim=2, M=M, jitter=1e-2, n_groups=n_groups))
        gp.Z = nn.Parameter(X[:M].clone())
        gp.groupsZ = nn.Parameter(gX[:M].clone(), requires_grad=False)
        gp.mu = nn.Parameter(0.1 * torch.randn(L, M))
        gp.Lu = nn.Parameter(0.05 * torch.randn(L, M, M) - 0.5 * torch.eye(M))
        model = MGGP_NSF(gp, y, L=L).cuda()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        torch.manual_seed(12)
        return train_batched(model, opt, X.cuda(), y.cuda(), steps=4, E=3, batch_size=128, fused=fused, groupsX=gX.cuda())

    a, b = run(True), run(False)
    assert all(abs(u - v) <= 2e-3 * abs(v) for u, v in zip(a, b)), (a, b)
