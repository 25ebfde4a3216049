"""GPU: loss.backward() through the gpzoo modules reproduces the reference's autograd gradients
w.r.t. mu and Lu (golden vectors), for every (GP class x kernel class) cell, fp64 and fp32."""
import pytest
import torch
import torch.nn as nn

from conftest import golden_cases
from helpers import load_case, rtol_for
from test_hip_api import build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [n for n in golden_cases() if n != "cfg1_f64"])
def test_backward_matches_reference_autograd(name):
    from gpzoo.utilities import whitened_KL_batched
    c = load_case(name)
    model = build(name, c)
    gp = model.gp
    for t in [gp.Z, *gp.kernel.parameters()]:
        t.requires_grad_(False)                      # frozen hyper-parameters (Slide-seq notebooks' mode)
    X, y = c["X"].cuda(), c["y"].cuda()
    kw = {"groupsX": c["gX"].cuda()} if "gX" in c else {}
    pY, qF, qU, pU = model(X=X, E=1, **kw)
    s = torch.nn.functional.softplus(model.noise)
    kl = whitened_KL_batched(qU.mean, qU.scale_tril).sum() if c["whitened"] else \
        torch.distributions.kl_divergence(qU, pU).sum()
    loss = -(pY.log_prob(y).sum() - (qF.scale ** 2).sum() / (2 * s ** 2) - kl)   # utilities.py:479-485
    loss.backward()
    dt = X.dtype
    rt = rtol_for(dt)
    gmu, gLu = gp.mu.grad.cpu(), gp.Lu.grad.cpu()
    sc_mu, sc_Lu = float(c["grad_mu"].abs().max()), float(c["grad_Lu"].abs().max())
    torch.testing.assert_close(gmu, c["grad_mu"], rtol=rt, atol=rt * sc_mu)
    torch.testing.assert_close(gLu, c["grad_Lu"], rtol=rt, atol=rt * sc_Lu)
    assert float(loss.detach()) == pytest.approx(-c["elbo"], rel=rt)


@pytest.mark.parametrize("edit", ["inplace", "rebind", "sigma"])
@pytest.mark.parametrize("name", ["svgp_nsf_rbf_f64", "wsvgp_matern32_f64", "mggp_svgp_mggp_nsf_rbf_f64", "svgp_matern32_f32"])
def test_backward_of_an_earlier_forward_after_data_edit_and_second_forward(name, edit):
    """forward A -> `.data` edit of a frozen input (invisible to autograd's version counters) -> forward B (refactors
    into the SHARED factor cache) -> backward(A): the gradients must be those of problem A -- the reference's autograd
    goldens -- not a mix of B's factor with A's saved tensors (the backward used to trust the
    cache on the host key alone).  The backward holds private copies of A's inputs and trusts the buffer only while its
    generation counter is the one forward A committed."""
    from gpzoo.utilities import whitened_KL_batched
    c = load_case(name)
    model = build(name, c)
    gp = model.gp
    for t in [gp.Z, *gp.kernel.parameters()]:
        t.requires_grad_(False)                      # frozen: the persistent cross-call cache is in play
    X, y = c["X"].cuda(), c["y"].cuda()
    kw = {"groupsX": c["gX"].cuda()} if "gX" in c else {}

    def loss_of():
        pY, qF, qU, pU = model(X=X, E=1, **kw)
        s = torch.nn.functional.softplus(model.noise)
        kl = whitened_KL_batched(qU.mean, qU.scale_tril).sum() if c["whitened"] else \
            torch.distributions.kl_divergence(qU, pU).sum()
        return -(pY.log_prob(y).sum() - (qF.scale ** 2).sum() / (2 * s ** 2) - kl)

    loss_a = loss_of()
    gen_a = gp._factor_cache.generation
    if edit == "inplace":
        v0 = gp.Z._version
        gp.Z.data.add_(0.37)
        assert gp.Z._version == v0
    elif edit == "rebind":
        gp.Z.data = gp.Z.data * 1.1 + 0.2
    else:
        gp.kernel.sigma.data.mul_(1.3)
    loss_b = loss_of()                               # another forward, other inputs, same cache buffer
    assert gp._factor_cache.generation != gen_a      # it refactored
    assert float(loss_b.detach()) != float(loss_a.detach())
    loss_a.backward()
    rt = rtol_for(X.dtype)
    sc_mu, sc_Lu = float(c["grad_mu"].abs().max()), float(c["grad_Lu"].abs().max())
    torch.testing.assert_close(gp.mu.grad.cpu(), c["grad_mu"], rtol=rt, atol=rt * sc_mu)
    torch.testing.assert_close(gp.Lu.grad.cpu(), c["grad_Lu"], rtol=rt, atol=rt * sc_Lu)
    # and B's own backward still differentiates B: same numbers as a cache-less evaluation of the edited model
    gp.mu.grad = gp.Lu.grad = None
    loss_b.backward()
    g_mu, g_Lu = gp.mu.grad.clone(), gp.Lu.grad.clone()
    gp.mu.grad = gp.Lu.grad = None
    gp.cache_factor = False
    loss_of().backward()
    torch.testing.assert_close(g_mu, gp.mu.grad, rtol=1e-9 if X.dtype == torch.float64 else 1e-4, atol=1e-9 * max(sc_mu, 1.0)
                               if X.dtype == torch.float64 else 1e-4 * max(sc_mu, 1.0))
    torch.testing.assert_close(g_Lu, gp.Lu.grad, rtol=1e-9 if X.dtype == torch.float64 else 1e-4, atol=1e-9 * max(sc_Lu, 1.0)
                               if X.dtype == torch.float64 else 1e-4 * max(sc_Lu, 1.0))


@pytest.mark.parametrize("order", ["a_then_b", "b_then_a"])
@pytest.mark.parametrize("name", ["svgp_nsf_rbf_f64", "wsvgp_matern32_f64", "svgp_matern32_f32"])
def test_two_forwards_with_other_variational_parameters_before_either_backward(name, order):
    """forward A (mu, Lu) -> forward B (other mu, Lu; SAME frozen Z / kernel, so the shared cache keeps its factor and B
    overwrites the q(U) operands kept behind it) -> the two backward passes in either order: each must differentiate
    ITS problem.  The backward takes LuE^T / muE / LuE from the buffer only while no other call has written them since
    its own forward (FactorCache.qu_generation); otherwise it prepares them again from its saved (mu, Lu)."""
    from gpzoo.utilities import whitened_KL_batched
    c = load_case(name)
    model = build(name, c)
    gp = model.gp
    for t in [gp.Z, *gp.kernel.parameters()]:
        t.requires_grad_(False)
    X, y = c["X"].cuda(), c["y"].cuda()
    kw = {"groupsX": c["gX"].cuda()} if "gX" in c else {}

    def loss_of():
        pY, qF, qU, pU = model(X=X, E=1, **kw)
        s = torch.nn.functional.softplus(model.noise)
        kl = whitened_KL_batched(qU.mean, qU.scale_tril).sum() if c["whitened"] else \
            torch.distributions.kl_divergence(qU, pU).sum()
        return -(pY.log_prob(y).sum() - (qF.scale ** 2).sum() / (2 * s ** 2) - kl)

    mu_a, Lu_a = gp.mu, gp.Lu
    g = torch.Generator().manual_seed(5)
    mu_b = nn.Parameter((mu_a.detach().cpu() + 0.3 * torch.randn(mu_a.shape, generator=g, dtype=mu_a.dtype)).cuda())
    Lu_b = nn.Parameter((Lu_a.detach().cpu() * 0.7 + 0.05 * torch.randn(Lu_a.shape, generator=g, dtype=Lu_a.dtype)).cuda())
    loss_a = loss_of()
    gen = gp._factor_cache.generation
    gp.mu, gp.Lu = mu_b, Lu_b
    loss_b = loss_of()
    assert gp._factor_cache.generation == gen        # same factor: only the q(U) operands were replaced
    for which in (("a", "b") if order == "a_then_b" else ("b", "a")):
        (loss_a if which == "a" else loss_b).backward()
    rt = rtol_for(X.dtype)
    sc_mu, sc_Lu = float(c["grad_mu"].abs().max()), float(c["grad_Lu"].abs().max())
    torch.testing.assert_close(mu_a.grad.cpu(), c["grad_mu"], rtol=rt, atol=rt * sc_mu)
    torch.testing.assert_close(Lu_a.grad.cpu(), c["grad_Lu"], rtol=rt, atol=rt * sc_Lu)
    # B against a cache-less evaluation of B alone
    g_mu, g_Lu = mu_b.grad.clone(), Lu_b.grad.clone()
    mu_b.grad = Lu_b.grad = None
    gp.cache_factor = False
    loss_of().backward()
    tol = 1e-9 if X.dtype == torch.float64 else 1e-4
    torch.testing.assert_close(g_mu, mu_b.grad, rtol=tol, atol=tol * max(float(g_mu.abs().max()), 1.0))
    torch.testing.assert_close(g_Lu, Lu_b.grad, rtol=tol, atol=tol * max(float(g_Lu.abs().max()), 1.0))


def test_backward_multi_chunk_matches_single_chunk():
    """Chunked accumulation of the (M x n)(n x M) gradient product: 3 chunks == 1 chunk."""
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    c = make_config(2, N=5000, M=300, L=3, dtype=torch.float64)
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g)
    out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True)
    gm = torch.randn_like(out["mean"])
    gs = torch.randn_like(out["scale"])
    a = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, gm, gs, out["scale"])
    b = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, gm, gs, out["scale"], chunk=2048)
    torch.testing.assert_close(a[0], b[0], rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(a[1], b[1], rtol=1e-10, atol=1e-10)
    # and against torch autograd through the oracle's formulas on the CPU
    from oracle import svgp_oracle as O
    mu = c["mu"].clone().requires_grad_(True)
    Lur = c["Lu_raw"].clone().requires_grad_(True)
    Kxx = O.kernel_diag(c["sigma"], c["X"].shape[0])
    Kzx = O.kernel_matrix(c["kind"], c["Z"], c["X"], c["sigma"], c["lengthscale"])
    Kzz = O.add_jitter_(O.kernel_matrix(c["kind"], c["Z"], c["Z"], c["sigma"], c["lengthscale"]).contiguous(), c["jitter"])
    mean, scale, _, _ = O.wsvgp_moments(Kxx, Kzx, Kzz, mu, Lur)
    ((mean * gm.cpu()).sum() + (scale * gs.cpu()).sum()).backward()
    torch.testing.assert_close(a[0].cpu(), mu.grad, rtol=1e-8, atol=1e-8)
    torch.testing.assert_close(a[1].cpu(), Lur.grad, rtol=1e-8, atol=1e-8)


def test_train_loop_decreases_loss():
    """A few Adam steps through gpzoo.utilities.train (reference signature) on the fused path."""
    import torch.nn as nn
    from gpzoo.gp import WSVGP
    from gpzoo.kernels import NSF_RBF
    from gpzoo.likelihoods import GaussianLikelihood
    from gpzoo.utilities import train, train_batched
    torch.manual_seed(0)
    N, M, L = 600, 40, 2
    X = (torch.rand(N, 2) * 20 - 10)
    y = torch.stack([torch.sin(X[:, 0] / 3), torch.cos(X[:, 1] / 4)]) + 0.1 * torch.randn(L, N)
    gp = WSVGP(NSF_RBF(sigma=1.0, lengthscale=3.0, L=L), dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(X[:M].clone(), requires_grad=False)
    gp.mu = nn.Parameter(torch.zeros(L, M))
    gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    for t in gp.kernel.parameters():
        t.requires_grad_(False)
    model = GaussianLikelihood(gp, noise=0.5).cuda()
    opt = torch.optim.Adam([gp.mu, gp.Lu], lr=5e-2)
    losses = train(model, opt, X.cuda(), y.cuda(), torch.device("cuda"), steps=40, E=4)
    assert losses[-1] < 0.7 * losses[0]
    losses_b = train_batched(model, opt, X.cuda(), y.cuda(), torch.device("cuda"), steps=5, E=2, batch_size=200)
    assert all(map(lambda v: v == v, losses_b))


def _safe_matern(A, B, sigma, ell):
    """Matern-3/2 with a sqrt whose gradient at r = 0 is 0 (the reference's is NaN there)."""
    d2 = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1)
    pos = d2 > 0
    r = torch.where(pos, d2, torch.ones_like(d2)).sqrt() * pos
    v = (3 ** 0.5) * r / ell.reshape(-1, 1, 1)
    return sigma.reshape(-1, 1, 1) ** 2 * (1 + v) * torch.exp(-v)


@pytest.mark.parametrize("cfg,kind", [(2, "nsf_rbf"), (3, "matern32"), (5, "mggp_nsf_rbf")])
def test_kernel_and_Z_gradients_match_autograd(cfg, kind):
    """dLoss/d(sigma, lengthscale, a, Z, mu, Lu) of the whitened path against torch autograd through the
    oracle's formulas (fp64, CPU), for an arbitrary upstream (g_mean, g_scale)."""
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    from oracle import svgp_oracle as O
    c = make_config(cfg, N=700, M=150, L=3, dtype=torch.float64)
    c["sigma"] = torch.tensor([0.8, 1.0, 1.3], dtype=torch.float64)
    if cfg == 5:
        c["lengthscale"] = torch.tensor([6.0, 8.0, 11.0], dtype=torch.float64)
        c["group_diff"] = torch.tensor([0.7, -0.4, 1.1], dtype=torch.float64)
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g)
    out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, **extra)
    gen = torch.Generator().manual_seed(11)
    gm = torch.randn(out["mean"].shape, generator=gen, dtype=torch.float64)
    gs = torch.randn(out["scale"].shape, generator=gen, dtype=torch.float64)
    gmu, gLu, gth, gZ = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, gm.cuda(),
                                          gs.cuda(), out["scale"], kernel_grads=True, **extra)
    # autograd reference
    leaf = {k: c[k].clone().requires_grad_(True) for k in ("Z", "sigma", "lengthscale", "mu", "Lu_raw")}
    kw = {}
    if cfg == 5:
        G = c["n_groups"]
        leaf["group_diff"] = c["group_diff"].clone().requires_grad_(True)
        kw = dict(embedding=O.embed_group_distances(torch.ones(G, G) - torch.eye(G)).double(), group_diff=leaf["group_diff"])
    def K(A, B, gA=None, gB=None):
        if kind == "matern32":
            return _safe_matern(A, B, leaf["sigma"], leaf["lengthscale"])
        return O.kernel_matrix(kind, A, B, leaf["sigma"], leaf["lengthscale"], gA=gA, gB=gB, **kw)
    Kzx = K(leaf["Z"], c["X"], c.get("gZ"), c.get("gX"))
    Kzz = K(leaf["Z"], leaf["Z"], c.get("gZ"), c.get("gZ")) + c["jitter"] * torch.eye(150, dtype=torch.float64)
    Kxx = (leaf["sigma"] ** 2)[:, None].expand(-1, 700)
    mean, scale, _, _ = O.wsvgp_moments(Kxx, Kzx, Kzz, leaf["mu"], leaf["Lu_raw"])
    ((mean * gm).sum() + (scale * gs).sum()).backward()
    tol = dict(rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(gmu.cpu(), leaf["mu"].grad, **tol)
    torch.testing.assert_close(gLu.cpu(), leaf["Lu_raw"].grad, **tol)
    torch.testing.assert_close(gth[:, 0].cpu(), leaf["sigma"].grad, **tol)
    torch.testing.assert_close(gth[:, 1].cpu(), leaf["lengthscale"].grad, **tol)
    torch.testing.assert_close(gZ.cpu(), leaf["Z"].grad, **tol)
    if cfg == 5:   # effective multiplier a^2 (MGGP_NSF_RBF): chain rule 2a
        torch.testing.assert_close((gth[:, 2].cpu() * 2 * c["group_diff"]), leaf["group_diff"].grad, **tol)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_all_parameter_gradients_with_columns_at_the_whitened_clamp(dtype):
    """gp.py:287 clamp(Kxx - sum W^2, min=0): no gradient flows through a clamped prior term.  A negative jitter (Kzz
    stays positive-definite: short lengthscales on scattered points) puts sigma^2 - colsum(Wt^2) below zero at the data
    points that coincide with inducing points, so those columns carry the weights gv2 (1 - c) of the correction terms
    (Hd in dLoss/dL, [Linv^T W] diag(gv2 (1 - c)) in Kbar_x), which run only in chunks that hold such a column (every
    other whitened test takes the gated-off side).  Two chunks; every gradient against torch autograd over the oracle."""
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    from oracle import svgp_oracle as O
    N, M, L = 3000, 150, 3
    c = make_config(2, N=N, M=M, L=L, dtype=torch.float64)
    c["sigma"] = torch.tensor([0.8, 1.0, 1.3], dtype=torch.float64)
    c["lengthscale"] = torch.tensor([3.0, 3.5, 4.0], dtype=torch.float64)
    c["Z"] = c["X"][300:300 + M].clone()                          # inducing points ARE data points (as the configuration draws them)
    jitter = -0.02                                                # (the smallest eigenvalue of Kzz is 0.046 here)
    g = {k: (v.to(dtype).cuda() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in c.items()}
    spec, extra = spec_for_config(g)
    chunk = 2048
    out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], jitter, True, chunk=chunk, **extra)
    gen = torch.Generator().manual_seed(12)
    gm = torch.randn(L, N, generator=gen, dtype=torch.float64)
    gs = torch.randn(L, N, generator=gen, dtype=torch.float64)
    gmu, gLu, gth, gZ = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], jitter, True, gm.to(dtype).cuda(),
                                          gs.to(dtype).cuda(), out["scale"], kernel_grads=True, chunk=chunk, **extra)
    leaf = {k: c[k].clone().requires_grad_(True) for k in ("Z", "sigma", "lengthscale", "mu", "Lu_raw")}
    Kzx = O.kernel_matrix("nsf_rbf", leaf["Z"], c["X"], leaf["sigma"], leaf["lengthscale"])
    Kzz = O.kernel_matrix("nsf_rbf", leaf["Z"], leaf["Z"], leaf["sigma"], leaf["lengthscale"]) + jitter * torch.eye(M, dtype=torch.float64)
    Kxx = (leaf["sigma"] ** 2)[:, None].expand(-1, N)
    with torch.no_grad():
        Wt = torch.linalg.solve_triangular(torch.linalg.cholesky(Kzz), Kzx, upper=False)
        at_clamp = (Kxx - (Wt ** 2).sum(1)) <= 0
    assert int(at_clamp.sum()) >= L * M                           # every coincident column, in every latent
    mean, scale, _, _ = O.wsvgp_moments(Kxx, Kzx, Kzz, leaf["mu"], leaf["Lu_raw"])
    ((mean * gm).sum() + (scale * gs).sum()).backward()
    rt = 1e-6 if dtype == torch.float64 else 2e-3
    def close(got, ref, name):
        torch.testing.assert_close(got.double().cpu(), ref, rtol=rt, atol=rt * float(ref.abs().max()), msg=lambda m: f"{name}: {m}")
    close(gmu, leaf["mu"].grad, "mu"); close(gLu, leaf["Lu_raw"].grad, "Lu")
    close(gth[:, 0], leaf["sigma"].grad, "sigma"); close(gth[:, 1], leaf["lengthscale"].grad, "lengthscale")
    close(gZ, leaf["Z"].grad, "Z")


@pytest.mark.parametrize("name", [n for n in golden_cases() if n != "cfg1_f64"])
def test_hyperparameter_gradients_match_reference(name):
    """Everything trainable at once (mu, Lu, Z, sigma, lengthscale, group_diff_param): loss.backward()
    through all four GP classes against the reference's own autograd gradients."""
    from gpzoo.utilities import whitened_KL_batched
    c = load_case(name)
    model = build(name, c)
    gp = model.gp
    X, y = c["X"].cuda(), c["y"].cuda()
    kw = {"groupsX": c["gX"].cuda()} if "gX" in c else {}
    pY, qF, qU, pU = model(X=X, E=1, **kw)
    s = torch.nn.functional.softplus(model.noise)
    kl = whitened_KL_batched(qU.mean, qU.scale_tril).sum() if c["whitened"] else \
        torch.distributions.kl_divergence(qU, pU).sum()
    loss = -(pY.log_prob(y).sum() - (qF.scale ** 2).sum() / (2 * s ** 2) - kl)
    loss.backward()
    rt = rtol_for(X.dtype)
    if not c["whitened"] and X.dtype == torch.float32:
        rt = 5e-3   # the reference's own fp32 un-whitened gradients carry W @ (S - Kzz) cancellation (SURVEY: 6e-5 on the ELBO)
    def close(got, ref):
        sc = float(ref.abs().max()) + 1e-30
        torch.testing.assert_close(got.cpu(), ref, rtol=rt, atol=rt * sc)
    close(gp.mu.grad, c["grad_mu"])
    close(gp.Lu.grad, c["grad_Lu"])
    close(gp.kernel.sigma.grad, c["grad_sigma"])
    close(gp.kernel.lengthscale.grad, c["grad_lengthscale"])
    if not torch.isnan(c["grad_Z"]).any():       # the reference's Matern Z-gradient is NaN (sqrt at r = 0)
        close(gp.Z.grad, c["grad_Z"])
    else:
        assert torch.isfinite(gp.Z.grad).all()
    if "grad_group_diff" in c:
        close(gp.kernel.group_diff_param.grad, c["grad_group_diff"])


def test_training_trajectory_matches_reference():
    """End to end: the reference's `utilities.train` ran 12 Adam steps on SVGP + NSF_RBF +
    GaussianLikelihood (all parameters trainable, rsample noise stored in the fixture); the same
    loop through the HIP forward/backward reproduces every loss and the final parameters."""
    import os
    import numpy as np
    import torch.distributions.normal as tdn
    import torch.nn as nn
    from helpers import GOLDEN
    from gpzoo.gp import SVGP
    from gpzoo.kernels import NSF_RBF
    from gpzoo.likelihoods import GaussianLikelihood
    from gpzoo.utilities import train
    z = np.load(os.path.join(GOLDEN, "ref_trajectory_svgp_f64.npz"))
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    L, M = z["init.gp.mu"].shape
    gp = SVGP(NSF_RBF(L=L), dim=2, M=M, jitter=float(z["jitter"]))
    gp.mu = nn.Parameter(torch.zeros(L, M)); gp.Lu = nn.Parameter(torch.zeros(L, M, M))
    model = GaussianLikelihood(gp, noise=float(z["noise0"])).double()
    model.load_state_dict({k[5:]: t(k) for k in z.files if k.startswith("init.")}, strict=True)
    model = model.cuda()
    eps = t("eps").cuda()
    queue = [e for e in eps]
    orig = tdn._standard_normal
    tdn._standard_normal = lambda shape, dtype, device: queue.pop(0).to(dtype)
    try:
        opt = torch.optim.Adam(model.parameters(), lr=float(z["lr"]))
        losses = train(model, opt, t("X").cuda(), t("y").cuda(), torch.device("cuda"), steps=eps.shape[0], E=eps.shape[1])
    finally:
        tdn._standard_normal = orig
    torch.testing.assert_close(torch.tensor(losses, dtype=torch.float64), t("losses"), rtol=1e-7, atol=0)
    for k, v in model.state_dict().items():
        torch.testing.assert_close(v.cpu(), t("final." + k), rtol=1e-5, atol=1e-7, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("whitened,kernel_grads", [(True, False), (True, True), (False, True)])
def test_retained_wt_gives_identical_gradients(whitened, kernel_grads):
    """Training keeps Wt of every chunk in HBM between forward and backward (gpz_svgp_problem.wt_cache):
    the gradients are bitwise those of the recomputing path, over several ragged chunks."""
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    c = make_config(2, N=5000, M=300, L=3)
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g)
    args = (spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], whitened)
    out = ops.svgp_forward(*args, chunk=2048, retain_wt=0.5, want_chol=not whitened, **extra)
    ref = ops.svgp_forward(*args, chunk=2048, want_chol=not whitened, **extra)
    assert "wt_cache" in out and torch.equal(out["mean"], ref["mean"]) and torch.equal(out["scale"], ref["scale"])
    gen = torch.Generator().manual_seed(5)
    gm = torch.randn(out["mean"].shape, generator=gen).cuda()
    gs = torch.randn(out["scale"].shape, generator=gen).cuda()
    a = ops.svgp_backward(*args, gm, gs, out["scale"], chunk=2048, kernel_grads=kernel_grads, wt_cache=out["wt_cache"], **extra)
    b = ops.svgp_backward(*args, gm, gs, out["scale"], chunk=2048, kernel_grads=kernel_grads, **extra)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    with pytest.raises(ValueError):
        ops.svgp_backward(*args, gm, gs, out["scale"], chunk=1024, wt_cache=out["wt_cache"], **extra)


@pytest.mark.parametrize("name", ["wsvgp_nsf_rbf_f64", "svgp_nsf_rbf_f64", "mggp_svgp_mggp_nsf_rbf_f64", "svgp_rbf_f64"])
def test_fused_kl_equals_torch_kl_in_value_and_gradients(name):
    """The KL carried by the returned q(U) (whitened: read by the training loops; un-whitened: what
    kl_divergence(qU, pU) resolves to) against torch's own formulas on the same distributions, value and
    gradients w.r.t. every parameter; and a q(U) paired with a p(U) of another call falls back to torch."""
    from torch import distributions
    from gpzoo.utilities import whitened_KL_batched
    c = load_case(name)
    X = c["X"].cuda()
    kw = {"groupsX": c["gX"].cuda()} if "gX" in c else {}
    grads, vals = [], []
    for fused in (True, False):
        model = build(name, c)
        gp = model.gp
        pY, qF, qU, pU = model(X=X, E=1, **kw)
        if c["whitened"]:
            kl = qU._gpz_kl if fused else whitened_KL_batched(qU.mean, qU.scale_tril)
        else:
            kl = distributions.kl_divergence(qU, pU) if fused else \
                distributions.kl._kl_multivariatenormal_multivariatenormal(qU, pU)
        w = torch.linspace(0.5, 1.5, kl.numel(), dtype=kl.dtype, device=kl.device).reshape(kl.shape)
        ((w * kl).sum() + 0.3 * qF.mean.sum() + 0.1 * qF.scale.sum()).backward()
        vals.append(kl.detach().cpu())
        grads.append({n: p.grad.detach().cpu().clone() for n, p in gp.named_parameters() if p.grad is not None})
    torch.testing.assert_close(vals[0], vals[1], rtol=1e-9, atol=1e-12)
    assert set(grads[0]) == set(grads[1]) and len(grads[0]) >= 5
    for n in grads[0]:
        ref = grads[1][n]
        torch.testing.assert_close(grads[0][n], ref, rtol=1e-7, atol=1e-9 * float(ref.abs().max() + 1e-30), msg=lambda m: f"{n}: {m}")
    if not c["whitened"]:
        m1, m2 = build(name, c), build(name, c)
        _, _, qU1, _ = m1(X=X, E=1, **kw)
        _, _, _, pU2 = m2(X=X, E=1, **kw)
        mixed = distributions.kl_divergence(qU1, pU2)
        torch.testing.assert_close(mixed.detach().cpu(), vals[1], rtol=1e-9, atol=1e-12)
        assert mixed.grad_fn is not qU1._gpz_kl.grad_fn


@pytest.mark.parametrize("whitened", [True, False])
def test_training_steps_do_not_accumulate_device_memory(whitened):
    """Nothing a step allocates (workspaces, the retained Wt, distributions carrying the fused KL) may outlive
    it: allocated device memory is flat over repeated optimisation steps without waiting for the cyclic GC."""
    import gc
    from gpzoo.gp import SVGP, WSVGP
    from gpzoo.kernels import NSF_RBF
    from gpzoo.likelihoods import GaussianLikelihood
    from gpzoo.utilities import train_batched
    torch.manual_seed(0)
    N, M, L = 4000, 200, 3
    X = (torch.rand(N, 2) * 50).cuda(); y = torch.randn(L, N).cuda()
    gp = (WSVGP if whitened else SVGP)(NSF_RBF(sigma=1.0, lengthscale=4.0, L=L), dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(X[:M].clone().cpu()); gp.mu = nn.Parameter(torch.zeros(L, M)); gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    model = GaussianLikelihood(gp, noise=0.5).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    gc.collect(); gc.disable()
    try:
        mem = []
        for _ in range(4):
            train_batched(model, opt, X, y, torch.device("cuda"), steps=8, E=2, batch_size=1000)
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert mem[-1] == mem[1], mem


def test_only_hyperparameters_trainable():
    """mu and Lu frozen, lengthscale / sigma / Z trainable (the length-scale estimation notebooks' mode): the
    gradients still flow, and equal those of the all-trainable run."""
    c = load_case("wsvgp_nsf_rbf_f64")
    X = c["X"].cuda()
    grads = []
    for freeze in (False, True):
        model = build("wsvgp_nsf_rbf_f64", c)
        gp = model.gp
        if freeze:
            gp.mu.requires_grad_(False); gp.Lu.requires_grad_(False)
        pY, qF, qU, pU = model(X=X, E=1)
        ((qF.mean * c["y"].cuda()).sum() + (qF.scale ** 2).sum()).backward()
        assert (gp.mu.grad is None) == freeze
        grads.append([gp.kernel.lengthscale.grad.clone(), gp.kernel.sigma.grad.clone(), gp.Z.grad.clone()])
    for a, b in zip(*grads):
        assert torch.equal(a, b) and float(a.abs().max()) > 0


@pytest.mark.parametrize("name,clamp", [("wsvgp_nsf_rbf_f64", False), ("wsvgp_nsf_rbf_f64", True), ("wsvgp_matern32_f32", False),
                                        ("wsvgp_rbf_f64", False)])
def test_forward_precomputed_is_differentiable(name, clamp):
    """WSVGP.forward_precomputed under loss.backward(): gradients to mu, Lu and sigma equal torch autograd
    through the reference's own expression (gp.py:308-322) evaluated on the same W -- also where the
    clamp(sigma^2 - sum W^2, min=0) is active (no gradient through a clamped term)."""
    c = load_case(name)
    model = build(name, c)
    gp = model.gp
    W = torch.linalg.solve_triangular(c["chol"], c["Kzx"], upper=False).transpose(-1, -2).contiguous()
    if clamp:
        W = W * 1.6                                  # pushes sigma^2 - sum W^2 below zero for part of the points
    W = W.cuda()
    R1, R2 = torch.randn_like(c["mean"]).cuda(), torch.rand_like(c["mean"]).cuda()
    qF, qU, pU = gp.forward_precomputed(W)
    assert qF.mean.requires_grad and pU is None
    loss = (qF.mean * R1).sum() + (qF.scale * R2).sum() + (qU.scale_tril ** 2).sum()
    loss.backward()
    got = {n: getattr(t, "grad").clone() for n, t in (("mu", gp.mu), ("Lu", gp.Lu), ("sigma", gp.kernel.sigma))}
    # the reference's expression, torch autograd
    mu = gp.mu.detach().clone().requires_grad_(True)
    Lur = gp.Lu.detach().clone().requires_grad_(True)
    sig = gp.kernel.sigma.detach().clone().requires_grad_(True)
    Lu = Lur.tril(-1) + torch.diag_embed(torch.diagonal(Lur, dim1=-2, dim2=-1).exp())
    s2 = (sig ** 2).reshape(-1, 1) if sig.dim() else sig ** 2
    cov = s2 - (W ** 2).sum(-1)
    if clamp:
        assert 0 < int((cov <= 0).sum()) < cov.numel()
    cov = cov.clamp(min=0.0) + ((W @ Lu) ** 2).sum(-1)
    mean = (W @ mu.unsqueeze(-1)).squeeze(-1)
    ref = (mean * R1).sum() + (cov ** 0.5 * R2).sum() + (Lu ** 2).sum()
    ref.backward()
    rt = rtol_for(W.dtype)
    assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=rt)
    for n, t in (("mu", mu), ("Lu", Lur), ("sigma", sig)):
        torch.testing.assert_close(got[n], t.grad, rtol=rt, atol=rt * float(t.grad.abs().max()), msg=lambda m: f"{n}: {m}")


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_config2_training_step_gradients_against_the_reference_itself(dt):
    """BASELINE configs[1] as stated (N=50 000, M=512, L=8, NSF_RBF): gradients of -ELBO w.r.t. EVERY parameter (mu, Lu, Z,
    sigma, lengthscale, the noise) through the module API (HIP forward + backward) against the reference's own autograd run
    of the same step in fp64 (tests/golden/make_baseline_golden.py): the configuration's fp32 at 1e-3 of each gradient's
    scale, an fp64 run of the HIP path at 1e-5 (north_star tolerances)."""
    import math
    import os
    import numpy as np
    import torch.nn as nn
    import gpzoo.gp as G
    import gpzoo.kernels as K
    from gpzoo.likelihoods import ExactLikelihood
    from gpzoo.utilities import whitened_KL_batched
    from gpzoo_amd.synthetic import make_config
    from helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "baseline_cfg2_grads.npz"), allow_pickle=False)
    c = make_config(2)                                    # fp32-representable inputs, as the generator used
    L, M = c["mu"].shape
    k = K.NSF_RBF(L=L)
    k.sigma = nn.Parameter(c["sigma"].reshape(L, 1, 1).to(dt).clone())
    k.lengthscale = nn.Parameter(c["lengthscale"].reshape(L, 1, 1).to(dt).clone())
    gp = G.WSVGP(k, dim=2, M=M, jitter=c["jitter"])
    gp.Z, gp.mu, gp.Lu = (nn.Parameter(c[n].to(dt).clone()) for n in ("Z", "mu", "Lu_raw"))
    model = ExactLikelihood(gp, noise=math.log(math.expm1(0.5)))
    model = (model.double() if dt == torch.float64 else model.float()).cuda()
    X, y = c["X"].to(dt).cuda(), c["y"].to(dt).cuda()
    pY, qF, qU, pU = model(X=X, E=1)
    s = torch.nn.functional.softplus(model.noise)
    loss = -(pY.log_prob(y).sum() - (qF.scale ** 2).sum() / (2 * s ** 2) - whitened_KL_batched(qU.mean, qU.scale_tril).sum())
    loss.backward()
    rt = 1e-5 if dt == torch.float64 else 1e-3
    assert float(loss) == pytest.approx(float(z["f64_loss"]), rel=rt)
    gLu = gp.Lu.grad.double().cpu()
    got = {"grad_mu": gp.mu.grad, "grad_Z": gp.Z.grad, "grad_sigma": k.sigma.grad.reshape(-1),
           "grad_lengthscale": k.lengthscale.grad.reshape(-1), "grad_Lu_rowsum": gLu.sum(-1), "grad_Lu_colsum": gLu.sum(-2),
           "grad_Lu_diag": torch.diagonal(gLu, dim1=-2, dim2=-1)}
    for n, v in got.items():
        ref = torch.from_numpy(z["f64_" + n])
        scale = float(z["f64_grad_Lu_absmax"]) * (10 if "sum" in n else 1) if n.startswith("grad_Lu") else float(ref.abs().max())
        torch.testing.assert_close(v.double().cpu().reshape(ref.shape), ref, rtol=rt, atol=rt * scale, msg=lambda m: f"{n}: {m}")
    assert float(model.noise.grad) == pytest.approx(float(z["f64_grad_noise"]), rel=rt)
