"""GPU: stand-alone kernel calls are differentiable like the reference's traced modules -- values and the
gradients the reference's own autograd produces (tests/golden/kernel_grads.npz, exact_mggp_step_f64.npz,
written by make_golden.py from /root/reference) for sigma, lengthscale, group_diff_param, X and Z in every
kernel class, plus one training step of exact_mggp.ipynb's inline ExactGP."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

KINDS = ["rbf", "nsf_rbf", "batched_rbf_vec", "batched_rbf_scalar", "matern32_vec", "matern32_scalar", "mggp_rbf",
         "mggp_nsf_rbf", "batched_mggp_rbf"]


def _load():
    z = np.load(os.path.join(GOLDEN, "kernel_grads.npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _make(kind, dtype, gold, tag):
    import gpzoo.kernels as K

    def vec(v):
        return nn.Parameter(torch.tensor(v, dtype=dtype))

    def p3(v):
        return nn.Parameter(torch.tensor(v, dtype=dtype).reshape(-1, 1, 1))

    sig, ell, a = [1.0, 0.8, 1.3], [2.5, 4.0, 6.0], [0.7, 0.4, 1.1]
    mggp = False
    if kind == "rbf":
        k = K.RBF(sigma=1.2, lengthscale=3.0)
    elif kind == "nsf_rbf":
        k = K.NSF_RBF(L=3)
        k.sigma, k.lengthscale = p3(sig), p3(ell)
    elif kind == "batched_rbf_vec":
        k = K.batched_RBF()
        k.sigma, k.lengthscale = vec(sig), vec(ell)
    elif kind == "batched_rbf_scalar":
        k = K.batched_RBF(sigma=1.2, lengthscale=3.0)
    elif kind == "matern32_vec":
        k = K.batched_Matern32()
        k.sigma, k.lengthscale = vec(sig), vec(ell)
    elif kind == "matern32_scalar":
        k = K.batched_Matern32(sigma=0.9, lengthscale=2.0)
    elif kind == "mggp_rbf":
        k, mggp = K.MGGP_RBF(sigma=1.1, lengthscale=3.5, group_diff_param=0.6, n_groups=3), True
    elif kind == "mggp_nsf_rbf":
        k, mggp = K.MGGP_NSF_RBF(n_groups=3, L=3), True
        k.sigma, k.lengthscale, k.group_diff_param = p3(sig), p3(ell), p3(a)
    else:
        k, mggp = K.batched_MGGP_RBF(sigma=1.1, lengthscale=3.5, group_diff_param=-0.6, n_groups=3), True
    k = k.to(dtype).cuda()
    if mggp:
        emb = gold[f"{tag}.{kind}.embedding"].to(dtype).cuda()
        k.embedding = nn.Parameter(emb, requires_grad=False) if isinstance(k.embedding, nn.Parameter) else emb
    return k, mggp


def _close(got, ref, dtype, what):
    ref = ref.to(got.dtype)
    if dtype == torch.float64:
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-9 * max(1.0, float(ref.abs().max())), msg=lambda m: f"{what}: {m}")
    else:   # fp32: 1e-3 of the gradient's scale (sums of ~700 signed terms; the reference's fp32 distances carry error too)
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-3, atol=1e-3 * max(1e-3, float(ref.abs().max())), msg=lambda m: f"{what}: {m}")


@pytest.mark.parametrize("tag", ["f64", "f32"])
@pytest.mark.parametrize("kind", KINDS)
def test_kernel_matrix_gradients_match_reference_autograd(kind, tag):
    gold = _load()
    dtype = torch.float64 if tag == "f64" else torch.float32
    k, mggp = _make(kind, dtype, gold, tag)
    gX, gZ = gold["gX"].cuda(), gold["gZ"].cuda()
    X = gold[f"{tag}.X"].cuda().requires_grad_(True)
    Z = gold[f"{tag}.Z"].cuda().requires_grad_(True)
    pre = f"{tag}.{kind}."
    K = k(X, Z, gX, gZ) if mggp else k(X, Z)
    assert K.requires_grad and K.shape == gold[pre + "K"].shape
    _close(K.detach(), gold[pre + "K"], dtype, "K")
    R = gold[f"{tag}.R"].cuda()
    (K * (R if K.dim() == 3 else R[0])).sum().backward()
    _close(X.grad, gold[pre + "grad_X"], dtype, "grad_X")
    _close(Z.grad, gold[pre + "grad_Z"], dtype, "grad_Z")
    _close(k.sigma.grad, gold[pre + "grad_sigma"], dtype, "grad_sigma")
    _close(k.lengthscale.grad, gold[pre + "grad_lengthscale"], dtype, "grad_lengthscale")
    if mggp:
        _close(k.group_diff_param.grad, gold[pre + "grad_group_diff_param"], dtype, "grad_group_diff_param")
    if "matern" in kind:
        return
    # the same tensor in both slots: autograd adds the two roles' gradients
    k2, _ = _make(kind, dtype, gold, tag)
    X2 = gold[f"{tag}.X"].cuda().requires_grad_(True)
    K2 = k2(X2, X2, gX, gX) if mggp else k2(X2, X2)
    R2 = gold[f"{tag}.R2"].cuda()
    (K2 * (R2 if K2.dim() == 3 else R2[0])).sum().backward()
    _close(K2.detach(), gold[pre + "xx.K"], dtype, "Kxx")
    _close(X2.grad, gold[pre + "xx.grad_X"], dtype, "xx.grad_X")
    _close(k2.sigma.grad, gold[pre + "xx.grad_sigma"], dtype, "xx.grad_sigma")
    _close(k2.lengthscale.grad, gold[pre + "xx.grad_lengthscale"], dtype, "xx.grad_lengthscale")
    if mggp:
        _close(k2.group_diff_param.grad, gold[pre + "xx.grad_group_diff_param"], dtype, "xx.grad_group_diff_param")


def test_no_graph_when_nothing_requires_grad_and_diag_is_differentiable():
    import gpzoo.kernels as K
    k = K.NSF_RBF(L=3).cuda()
    X, Z = torch.randn(20, 2, device="cuda"), torch.randn(7, 2, device="cuda")
    with torch.no_grad():
        assert not k(X, Z).requires_grad
    for p in k.parameters():
        p.requires_grad_(False)
    assert not k(X, Z).requires_grad
    k.sigma.requires_grad_(True)
    d = k(X, X, diag=True)
    d.sum().backward()
    torch.testing.assert_close(k.sigma.grad, (2 * 20 * k.sigma.detach()))


def test_exact_mggp_notebook_step():
    """exact_mggp.ipynb: pY = MVN(0, kernel(X, X, g, g) + noise^2 I); loss = -log_prob(y); loss.backward()."""
    import gpzoo.kernels as K
    z = np.load(os.path.join(GOLDEN, "exact_mggp_step_f64.npz"), allow_pickle=False)
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    kernel = K.MGGP_RBF(sigma=1.3, lengthscale=1.7, group_diff_param=0.8, n_groups=2).double().cuda()
    kernel.embedding = g["embedding"].cuda()
    kernel.input_dim = 1
    noise = nn.Parameter(torch.tensor(0.4, dtype=torch.float64, device="cuda"))
    X, gX, y = g["X"].cuda(), g["gX"].cuda(), g["y"].cuda()
    N = len(X)
    Kxx = kernel.forward(X, X, gX, gX)
    torch.testing.assert_close(Kxx.detach().cpu(), g["Kxx"], rtol=1e-10, atol=1e-12)
    pY = torch.distributions.MultivariateNormal(torch.zeros(N, dtype=torch.float64, device="cuda"),
                                                Kxx + (noise ** 2) * torch.eye(N, dtype=torch.float64, device="cuda"))
    loss = -pY.log_prob(y).sum()
    loss.backward()
    assert float(loss) == pytest.approx(float(g["loss"]), rel=1e-9)
    for name, got in (("grad_sigma", kernel.sigma.grad), ("grad_lengthscale", kernel.lengthscale.grad),
                      ("grad_group_diff_param", kernel.group_diff_param.grad), ("grad_noise", noise.grad)):
        torch.testing.assert_close(got.cpu(), g[name], rtol=1e-6, atol=1e-9, msg=lambda m: f"{name}: {m}")


@pytest.mark.parametrize("d", [1, 3, 4])
@pytest.mark.parametrize("cls", ["NSF_RBF", "batched_Matern32"])
def test_kernel_gradients_in_other_input_dimensions(d, cls):
    """d = 1, 3, 4 inputs (the fixtures are 2-D): gradients of sum(K * R) against torch autograd through the closed
    form written out in torch (fp64)."""
    import gpzoo.kernels as K
    g = torch.Generator().manual_seed(40 + d)
    L, N, M = 3, 37, 11
    X = (torch.rand(N, d, generator=g, dtype=torch.float64) - 0.5) * 8
    Z = (torch.rand(M, d, generator=g, dtype=torch.float64) - 0.5) * 8
    R = torch.randn(L, N, M, generator=g, dtype=torch.float64)
    sig = torch.tensor([1.0, 0.8, 1.3], dtype=torch.float64)
    ell = torch.tensor([2.5, 4.0, 1.5], dtype=torch.float64)
    if cls == "NSF_RBF":
        k = K.NSF_RBF(L=L)
        k.sigma, k.lengthscale = nn.Parameter(sig.reshape(L, 1, 1).clone()), nn.Parameter(ell.reshape(L, 1, 1).clone())
    else:
        k = K.batched_Matern32()
        k.sigma, k.lengthscale = nn.Parameter(sig.clone()), nn.Parameter(ell.clone())
    k = k.double().cuda()
    Xg, Zg = X.cuda().requires_grad_(True), Z.cuda().requires_grad_(True)
    (k(Xg, Zg) * R.cuda()).sum().backward()
    Xr, Zr = X.clone().requires_grad_(True), Z.clone().requires_grad_(True)
    sr, lr = sig.clone().requires_grad_(True), ell.clone().requires_grad_(True)
    d2 = ((Xr[:, None, :] - Zr[None, :, :]) ** 2).sum(-1)
    if cls == "NSF_RBF":
        Kr = sr[:, None, None] ** 2 * torch.exp(-0.5 * d2 / lr[:, None, None] ** 2)
    else:
        v = (3 ** 0.5) * torch.sqrt(d2) / lr[:, None, None]
        Kr = sr[:, None, None] ** 2 * (1 + v) * torch.exp(-v)
    (Kr * R).sum().backward()
    for got, ref, nm in ((Xg.grad, Xr.grad, "X"), (Zg.grad, Zr.grad, "Z"), (k.sigma.grad.reshape(-1), sr.grad, "sigma"),
                         (k.lengthscale.grad.reshape(-1), lr.grad, "lengthscale")):
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-8, atol=1e-10, msg=lambda m: f"{nm} d={d}: {m}")
