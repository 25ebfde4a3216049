"""GPU: the HIP path reproduces the reference's golden vectors (through the C ABI)."""
import pytest
import torch

from conftest import golden_cases
from helpers import load_case, rtol_for, spec_from_case

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", golden_cases())
def test_kernel_matrices(name):
    from gpzoo_amd import ops
    c = load_case(name)
    d = dev()
    spec = spec_from_case(c, d)
    X, Z = c["X"].to(d), c["Z"].to(d)
    g = dict(gA=c["gZ"].to(d), gB=c["gX"].to(d)) if "gX" in c else {}
    Kzx = ops.kfill(spec, Z, X, **g)
    rt = rtol_for(X.dtype)
    torch.testing.assert_close(Kzx.cpu(), c["Kzx"], rtol=rt, atol=rt * 1e-2)
    g = dict(gA=c["gZ"].to(d), gB=c["gZ"].to(d)) if "gX" in c else {}
    Kzz = ops.kfill(spec, Z, Z, jitter=c["jitter"], **g)
    torch.testing.assert_close(Kzz.cpu(), c["Kzz_jit"], rtol=rt, atol=rt * 1e-2)


@pytest.mark.parametrize("name", golden_cases())
def test_cholesky(name):
    from gpzoo_amd import ops
    c = load_case(name)
    Lc = ops.cholesky(c["Kzz_jit"].to(dev()))
    rt = rtol_for(c["X"].dtype)
    torch.testing.assert_close(Lc.cpu(), c["chol"], rtol=rt, atol=rt * 1e-2)


@pytest.mark.parametrize("name", golden_cases())
def test_forward_and_elbo(name):
    from gpzoo_amd import ops
    c = load_case(name)
    d = dev()
    spec = spec_from_case(c, d)
    dt = c["X"].dtype
    L = spec.L
    kw = dict(gX=c["gX"].to(d), gZ=c["gZ"].to(d)) if "gX" in c else {}
    clamp = 5e-2 if name.startswith("mggp_svgp") else 1e-6
    out = ops.svgp_forward(spec, c["X"].to(d), c["Z"].to(d), c["mu"].to(d), c["Lu_raw"].to(d), c["jitter"],
                           c["whitened"], y=c["y"].to(d), noise_sd=c["noise_sd"], clamp_min=clamp,
                           want_chol=True, **kw)
    rt = rtol_for(dt)
    shp = c["mean"].shape
    torch.testing.assert_close(out["mean"].cpu().reshape(shp), c["mean"], rtol=rt, atol=rt * 1e-1)
    torch.testing.assert_close(out["scale"].cpu().reshape(shp), c["scale"], rtol=rt, atol=rt * 1e-1)
    torch.testing.assert_close(out["Lu"].cpu().reshape(c["Lu"].shape), c["Lu"], rtol=rt, atol=rt * 1e-2)
    torch.testing.assert_close(out["chol"].cpu().reshape(c["chol"].shape), c["chol"], rtol=rt, atol=rt * 1e-2)
    torch.testing.assert_close(out["kl"].cpu().to(dt).reshape(c["kl"].shape), c["kl"], rtol=rt, atol=rt * 1e-2)
    assert float(out["elbo"]) == pytest.approx(c["elbo"], rel=rt)
