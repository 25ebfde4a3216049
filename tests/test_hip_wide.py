"""GPU: the three ways the forward pass can run its two big fp32 products give the same numbers.

  narrow + kfill   kfill_kernel, then gemm128_kernel for both products (the path of round 2 and of every other precision)
  wide + kfill     kfill_kernel, then gemmw_kernel<128,256,mem> for both products (the default)
  generated        gemmw_kernel<512,128,gen> (16 waves): stage 1 computes its covariance operand itself, Kzx is never written
  panel            kfill_kernel, then panel_kernel (csrc/gemmp.hip; fp32, M <= 512, on request): BOTH products in one launch, a
                   workgroup holding 64 columns x all rows in LDS; Wt is stored only when retained

cov.h is shared by the fill and the generator and all three kernels give lane group q the k = 4q..4q+3 slots of a
16-deep chunk, so Wt must agree BIT FOR BIT; the column statistics are summed in different orders, so mean / scale /
ELBO agree to fp32 rounding.  Against the oracle: the usual north_star tolerances (helpers.rtol_for)."""
import pytest
import torch

from helpers import rtol_for

pytestmark = pytest.mark.gpu

SHAPES = [
    # cfg, N, M, L, d
    (3, 12288, 2048, 2, 2),      # one config-3 chunk: 16 blocks (2 rounds of workgroups: single row tiles per workgroup)
    (3, 12288, 2048, 24, 2),     # ... with enough latents for the paired schedule (pairs of row tiles, column-major)
    (3, 5000, 2048, 1, 2),       # ragged last column tile
    (3, 3000, 3000, 2, 2),       # Mp = 3072, padded rows
    (3, 2000, 384, 3, 2),        # 3 blocks: partial last row tile of both tile heights
    (3, 1500, 640, 2, 2),        # 5 blocks
    (3, 777, 100, 2, 2),         # a single block
    (2, 20000, 512, 8, 2),       # config 2's kernel (RBF), four row tiles, single tiles per workgroup
    (2, 50000, 512, 8, 2),       # configs[1] itself: pairs of row tiles, column-major, a partial last column tile
    (2, 4000, 640, 3, 1),        # 1-D inputs
    (3, 4000, 640, 3, 1),
]


def _problem(cfg, N, M, L, d):
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    c = make_config(cfg, N=N, M=M, L=L)
    if d == 1:
        c["X"], c["Z"] = c["X"][:, :1].contiguous(), c["Z"][:, :1].contiguous()
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, torch.device("cuda", 0))
    return c, g, spec, extra


def _run(c, g, spec, extra, **kw):
    from gpzoo_amd import ops
    return ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                            noise_sd=c["noise_sd"], want_Lu=False, retain_wt=0.9, **extra, **kw)


@pytest.mark.parametrize("cfg,N,M,L,d", SHAPES)
def test_three_paths_agree_bitwise_in_wt(cfg, N, M, L, d):
    c, g, spec, extra = _problem(cfg, N, M, L, d)
    Mp, ncp = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    nwt = L * Mp * ncp
    ref = _run(c, g, spec, extra, materialize_kzx=True, narrow_tiles=True)
    wref = ref["wt_cache"].view(torch.int32)[:nwt]
    assert int(wref.count_nonzero()) > nwt // 4            # the comparison below is not between two empty buffers
    for kw in (dict(), dict(materialize_kzx=True), dict(materialize_kzx=False)):
        out = _run(c, g, spec, extra, **kw)
        assert torch.equal(out["wt_cache"].view(torch.int32)[:nwt], wref), kw
        torch.testing.assert_close(out["mean"], ref["mean"], rtol=1e-5, atol=1e-5 * float(ref["mean"].abs().max()))
        torch.testing.assert_close(out["scale"], ref["scale"], rtol=1e-5, atol=0)      # strictly positive: pure rtol
        assert float(out["elbo"]) == pytest.approx(float(ref["elbo"]), rel=1e-7)
        assert torch.equal(out["kl"], ref["kl"])


def _random_shapes(n, seed=11):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        N = int(torch.randint(200, 40000, (1,), generator=g))
        M = int(torch.randint(20, 1700, (1,), generator=g))
        L = int(torch.randint(1, 9, (1,), generator=g))
        out.append((N, M, L))
    return out


@pytest.mark.parametrize("N,M,L", _random_shapes(14) + [(30000, 1100, 8), (26000, 640, 16)])   # + two paired launches with an odd block count
def test_wide_tiles_agree_with_narrow_tiles_on_random_shapes(N, M, L):
    """Odd counts of row and column tiles, partial last tiles, one to a dozen strips, both schedules of the wide kernel
    (single row tiles / pairs dispatched column-major, decided by the launch size): Wt bit for bit the narrow kernel's."""
    c, g, spec, extra = _problem(3, N, M, L, 2)
    Mp, ncp = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    nwt = L * Mp * ncp
    ref = _run(c, g, spec, extra, materialize_kzx=True, narrow_tiles=True)
    out = _run(c, g, spec, extra)
    assert torch.equal(out["wt_cache"].view(torch.int32)[:nwt], ref["wt_cache"].view(torch.int32)[:nwt])
    torch.testing.assert_close(out["scale"], ref["scale"], rtol=1e-5, atol=0)
    assert float(out["elbo"]) == pytest.approx(float(ref["elbo"]), rel=1e-7)


PANEL_SHAPES = [
    # cfg, N, M, L, whitened, d
    (2, 50000, 512, 8, True, 2),        # configs[1] itself: 16 waves, one latent per XCD
    (2, 20000, 512, 3, True, 2),        # fewer latents than XCDs: an XCD's range of panels spans two latents
    (3, 5000, 500, 5, True, 2),         # Mp = 512 with padded rows; Matern-3/2; ragged last panel (N not a multiple of 64)
    (3, 2000, 384, 3, True, 2),         # 12 row blocks (three waves per SIMD)
    (2, 9000, 256, 11, True, 2),        # 8 row blocks, two workgroups per CU
    (3, 777, 100, 2, True, 2),          # 4 row blocks: a single 128-block
    (3, 3000, 300, 4, False, 2),        # un-whitened SVGP: LuE = Linv Lu, clamp
    (2, 64, 40, 1, True, 2),            # one panel in all
    (2, 4000, 200, 3, True, 1),         # 1-D inputs (RBF)
    (3, 4000, 333, 3, True, 1),         # 1-D inputs (Matern-3/2)
    (5, 6000, 300, 3, True, 2),         # multi-group RBF in fp32: cov.h does not cover it, the panel is read from the fill's Kzx
    (5, 3000, 512, 2, True, 2),         # ... at four 128-blocks (16 waves), and at one (not the library's own choice there)
    (5, 3000, 128, 2, True, 2),
]


@pytest.mark.parametrize("cfg,N,M,L,whitened,d", PANEL_SHAPES)
def test_panel_kernel_agrees_with_the_tile_kernels(cfg, N, M, L, whitened, d):
    """The one-launch panel path (GPZ_SVGP_PANEL_PRODUCTS; its covariance panel computed inside the kernel where cov.h
    covers the kernel family, read from the stand-alone fill's Kzx otherwise): the library reports it took it, the
    retained Wt is bit for bit the tile kernels', mean / scale / ELBO agree to fp32 rounding (its column statistics are
    summed wave by wave), and without retention (the evaluation path: Wt never reaches memory) the moments are the same
    bits as with it."""
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    c = make_config(cfg, N=N, M=M, L=L, dtype=torch.float32)
    c["whitened"] = whitened
    if d == 1:
        c["X"], c["Z"] = c["X"][:, :1].contiguous(), c["Z"][:, :1].contiguous()
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, torch.device("cuda", 0))
    Mp, ncp = (M + 127) // 128 * 128, (N + 127) // 128 * 128
    nwt = L * Mp * ncp
    ref = _run(c, g, spec, extra, materialize_kzx=True)          # names the fill + tile-product path
    out = _run(c, g, spec, extra, panel_products=True)
    assert ref["path"] in (0, 1) and out["path"] == 4
    # left to itself the library takes the panel kernel where it measured faster -- wherever it computes the covariance
    # itself (RBF / Matern-3/2), else for 128 < M <= 512 -- retained Wt or not
    own_choice = cfg in (2, 3) or Mp in (256, 384, 512)
    assert (_run(c, g, spec, extra)["path"] == 4) == own_choice
    assert torch.equal(out["wt_cache"].view(torch.int32)[:nwt], ref["wt_cache"].view(torch.int32)[:nwt])
    torch.testing.assert_close(out["mean"], ref["mean"], rtol=1e-5, atol=1e-5 * float(ref["mean"].abs().max()))
    torch.testing.assert_close(out["scale"], ref["scale"], rtol=1e-5, atol=0)
    assert float(out["elbo"]) == pytest.approx(float(ref["elbo"]), rel=1e-7)
    assert torch.equal(out["kl"], ref["kl"])
    bare = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                            noise_sd=c["noise_sd"], want_Lu=False, panel_products=True, **extra)
    assert bare["path"] == 4 and "wt_cache" not in bare
    auto = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                            noise_sd=c["noise_sd"], want_Lu=False, **extra)
    assert (auto["path"] == 4) == own_choice
    assert torch.equal(bare["mean"], out["mean"]) and torch.equal(bare["scale"], out["scale"])
    assert float(bare["elbo"]) == float(out["elbo"])


def test_panel_kernel_in_chunks_and_against_the_oracle():
    """N=3000, M=300, L=3 Matern-3/2 fp32 on the panel path, whole and in three chunks of columns, against the CPU oracle;
    and the request is ignored where the kernel does not apply (M > 512, fp64): the tile kernels run."""
    from oracle import svgp_oracle as O
    from gpzoo_amd import ops
    c, g, spec, extra = _problem(3, 3000, 300, 3, 2)
    e, mean, scale = O.elbo_eval(c["kind"], c["whitened"], c["X"], c["y"], c["Z"], c["sigma"], c["lengthscale"], c["mu"],
                                 c["Lu_raw"], c["jitter"], c["noise_sd"])
    rt = rtol_for(torch.float32)
    for chunk in (0, 1024):
        out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                               noise_sd=c["noise_sd"], want_Lu=False, panel_products=True, chunk=chunk, **extra)
        assert out["path"] == 4
        torch.testing.assert_close(out["mean"].cpu(), mean, rtol=rt, atol=rt * float(mean.abs().max()))
        torch.testing.assert_close(out["scale"].cpu(), scale, rtol=rt, atol=0)
        assert float(out["elbo"]) == pytest.approx(float(e), rel=rt)
    c, g, spec, extra = _problem(3, 2000, 640, 2, 2)
    assert _run(c, g, spec, extra, panel_products=True)["path"] == 1


@pytest.mark.parametrize("materialize", [None, True, False])
def test_each_path_against_the_oracle(materialize):
    """N=3000, M=300, L=3 Matern-3/2 fp32 (a partial 512-row tile, padded rows and columns) against the CPU oracle."""
    from oracle import svgp_oracle as O
    c, g, spec, extra = _problem(3, 3000, 300, 3, 2)
    out = _run(c, g, spec, extra, materialize_kzx=materialize)
    e, mean, scale = O.elbo_eval(c["kind"], c["whitened"], c["X"], c["y"], c["Z"], c["sigma"], c["lengthscale"], c["mu"],
                                 c["Lu_raw"], c["jitter"], c["noise_sd"])
    rt = rtol_for(torch.float32)
    torch.testing.assert_close(out["mean"].cpu(), mean, rtol=rt, atol=rt * float(mean.abs().max()))
    torch.testing.assert_close(out["scale"].cpu(), scale, rtol=rt, atol=0)
    assert float(out["elbo"]) == pytest.approx(float(e), rel=rt)


def test_generated_operand_does_not_read_the_kzx_buffer():
    """The generated path must not depend on what the workspace (where the materialised paths keep their Kzx chunk)
    holds: poison it with NaN bit patterns, evaluate, and compare with the materialised path bit for bit."""
    from gpzoo_amd import ops
    c, g, spec, extra = _problem(3, 6000, 1024, 2, 2)
    a = _run(c, g, spec, extra, materialize_kzx=True)
    assert ops._workspaces
    for t in ops._workspaces.values():
        t.fill_(0xFF)
    b = _run(c, g, spec, extra, materialize_kzx=False)
    nwt = 2 * 1024 * 6016
    assert torch.equal(a["wt_cache"].view(torch.int32)[:nwt], b["wt_cache"].view(torch.int32)[:nwt])
    assert torch.isfinite(b["mean"]).all() and torch.isfinite(b["scale"]).all()


def test_panel_kernel_does_not_read_the_kzx_buffer():
    """M <= 512, RBF / Matern-3/2: the panel kernel computes its covariance panel itself -- no fill launch, nothing read from
    the workspace where the other paths keep their Kzx chunk: poison it with NaN bit patterns, evaluate, and compare with
    the fill + tile-kernel path bit for bit."""
    from gpzoo_amd import ops
    c, g, spec, extra = _problem(3, 6000, 500, 2, 2)
    a = _run(c, g, spec, extra, materialize_kzx=True)
    assert ops._workspaces
    for t in ops._workspaces.values():
        t.fill_(0xFF)
    b = _run(c, g, spec, extra)
    assert b["path"] == 4
    nwt = 2 * 512 * 6016
    assert torch.equal(a["wt_cache"].view(torch.int32)[:nwt], b["wt_cache"].view(torch.int32)[:nwt])
    assert torch.isfinite(b["mean"]).all() and torch.isfinite(b["scale"]).all()


@pytest.mark.parametrize("whitened", [True, False])
@pytest.mark.parametrize("N,M,L,retain", [(5000, 640, 2, True), (5000, 640, 2, False), (3000, 1024, 1, True), (2000, 384, 3, False)])
def test_backward_wide_matches_narrow(N, M, L, retain, whitened):
    """The backward pass's fp32 products on the wide kernels (W = Linv Kzx, Pbar with the column scale, Wbar, Kbar_x and the
    two A B^T accumulations over the N-chunk) against the 128 x 128-tile kernel: every gradient agrees to fp32 rounding.
    (The products keep the k order; the column statistics of a recomputed W are summed in a different order, and with few
    tiles -- these shapes -- the wide A B^T accumulation cuts its k extent into pieces that run side by side and are added
    up afterwards: another order of the fp32 sums, 5e-5 on the sigma / lengthscale gradients that sum everything.)"""
    from gpzoo_amd import ops
    c, g, spec, extra = _problem(3, N, M, L, 2)
    out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], whitened, want_Lu=False,
                           retain_wt=0.9 if retain else 0.0, **extra)
    gen = torch.Generator().manual_seed(5)
    gm = torch.randn(out["mean"].shape, generator=gen).cuda()
    gs = torch.randn(out["scale"].shape, generator=gen).cuda()
    # ... and both forms of the pass -- "classic": the products autograd would run (Pbar, W Pbar^T, Wbar, Kbar_x, Kbar_x W^T);
    # "algebra": H += W diag(gv2) W^T, one dense product for Kbar_x and M x M products -- on both kernel families, with and
    # without the kernel / Z gradients
    for kg in (True, False):
        ref = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], whitened, gm, gs, out["scale"],
                                kernel_grads=kg, wt_cache=out.get("wt_cache"), narrow_tiles=True, form="classic", **extra)
        for narrow, form in ((False, "classic"), (True, "algebra"), (False, "algebra"), (False, None)):
            got = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], whitened, gm, gs, out["scale"],
                                    kernel_grads=kg, wt_cache=out.get("wt_cache"), narrow_tiles=narrow, form=form, **extra)
            for a, b, what in zip(ref, got, ("grad_mu", "grad_Lu", "grad_theta", "grad_Z")):
                a, b = a.double(), b.double()
                assert torch.isfinite(b).all(), what
                torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-4 * float(a.abs().max()),
                                           msg=lambda m: f"{what} (narrow={narrow}, form={form}, kernel_grads={kg}): {m}")


def test_backward_wide_matches_narrow_on_the_paired_schedule():
    """One config-3 chunk (N=12288, M=2048, L=32: 24 rounds of workgroups) puts every wide product of the backward pass on
    the paired schedule (two row tiles per workgroup, column-major dispatch); same comparison as above (whose shapes run
    both schedules: N=5000, M=640 is 2.3 rounds, the others less)."""
    from gpzoo_amd import ops
    c, g, spec, extra = _problem(3, 12288, 2048, 32, 2)
    out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, want_Lu=False, retain_wt=0.5, **extra)
    gen = torch.Generator().manual_seed(6)
    gm = torch.randn(out["mean"].shape, generator=gen).cuda()
    gs = torch.randn(out["scale"].shape, generator=gen).cuda()
    ref = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, gm, gs, out["scale"],
                            kernel_grads=True, wt_cache=out.get("wt_cache"), narrow_tiles=True, form="classic", **extra)
    for form in ("classic", "algebra"):
        got = ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], True, gm, gs, out["scale"],
                                kernel_grads=True, wt_cache=out.get("wt_cache"), form=form, **extra)
        for a, b, what in zip(ref, got, ("grad_mu", "grad_Lu", "grad_theta", "grad_Z")):
            a, b = a.double(), b.double()
            assert torch.isfinite(b).all(), what
            torch.testing.assert_close(b, a, rtol=5e-5, atol=5e-5 * float(a.abs().max()), msg=lambda m: f"{what} ({form}): {m}")


def test_full_benchmark_size_paths_agree():
    """BASELINE configs[2] at its full size (N=200 000, M=2048, L=32, 17 chunks): the retained Wt (its 16 full chunks) of the default path, of
    the generated-operand path and of the 128 x 128-tile path (52 GB each) have the same bits -- compared through a 64-bit
    sum and an xor-fold of their int32 views -- and q(F) / the ELBO agree to fp32 rounding."""
    from gpzoo_amd import ops
    c, g, spec, extra = _problem(3, 200_000, 2048, 32, 2)

    def digest(**kw):
        out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                               noise_sd=c["noise_sd"], want_Lu=False, retain_wt=0.6, **extra, **kw)
        assert "wt_cache" in out
        w = out.pop("wt_cache").view(torch.int32)
        nfull = 200_000 // 12_288                               # the 16 full chunks (the last slot is only partly written)
        nwt = nfull * 32 * 2048 * 12_288                        # their Wt slots; the column-sum slabs behind them differ in order
        w = w[:nwt].view(-1, 1 << 20)
        s = int(w.sum(dim=1, dtype=torch.int64).sum())
        x = w[:, 0].clone()
        for j in range(1, 64):
            x ^= w[:, j * 1000]
        res = (s, int(x.sum(dtype=torch.int64)), out["mean"], out["scale"], float(out["elbo"]))
        del w, out
        torch.cuda.empty_cache()
        return res

    ref = digest(materialize_kzx=True, narrow_tiles=True)
    for kw in (dict(), dict(materialize_kzx=False)):
        got = digest(**kw)
        assert got[0] == ref[0] and got[1] == ref[1], kw
        torch.testing.assert_close(got[2], ref[2], rtol=1e-5, atol=1e-5 * float(ref[2].abs().max()))
        torch.testing.assert_close(got[3], ref[3], rtol=1e-5, atol=0)
        assert got[4] == pytest.approx(ref[4], rel=1e-7)
