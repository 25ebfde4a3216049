"""CPU: the gpzoo.kernels / gpzoo.gp class API (constructor signatures, parameter names and
shapes = state_dict keys, post-construction replacement) and the loud failure off-GPU."""
import pytest
import torch
import torch.nn as nn


def test_drop_in_import_names():
    from gpzoo.gp import MGGP_SVGP, MGGP_WSVGP, SVGP, WSVGP  # noqa: F401
    from gpzoo.kernels import (MGGP_NSF_RBF, MGGP_RBF, NSF_RBF, RBF, batched_Matern32,  # noqa: F401
                               batched_MGGP_RBF, batched_RBF)
    from gpzoo.likelihoods import ExactLikelihood, GaussianLikelihood  # noqa: F401
    from gpzoo.utilities import (_embed_distance_matrix, _squared_dist, add_jitter, reshape_param,  # noqa: F401
                                 svgp_forward, whitened_KL)


def test_state_dict_keys_and_shapes():
    from gpzoo.gp import MGGP_SVGP, WSVGP
    from gpzoo.kernels import MGGP_NSF_RBF, NSF_RBF
    m = WSVGP(NSF_RBF(sigma=1.0, lengthscale=2.0, L=5), dim=2, M=30, jitter=1e-3)
    sd = m.state_dict()
    assert set(sd) == {"Z", "Lu", "mu", "kernel.sigma", "kernel.lengthscale"}
    assert sd["Z"].shape == (30, 2) and sd["Lu"].shape == (30, 30) and sd["mu"].shape == (30,)
    assert sd["kernel.sigma"].shape == (5, 1, 1)
    g = MGGP_SVGP(MGGP_NSF_RBF(n_groups=3, L=4), dim=2, M=20, n_groups=3)
    sd = g.state_dict()
    assert {"groupsZ", "kernel.group_diff_param", "kernel.embedding"} <= set(sd)
    assert sd["groupsZ"].dtype == torch.int64 and sd["kernel.embedding"].shape == (3, 3)
    assert g.jitter == 1e-4 and m.jitter == 1e-3


def test_defaults_match_reference():
    from gpzoo.gp import SVGP
    from gpzoo.kernels import MGGP_RBF, RBF, batched_MGGP_RBF
    k = RBF()
    assert float(k.sigma) == 1.0 and float(k.lengthscale) == 2.0 and k.input_dim == 2
    s = SVGP(k)
    assert s.Z.shape == (50, 1) and s.Lu.shape == (50, 50) and float(s.mu.abs().sum()) == 0.0
    assert MGGP_RBF(n_groups=2).embedding.shape == (2, 2)
    assert not isinstance(MGGP_RBF().embedding, nn.Parameter)      # plain tensor, as in the reference
    assert isinstance(batched_MGGP_RBF().embedding, nn.Parameter)
    assert batched_MGGP_RBF().embedding.shape == (10, 10)


def test_diag_contract():
    from gpzoo.kernels import NSF_RBF, RBF, batched_Matern32
    X = torch.randn(7, 2)
    assert RBF(sigma=1.5)(X, X, diag=True).shape == (7,)
    torch.testing.assert_close(RBF(sigma=1.5)(X, X, diag=True), torch.full((7,), 2.25))
    assert NSF_RBF(L=3)(X, X, diag=True).shape == (3, 7)
    k = batched_Matern32()
    k.sigma = nn.Parameter(torch.tensor([1.0, 2.0]))
    assert k(X, X, diag=True).shape == (2, 7)


def test_cpu_tensors_fail_loudly():
    """No CPU fallback: the product path refuses CPU tensors instead of silently using torch."""
    from gpzoo.gp import WSVGP
    from gpzoo.kernels import RBF
    X = torch.randn(10, 2)
    with pytest.raises(RuntimeError, match="GPU only"):
        RBF()(X, X)
    with pytest.raises(RuntimeError, match="GPU only"):
        WSVGP(RBF(), dim=2, M=5)(X)


def test_helpers_match_oracle():
    from gpzoo.utilities import _embed_distance_matrix, add_jitter, whitened_KL
    from oracle import svgp_oracle as O
    D = torch.tensor([[0.0, 1.0, 2.0], [1.0, 0.0, 1.5], [2.0, 1.5, 0.0]])
    torch.testing.assert_close(_embed_distance_matrix(D).abs(), O.embed_group_distances(D).abs(), rtol=1e-5, atol=1e-6)
    K = torch.zeros(2, 4, 4)
    assert add_jitter(K, 0.5) is K and float(K.sum()) == 4.0   # in place, returns the same tensor
    assert add_jitter(torch.zeros(3), 1.0) is None
    mu, Lu = torch.randn(6), torch.randn(6, 6).tril() + 3 * torch.eye(6)
    torch.testing.assert_close(whitened_KL(mu, Lu), O.whitened_kl(mu, Lu))


def test_shard_latents_partition():
    from gpzoo_amd.synthetic import shard_latents
    for L, w in ((256, 8), (32, 8), (5, 3), (3, 8)):
        blocks = [shard_latents(L, w, r) for r in range(w)]
        assert sum(len(b) for b in blocks) == L
        assert [i for b in blocks for i in b] == list(range(L))
    assert len(shard_latents(256, 8, 3)) == 32 and len(shard_latents(256, 2, 1)) == 128


def _state_dict_builders():
    import torch
    import gpzoo.gp as g
    import gpzoo.kernels as k
    import gpzoo.likelihoods as li
    y = torch.ones(7, 11)
    gp_of = lambda cls, kern, **kw: cls(kern, dim=2, M=6, **kw)  # noqa: E731
    return {
        "RBF": lambda: k.RBF(), "NSF_RBF": lambda: k.NSF_RBF(L=3), "batched_RBF": lambda: k.batched_RBF(),
        "batched_Matern32": lambda: k.batched_Matern32(), "MGGP_RBF": lambda: k.MGGP_RBF(n_groups=3),
        "MGGP_NSF_RBF": lambda: k.MGGP_NSF_RBF(L=3, n_groups=3), "batched_MGGP_RBF": lambda: k.batched_MGGP_RBF(n_groups=3),
        "VNNGP": lambda: gp_of(g.VNNGP, k.NSF_RBF(L=3), K=2), "SVGP": lambda: gp_of(g.SVGP, k.RBF()),
        "WSVGP": lambda: gp_of(g.WSVGP, k.NSF_RBF(L=3)),
        "MGGP_SVGP": lambda: gp_of(g.MGGP_SVGP, k.MGGP_RBF(n_groups=3), n_groups=3),
        "MGGP_WSVGP": lambda: gp_of(g.MGGP_WSVGP, k.MGGP_NSF_RBF(L=3, n_groups=3), n_groups=3),
        "GaussianPrior": lambda: g.GaussianPrior(y, L=3),
        "GaussianLikelihood": lambda: li.GaussianLikelihood(gp_of(g.SVGP, k.RBF())),
        "ExactLikelihood": lambda: li.ExactLikelihood(gp_of(g.WSVGP, k.RBF())),
        "PNMF": lambda: li.PNMF(g.GaussianPrior(y, L=3), y, L=3),
        "NSF2": lambda: li.NSF2(gp_of(g.WSVGP, k.NSF_RBF(L=3)), y, L=3),
        "NSF": lambda: li.NSF(gp_of(g.WSVGP, k.NSF_RBF(L=3)), y, L=3),
        "MGGP_NSF": lambda: li.MGGP_NSF(gp_of(g.MGGP_WSVGP, k.MGGP_NSF_RBF(L=3, n_groups=3), n_groups=3), y, L=3),
        "Hybrid_NSF2": lambda: li.Hybrid_NSF2(gp_of(g.WSVGP, k.NSF_RBF(L=3)), g.GaussianPrior(y, L=2), y, L=3, T=2),
        "Hybrid_NSF_Exact": lambda: li.Hybrid_NSF_Exact(gp_of(g.WSVGP, k.NSF_RBF(L=3)), g.GaussianPrior(y, L=2), y, L=3, T=2),
        "Hybrid_NSF": lambda: li.Hybrid_NSF(gp_of(g.WSVGP, k.NSF_RBF(L=3)), y, L=3, non_spatial_factors=2),
    }


def test_state_dict_names_shapes_match_reference():
    """A checkpoint saved by the reference loads here and vice versa: same state_dict keys, shapes and
    dtypes for every model class (fixture written from the reference by tests/golden/make_golden.py)."""
    import json
    import os
    from helpers import GOLDEN
    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as f:
        ref = json.load(f)
    builders = _state_dict_builders()
    assert sorted(builders) == sorted(ref)
    for name, make in builders.items():
        sd = make().state_dict()
        got = {k: [list(v.shape), str(v.dtype)] for k, v in sd.items()}
        assert got == ref[name], f"{name}: {got} != {ref[name]}"


def test_call_signatures_match_reference():
    """Every public constructor / method / helper the notebooks call keeps the reference's parameter names,
    order and defaults (fixture written from the reference by tests/golden/make_golden.py).  Extra trailing
    keyword parameters with defaults (e.g. ``fused=True``) and a trailing ``**kwargs`` are allowed."""
    import importlib
    import inspect
    import json
    import os
    from helpers import GOLDEN
    with open(os.path.join(GOLDEN, "api_signatures.json")) as f:
        ref = json.load(f)
    bad = []
    for key, want in ref.items():
        mod_name, *path = key.split(".", 2)[0:1] + key.split(".")[1:]
        parts = key.split(".")
        mod = importlib.import_module(".".join(parts[:2]))
        obj = mod
        try:
            for a in parts[2:]:
                obj = getattr(obj, a)
        except AttributeError:
            bad.append(f"{key}: missing")
            continue
        got = [[n, str(q.kind), None if q.default is inspect.Parameter.empty else repr(q.default)]
               for n, q in inspect.signature(obj).parameters.items()]
        core = lambda ps: [p for p in ps if p[1] not in ("VAR_KEYWORD", "VAR_POSITIONAL")]  # noqa: E731
        w, g = core(want), core(got)
        w = [[n, k, d if n != "device" else None] for n, k, d in w]       # `device` may gain a default of None here
        g_cmp = [[n, k, d if n != "device" else None] for n, k, d in g[:len(w)]]
        if g_cmp != w or any(p[2] is None and p[0] != "device" for p in g[len(w):]):
            bad.append(f"{key}: reference {want} != {got}")
    assert not bad, "\n".join(bad)


def test_user_defined_covariance_is_refused_not_replaced():
    """The reference's vmap kernels evaluate an overridable ``covariance`` (kernels.py:14-20, 42-47); only the shipped
    closed forms run in HIP, so an override must raise (on any device, before any launch) instead of silently
    computing the built-in formula."""
    import pytest
    import torch
    from gpzoo.kernels import batched_Matern32, batched_RBF

    class MyKernel(batched_RBF):
        def covariance(self, x1, x2):
            return (x1 * x2).sum()

    X = torch.zeros(3, 2)
    with pytest.raises(NotImplementedError, match="user-defined"):
        MyKernel()(X, X)
    k = batched_Matern32()
    k.covariance = lambda a, b: (a - b).abs().sum()
    with pytest.raises(NotImplementedError, match="user-defined"):
        k(X, X)
    with pytest.raises(RuntimeError, match="GPU only"):
        batched_RBF()(X, X)                      # the shipped form passes the check and then wants CUDA tensors


def test_fused_qu_expands_like_a_multivariate_normal():
    """q(U) of the training modules keeps its raw parameter and builds scale_tril lazily; `expand` -- which torch asks a
    subclass with its own __init__ to define -- gives what the reference's plain MultivariateNormal gives."""
    from torch import distributions
    from gpzoo_amd.gp import _FusedQU
    g = torch.Generator().manual_seed(0)
    raw = torch.randn(3, 5, 5, generator=g, dtype=torch.float64)
    loc = torch.randn(3, 5, generator=g, dtype=torch.float64)
    q = _FusedQU(loc, raw=raw)
    tril = raw.tril(-1) + torch.diag_embed(torch.diagonal(raw, dim1=-2, dim2=-1).exp())
    ref = distributions.MultivariateNormal(loc, scale_tril=tril)
    e, r = q.expand((2, 3)), ref.expand((2, 3))
    assert e.batch_shape == r.batch_shape == (2, 3) and e.event_shape == r.event_shape
    torch.testing.assert_close(e.scale_tril, r.scale_tril)
    torch.testing.assert_close(e.loc, r.loc)
    x = torch.randn(2, 3, 5, generator=g, dtype=torch.float64)
    torch.testing.assert_close(e.log_prob(x), r.log_prob(x))
    torch.testing.assert_close(distributions.kl_divergence(e, r), torch.zeros(2, 3, dtype=torch.float64))


def test_reduce_scatter_through_a_c_abi_communicator_alone():
    """parallel._dist_reduce_scatter with only an AbiCommunicator (no torch.distributed group): fp32 goes to
    gpz_reduce_scatter_sum_f32, fp64 to the ABI's all-reduce + slice, anything else is refused by name."""
    from gpzoo_amd import parallel

    class Comm:                                    # the two entry points the function may take, with a 2-rank sum faked
        world, rank = 2, 1
        calls = []
        def reduce_scatter_sum(self, t):
            self.calls.append(("rs32", t.dtype)); return 2 * t[self.rank]
        def allreduce_sum_(self, t):
            self.calls.append(("ar64", t.dtype)); return t.mul_(2)
    c = Comm()
    assert not torch.distributed.is_initialized()
    t32 = torch.arange(12, dtype=torch.float32).reshape(2, 2, 3)
    torch.testing.assert_close(parallel._dist_reduce_scatter(t32, None, c), 2 * t32[1])
    t64 = t32.double()
    keep = t64.clone()
    torch.testing.assert_close(parallel._dist_reduce_scatter(t64, None, c), 2 * keep[1])
    torch.testing.assert_close(t64, keep)          # the caller's tensor is not summed in place
    assert c.calls == [("rs32", torch.float32), ("ar64", torch.float64)]
    with pytest.raises(ValueError, match="float32 or float64"):
        parallel._dist_reduce_scatter(t32.half(), None, c)


def test_deferred_argument_validation_raises_torchs_own_error_at_the_end_of_the_block():
    """ops.checked_dist inside ops.deferred_info(): the distribution is built without its host-synchronising argument
    checks, the checks are queued, and the end of the block raises what the constructor raises; outside a block it is
    the plain constructor."""
    from torch import distributions
    from gpzoo_amd import ops
    good = ops.checked_dist(distributions.Normal, torch.zeros(3), torch.ones(3))
    assert isinstance(good, distributions.Normal) and good._validate_args
    with pytest.raises(ValueError, match="scale"):
        ops.checked_dist(distributions.Normal, torch.zeros(3), torch.tensor([1.0, -1.0, 2.0]))
    with ops.deferred_info() as pend:
        d = ops.checked_dist(distributions.Normal, torch.zeros(3), torch.ones(3))
        assert not d._validate_args and len(pend.flags) == 1
    assert pend.checked
    reached = []
    with pytest.raises(ValueError, match="scale"):
        with ops.deferred_info():
            ops.checked_dist(distributions.Normal, torch.zeros(3), torch.tensor([1.0, float("nan"), 2.0]))
            reached.append(1)
    assert reached == [1]
    with pytest.raises(ValueError, match="rate"):
        with ops.deferred_info():
            ops.checked_dist(distributions.Poisson, torch.tensor([1.0, -2.0]))
    with pytest.raises(KeyError):                        # an exception of the block's own passes through unchecked
        with ops.deferred_info():
            ops.checked_dist(distributions.Normal, torch.zeros(2), -torch.ones(2))
            raise KeyError("user error")
