"""CPU, world_size 2, gloo: latent sharding (L >= ranks) or spot sharding (L < ranks) + scalar
all-reduce reproduce the single-rank ELBO.
The per-rank evaluator is injected (the oracle stands in for the HIP pass, which needs a GPU)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist

from helpers import from_wire, to_wire
import torch.multiprocessing as mp

from gpzoo_amd.synthetic import make_config

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # bench.py


def _oracle_eval(p):
    from oracle import svgp_oracle as O
    e, _, _ = O.elbo_eval(p["kind"], p["whitened"], p["X"], p["y"], p["Z"], p["sigma"], p["lengthscale"], p["mu"],
                          p["Lu_raw"], p["jitter"], p["noise_sd"])
    return e


def _oracle_terms(p):
    """(log-lik sum, KL sum): ELBO of the slice with and without its data term."""
    from oracle import svgp_oracle as O
    Kxx = O.kernel_diag(p["sigma"], p["X"].shape[0])
    Kzx = O.kernel_matrix(p["kind"], p["Z"], p["X"], p["sigma"], p["lengthscale"])
    Kzz = O.add_jitter_(O.kernel_matrix(p["kind"], p["Z"], p["Z"], p["sigma"], p["lengthscale"]).contiguous(), p["jitter"])
    mean, scale, Lu, chol = O.wsvgp_moments(Kxx, Kzx, Kzz, p["mu"], p["Lu_raw"])
    kl = O.whitened_kl(p["mu"], Lu)
    zero = torch.zeros_like(kl)
    return O.gaussian_elbo(p["y"], mean, scale, p["noise_sd"], zero), kl.sum()


def _worker(rank, world, port, L, q, threads=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpzoo_amd.parallel import sharded_elbo
    torch.set_num_threads(threads)
    full = make_config(2, N=400, M=48, L=L, dtype=torch.float64)
    e = sharded_elbo(full, L, local_eval=_oracle_eval, local_terms=_oracle_terms)   # L < world: spots shard
    # a rank may also draw only its own block (what bench.py does): same numbers, same sum
    from gpzoo_amd.synthetic import shard_latents
    own = make_config(2, N=400, M=48, L=L, latents=shard_latents(L, world, rank), dtype=torch.float64)
    own["presharded"] = True
    e2 = sharded_elbo(own, L, local_eval=_oracle_eval)
    if rank == 0:
        q.put((float(e), float(e2)))
    dist.destroy_process_group()


@pytest.mark.parametrize("L", [5, 1])
def test_two_rank_sum_matches_single_rank(L):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, L, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    e, e2 = q.get()
    ref = float(_oracle_eval(make_config(2, N=400, M=48, L=L, dtype=torch.float64)))
    assert e == pytest.approx(ref, rel=1e-12)
    assert e2 == pytest.approx(ref, rel=1e-12)


def test_eight_rank_plan_of_baseline_configs3_sums_to_the_single_rank_elbo():
    """The 8-GPU plan of BASELINE configs[3] rehearsed on the CPU: eight gloo ranks, 256 latents block-sharded 32 per rank
    (the blocks bench.plan_latents deals), one scalar all-reduce -- the sum equals the single-process 256-latent ELBO.  (On
    the GPU pool at most 6 processes may share a card, so this world size is rehearsed here, with the oracle as evaluator.)"""
    L, world = 256, 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, L, q, 1)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    e, e2 = q.get()
    ref = float(_oracle_eval(make_config(2, N=400, M=48, L=L, dtype=torch.float64)))
    assert e == pytest.approx(ref, rel=1e-12)
    assert e2 == pytest.approx(ref, rel=1e-12)
    import bench
    assert [len(bench.plan_latents(3, world, r, None, L, None)[3]) for r in range(world)] == [32] * world


def _grad_worker(rank, world, port, L, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpzoo_amd.parallel import allreduce_shared_grads, shard_problem
    torch.set_num_threads(2)
    full = make_config(2, N=300, M=40, L=L, dtype=torch.float64)
    p = shard_problem(full, L, world, rank)
    Z = p["Z"].clone().requires_grad_(True)              # shared by every latent: its gradient needs the exchange
    mu = p["mu"].clone().requires_grad_(True)            # per latent: owned by this rank
    q_ = dict(p, Z=Z, mu=mu)
    loss = -_oracle_eval(q_)
    loss.backward()
    allreduce_shared_grads([Z])
    q.put(to_wire((rank, p["latents"].start, p["latents"].stop, Z.grad.clone(), mu.grad.clone())))
    dist.barrier()
    dist.destroy_process_group()


def test_shared_parameter_gradients_sum_over_latent_shards():
    """World size 2, gloo: after allreduce_shared_grads the gradient of the (negative) ELBO w.r.t. the shared inducing points
    equals the single-process gradient on every rank; the per-latent gradients are the single-process rows of the shard."""
    L = 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, L, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [from_wire(q.get()) for _ in range(2)]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    full = make_config(2, N=300, M=40, L=L, dtype=torch.float64)
    Z = full["Z"].clone().requires_grad_(True)
    mu = full["mu"].clone().requires_grad_(True)
    (-_oracle_eval(dict(full, Z=Z, mu=mu))).backward()
    for rank, lo, hi, gz, gmu in got:
        torch.testing.assert_close(gz, Z.grad, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(gmu, mu.grad[lo:hi], rtol=1e-10, atol=1e-12)


def _missing_grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpzoo_amd.parallel import allreduce_shared_grads
    a = torch.nn.Parameter(torch.ones(3, dtype=torch.float64))
    b = torch.nn.Parameter(torch.ones(2, 2, dtype=torch.float64))
    # rank 0 uses both parameters, rank 1's shard leaves `b` unused: its .grad stays None there
    loss = (a * (rank + 1.0)).sum() + ((b * 3.0).sum() if rank == 0 else 0.0)
    loss.backward()
    allreduce_shared_grads([a, None, b])
    q.put(to_wire((rank, a.grad.clone(), b.grad.clone())))
    dist.barrier()
    dist.destroy_process_group()


def test_shared_gradient_exchange_with_a_parameter_unused_on_one_rank():
    """The flattened all-reduce is sized from the parameter list, not from the gradients that happen to exist: a
    parameter without a gradient on one rank contributes zeros instead of shortening that rank's buffer (the
    ranks then entered the collective with different lengths -- a hang under RCCL, corruption under gloo)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_missing_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [from_wire(q.get()) for _ in range(2)]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ga, gb in got:
        torch.testing.assert_close(ga, torch.full((3,), 3.0, dtype=torch.float64))
        torch.testing.assert_close(gb, torch.full((2, 2), 3.0, dtype=torch.float64))


# ---- latent-sharded Poisson NSF step (SURVEY §8e caveat / §8f #2): all-gather of q(F), reduce-scatter of its gradient ----

def _torch_local_poisson(mean, scale, eps, W, V, y, with_lgamma):
    """The per-rank evaluation ops.poisson_nsf performs on the GPU, in plain torch: value and the four gradients of
    (1/E) sum_e sum_dn log Poisson(y | V * (W @ exp(mean + scale * eps_e)))   (reference likelihoods.py:49-53, 74-97)."""
    with torch.enable_grad():              # (called from inside an autograd.Function's forward)
        lv = [t.detach().clone().requires_grad_(True) for t in (mean, scale, W, V)]
        rate = lv[3] * torch.matmul(lv[2], torch.exp(lv[0] + lv[1] * eps))
        ll = (torch.distributions.Poisson(rate).log_prob(y) if with_lgamma else y * torch.log(rate) - rate).mean(0).sum()
        g = torch.autograd.grad(ll, lv)
    return (ll.detach(), *g)


class _OracleGP(torch.nn.Module):
    """A whitened SVGP over a block of latents in torch (the oracle's arithmetic), standing in for gpzoo.gp.WSVGP."""

    def __init__(self, Z, lengthscale, mu, Lu_raw, jitter):
        super().__init__()
        self.Z = torch.nn.Parameter(Z.clone())
        self.lengthscale = torch.nn.Parameter(lengthscale.clone())
        self.mu = torch.nn.Parameter(mu.clone())
        self.Lu = torch.nn.Parameter(Lu_raw.clone())
        self.jitter = jitter

    def forward(self, X):
        from oracle import svgp_oracle as O
        sigma = torch.ones_like(self.lengthscale)
        Kxx = O.kernel_diag(sigma, X.shape[0])
        Kzx = O.kernel_matrix("nsf_rbf", self.Z, X, sigma, self.lengthscale)
        Kzz = O.add_jitter_(O.kernel_matrix("nsf_rbf", self.Z, self.Z, sigma, self.lengthscale).contiguous(), self.jitter)
        mean, scale, Lu, _ = O.wsvgp_moments(Kxx, Kzx, Kzz, self.mu, self.Lu)
        return (torch.distributions.Normal(mean, scale),
                torch.distributions.MultivariateNormal(self.mu, scale_tril=Lu), None)


def _nsf_problem(L=5, D=7, N=40, M=6, E=2):
    g = torch.Generator().manual_seed(31)
    f64 = torch.float64
    X = 4.0 * torch.rand(N, 2, generator=g, dtype=f64)
    return dict(X=X, Z=X[torch.randperm(N, generator=g)[:M]].clone(), lengthscale=1.0 + torch.rand(L, generator=g, dtype=f64),
                mu=0.3 * torch.randn(L, M, generator=g, dtype=f64), Lu_raw=0.1 * torch.randn(L, M, M, generator=g, dtype=f64),
                W=torch.randn(D, L, generator=g, dtype=f64), V=0.2 * torch.randn(N, generator=g, dtype=f64),
                y=torch.poisson(3.0 * torch.rand(D, N, generator=g, dtype=f64), generator=g),
                eps=torch.randn(E, L, N, generator=g, dtype=f64), jitter=1e-2, L=L, D=D)


def _nsf_worker(rank, world, port, q):
    try:
        _nsf_worker_body(rank, world, port, q)
    except Exception:                       # the parent must not wait for a result that will never come
        import traceback
        q.put((rank, None, None, traceback.format_exc(), None))
        raise


def _nsf_worker_body(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpzoo_amd.parallel import sharded_nsf_step
    from gpzoo_amd.synthetic import shard_latents
    torch.set_num_threads(2)
    p = _nsf_problem()
    lat, genes = shard_latents(p["L"], world, rank), shard_latents(p["D"], world, rank)
    sl = slice(lat.start, lat.stop)
    gp = _OracleGP(p["Z"], p["lengthscale"][sl], p["mu"][sl], p["Lu_raw"][sl], p["jitter"])
    W = torch.nn.Parameter(p["W"][genes.start:genes.stop].clone())
    V = torch.nn.Parameter(p["V"].clone())
    loss = sharded_nsf_step(gp, p["X"], W, V, p["y"][genes.start:genes.stop], p["eps"], p["L"],
                            local=_torch_local_poisson, shared_params=[gp.Z])
    q.put(to_wire((rank, (lat.start, lat.stop), (genes.start, genes.stop), float(loss),
                   {k: v.grad.clone() for k, v in dict(Z=gp.Z, lengthscale=gp.lengthscale, mu=gp.mu, Lu=gp.Lu, W=W, V=V).items()})))
    dist.barrier()
    dist.destroy_process_group()


def test_latent_sharded_poisson_nsf_step_matches_single_process():
    """World size 2, gloo, 5 latents (3 + 2) and 7 genes (4 + 3): q(F)'s moments are all-gathered, every rank evaluates
    the Poisson terms of its genes over ALL latents, the gradients w.r.t. the moments are reduce-scattered back to the
    latents' owners; the loss and every parameter's gradient equal the single-process step to 1e-9."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_nsf_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [from_wire(q.get()) for _ in range(2)]
    [p.join(180) for p in procs]
    assert all(g[4] is not None for g in got), [g[3] for g in got if g[4] is None]
    assert all(p.exitcode == 0 for p in procs)
    # the single-process step, written out directly
    p = _nsf_problem()
    gp = _OracleGP(p["Z"], p["lengthscale"], p["mu"], p["Lu_raw"], p["jitter"])
    W = torch.nn.Parameter(p["W"].clone())
    V = torch.nn.Parameter(p["V"].clone())
    qF, qU, _ = gp(p["X"])
    F = qF.mean + qF.scale * p["eps"]
    sp = torch.nn.functional.softplus
    rate = sp(V) * torch.matmul(sp(W), torch.exp(F))
    from oracle import svgp_oracle as O
    kl = O.whitened_kl(gp.mu, qU.scale_tril).sum()
    loss = -(torch.distributions.Poisson(rate).log_prob(p["y"]).mean(0).sum() - kl)
    loss.backward()
    ref = dict(Z=gp.Z.grad, lengthscale=gp.lengthscale.grad, mu=gp.mu.grad, Lu=gp.Lu.grad, W=W.grad, V=V.grad)
    for rank, (l0, l1), (d0, d1), lv, grads in got:
        assert lv == pytest.approx(float(loss), rel=1e-12)
        for k in ("Z", "V"):
            torch.testing.assert_close(grads[k], ref[k], rtol=1e-9, atol=1e-12, msg=lambda m: f"rank {rank} {k}: {m}")
        for k in ("lengthscale", "mu", "Lu"):
            torch.testing.assert_close(grads[k], ref[k][l0:l1], rtol=1e-9, atol=1e-12, msg=lambda m: f"rank {rank} {k}: {m}")
        torch.testing.assert_close(grads["W"], ref["W"][d0:d1], rtol=1e-9, atol=1e-12)
