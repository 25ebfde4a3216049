"""GPU: the product path of the multi-rank evaluation -- two processes (gloo rendezvous, both on the one
GPU of the test box) run gpzoo_amd.parallel.sharded_elbo with the HIP evaluator: latent sharding (L >=
ranks) and spot sharding (L < ranks) both reproduce the single-process ELBO."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from helpers import from_wire, to_wire
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, L, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpzoo_amd.parallel import sharded_elbo
    from gpzoo_amd.synthetic import make_config
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    c = make_config(2, N=3000, M=200, L=L, dtype=torch.float64)
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    e = sharded_elbo(g, L).cpu()        # all-reduce of the fp64 scalar goes through gloo on the host copy
    if rank == 0:
        q.put(float(e))
    dist.destroy_process_group()


@pytest.mark.parametrize("L", [5, 1])
def test_two_rank_hip_sum_matches_single_process(L):
    from gpzoo_amd.parallel import hip_local_elbo
    from gpzoo_amd.synthetic import make_config
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, L, q)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    c = make_config(2, N=3000, M=200, L=L, dtype=torch.float64)
    g = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    ref = float(hip_local_elbo(g))
    assert q.get() == pytest.approx(ref, rel=1e-11)


def _model(c, lat, dev):
    """WSVGP + NSF_RBF on the latents `lat` of problem c (the whole problem when lat covers all of them)."""
    import torch.nn as nn
    import gpzoo.gp as G
    import gpzoo.kernels as K
    L = len(lat)
    k = K.NSF_RBF(L=L, sigma=1.0, lengthscale=1.0)
    k.sigma = nn.Parameter(c["sigma"][lat.start:lat.stop].reshape(L, 1, 1).clone())
    k.lengthscale = nn.Parameter(c["lengthscale"][lat.start:lat.stop].reshape(L, 1, 1).clone())
    m = G.WSVGP(k, dim=2, M=c["Z"].shape[0], jitter=c["jitter"])
    m.Z = nn.Parameter(c["Z"].clone())
    m.mu = nn.Parameter(c["mu"][lat.start:lat.stop].clone())
    m.Lu = nn.Parameter(c["Lu_raw"][lat.start:lat.stop].clone())
    return m.to(dev)


def _loss(m, X, y):
    qF, qU, _ = m(X)
    return ((qF.mean - y) ** 2).sum() + qF.scale.sum() + (qU.scale_tril ** 2).sum()


def _train_worker(rank, world, port, L, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpzoo_amd.parallel import allreduce_shared_grads
    from gpzoo_amd.synthetic import make_config, shard_latents
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    c = make_config(2, N=2000, M=150, L=L, dtype=torch.float64)
    lat = shard_latents(L, world, rank)
    m = _model(c, lat, dev)
    _loss(m, c["X"].to(dev), c["y"][lat.start:lat.stop].to(dev)).backward()
    allreduce_shared_grads([m.Z])
    q.put(to_wire((lat.start, lat.stop, m.Z.grad.cpu(), m.mu.grad.cpu(), m.kernel.lengthscale.grad.cpu())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_gradients_match_single_process():
    """Two processes, each with its block of latents behind the module API (WSVGP + NSF_RBF, HIP forward and backward):
    after parallel.allreduce_shared_grads the gradient of the shared inducing points equals the single-process one on
    both ranks; per-latent gradients are the single-process rows of the block."""
    from gpzoo_amd.synthetic import make_config
    L = 6
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, L, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [from_wire(q.get()) for _ in range(2)]
    [p.join(300) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    c = make_config(2, N=2000, M=150, L=L, dtype=torch.float64)
    m = _model(c, range(L), torch.device("cuda", 0))
    _loss(m, c["X"].cuda(), c["y"].cuda()).backward()
    for lo, hi, gz, gmu, gell in got:
        torch.testing.assert_close(gz, m.Z.grad.cpu(), rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(gmu, m.mu.grad[lo:hi].cpu(), rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(gell, m.kernel.lengthscale.grad[lo:hi].cpu(), rtol=1e-9, atol=1e-11)
