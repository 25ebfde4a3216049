"""GPU: the product path of the multi-rank evaluation -- two processes (gloo rendezvous, both on the one
GPU of the test box) run gpzoo_amd.parallel.sharded_elbo with the HIP evaluator: latent sharding (L >=
ranks) and spot sharding (L < ranks) both reproduce the single-process ELBO."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
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
