"""GPU: the latent-sharded Poisson NSF step (SURVEY §8e caveat, §8f #2; reference likelihoods.py:49-53, 74-97 with the
minibatch objective of utilities.py:600-632) on the product path -- fused HIP GP pass per latent block, all-gather of
q(F)'s moments, gpz_poisson_nsf on the rank's genes, reduce-scatter of the moment gradients:
  * two processes on the one GPU of the test box (gloo rendezvous) against the single-process step;
  * one rank over RCCL (torch's nccl backend) and over the C-ABI communicator (gpz_allgather,
    gpz_reduce_scatter_sum_f32, gpz_allreduce_sum_f32): the device-side branches the 8-GPU job takes."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

from helpers import from_wire, to_wire
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(L=6, D=37, N=600, M=40, E=3):
    g = torch.Generator().manual_seed(123)
    f64 = torch.float64
    X = 20.0 * torch.rand(N, 2, generator=g, dtype=f64)
    return dict(X=X, Z=X[torch.randperm(N, generator=g)[:M]].clone(), lengthscale=3.0 + 2.0 * torch.rand(L, generator=g, dtype=f64),
                mu=0.3 * torch.randn(L, M, generator=g, dtype=f64), Lu_raw=0.05 * torch.randn(L, M, M, generator=g, dtype=f64),
                W=torch.randn(D, L, generator=g, dtype=f64), V=0.2 * torch.randn(N, generator=g, dtype=f64),
                y=torch.poisson(3.0 * torch.rand(D, N, generator=g, dtype=f64), generator=g),
                eps=torch.randn(E, L, N, generator=g, dtype=f64), jitter=1e-2, L=L, D=D)


def _gp(p, lat, dev):
    """gpzoo.gp.WSVGP + NSF_RBF on the latents `lat` (HIP forward and backward behind the module API)."""
    import torch.nn as nn
    import gpzoo.gp as G
    import gpzoo.kernels as K
    L = len(lat)
    k = K.NSF_RBF(L=L, sigma=1.0, lengthscale=1.0)
    k.sigma = nn.Parameter(torch.ones(L, 1, 1, dtype=torch.float64), requires_grad=False)
    k.lengthscale = nn.Parameter(p["lengthscale"][lat.start:lat.stop].reshape(L, 1, 1).clone())
    m = G.WSVGP(k, dim=2, M=p["Z"].shape[0], jitter=p["jitter"])
    m.Z = nn.Parameter(p["Z"].clone())
    m.mu = nn.Parameter(p["mu"][lat.start:lat.stop].clone())
    m.Lu = nn.Parameter(p["Lu_raw"][lat.start:lat.stop].clone())
    return m.double().to(dev)


def _step(p, lat, genes, dev, **kw):
    import torch.nn as nn
    from gpzoo_amd.parallel import sharded_nsf_step
    gp = _gp(p, lat, dev)
    W = nn.Parameter(p["W"][genes.start:genes.stop].clone().to(dev))
    V = nn.Parameter(p["V"].clone().to(dev))
    loss = sharded_nsf_step(gp, p["X"].to(dev), W, V, p["y"][genes.start:genes.stop].to(dev), p["eps"].to(dev), p["L"],
                            shared_params=[gp.Z], **kw)
    grads = dict(Z=gp.Z.grad, lengthscale=gp.kernel.lengthscale.grad.reshape(-1), mu=gp.mu.grad, Lu=gp.Lu.grad, W=W.grad,
                 V=V.grad)
    return float(loss), {k: v.detach().double().cpu() for k, v in grads.items()}


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from gpzoo_amd.synthetic import shard_latents
        dev = torch.device("cuda", rank % torch.cuda.device_count())
        torch.cuda.set_device(dev)
        p = _problem()
        lat, genes = shard_latents(p["L"], world, rank), shard_latents(p["D"], world, rank)
        loss, grads = _step(p, lat, genes, dev)
        q.put(to_wire((rank, (lat.start, lat.stop), (genes.start, genes.stop), loss, grads)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, None, None, traceback.format_exc(), None))
        raise


def _check(got, ref_loss, ref):
    for rank, (l0, l1), (d0, d1), loss, grads in got:
        assert loss == pytest.approx(ref_loss, rel=2e-6), rank
        want = dict(Z=ref["Z"], V=ref["V"], lengthscale=ref["lengthscale"][l0:l1], mu=ref["mu"][l0:l1], Lu=ref["Lu"][l0:l1],
                    W=ref["W"][d0:d1])
        for k, w in want.items():       # the Poisson kernel works in fp32 and sums its gene blocks in another order
            torch.testing.assert_close(grads[k], w, rtol=2e-3, atol=2e-3 * float(ref[k].abs().max()),
                                       msg=lambda m: f"rank {rank} {k}: {m}")


def test_two_rank_hip_step_matches_single_process():
    """Latents 6 = 3 + 3, genes 37 = 19 + 18 over two processes sharing the GPU: the loss of the whole model and every
    parameter's gradient equal the single-process fused step (fp32 Poisson arithmetic: 2e-3 of each gradient's scale)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = [from_wire(q.get()) for _ in range(2)]
    [p.join(300) for p in procs]
    assert all(g[4] is not None for g in got), [g[3] for g in got if g[4] is None]
    assert all(p.exitcode == 0 for p in procs)
    p = _problem()
    ref_loss, ref = _step(p, range(p["L"]), range(p["D"]), torch.device("cuda", 0))
    _check(got, ref_loss, ref)
    # ... and the single-process step is the plain torch evaluation of the reference's formulas (fp64)
    import torch.nn.functional as Fn
    gp = _gp(p, range(p["L"]), torch.device("cuda", 0))
    qF, qU, _ = gp(p["X"].cuda())
    rate = Fn.softplus(p["V"].cuda()) * torch.matmul(Fn.softplus(p["W"].cuda()), torch.exp(qF.mean + qF.scale * p["eps"].cuda()))
    from gpzoo_amd.parallel import _whitened_kl
    plain = -(torch.distributions.Poisson(rate).log_prob(p["y"].cuda()).mean(0).sum() - _whitened_kl(qU).sum())
    assert ref_loss == pytest.approx(float(plain), rel=2e-6)


_CHILD = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import test_hip_poisson_sharded as T
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
p = T._problem()
ref_loss, ref = T._step(p, range(p["L"]), range(p["D"]), dev)          # no group, no communicator: no exchange at all
mode = {mode!r}
if mode == "nccl":
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    loss, grads = T._step(p, range(p["L"]), range(p["D"]), dev)         # all_gather_into_tensor / reduce_scatter_tensor on RCCL
    extra = dict(backend=dist.get_backend())
    dist.destroy_process_group()
else:
    from gpzoo_amd.parallel import AbiCommunicator
    comm = AbiCommunicator(dev)                                         # gpz_comm_init, no torch.distributed
    a = torch.arange(12, dtype=torch.float32, device=dev).reshape(1, 3, 4)
    extra = dict(gather=comm.allgather(a[0]).reshape(-1).tolist(), scatter=comm.reduce_scatter_sum(a).reshape(-1).tolist(),
                 reduce=comm.allreduce_sum_f32_(a.clone()).reshape(-1).tolist())
    loss, grads = T._step(p, range(p["L"]), range(p["D"]), dev, comm=comm)
    torch.cuda.synchronize()
    comm.close()
err = max(float((grads[k] - ref[k]).abs().max()) for k in ref)
print(json.dumps(dict(loss=loss, ref=ref_loss, err=err, **extra)))
"""


@pytest.mark.parametrize("mode", ["nccl", "abi"])
def test_one_rank_exchange_runs_on_the_device(mode):
    """The collectives of the sharded step executed on device buffers with one rank -- RCCL through torch.distributed's
    nccl backend, and through the C ABI (gpz_allgather / gpz_reduce_scatter_sum_f32 / gpz_allreduce_sum_f32): identity
    exchanges, so loss and gradients are bitwise those of the step without any group."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, mode=mode, port=port)], capture_output=True, text=True,
                        timeout=420, env=env)
    assert pr.returncode == 0, pr.stderr[-3000:]
    out = json.loads(pr.stdout.strip().splitlines()[-1])
    assert out["loss"] == out["ref"] and out["err"] == 0.0
    if mode == "nccl":
        assert out["backend"] == "nccl"
    else:
        seq = [float(i) for i in range(12)]
        assert out["gather"] == seq and out["scatter"] == seq and out["reduce"] == seq
