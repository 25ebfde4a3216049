"""GPU: SURVEY §8e on one box -- RCCL initialised and the device all-reduce executed (world size 1, backend
nccl), BASELINE configs[3] (L=256 sharded 8 x 32) and configs[4] (MGGP fp64, 8 x 4 latents) at their full
sizes through additivity over the shards the ranks would own, and ``bench.py --gpus 2`` typed without a
launcher (two ranks on the one GPU, gloo rendezvous)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RCCL_CHILD = r"""
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
from gpzoo_amd.parallel import sharded_elbo, hip_local_elbo, _allreduce_scalar
from gpzoo_amd.synthetic import make_config
c = make_config(2, N=3000, M=200, L=5, dtype=torch.float64)
g = {{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}}
e = sharded_elbo(g, 5)                      # latent block of rank 0 of 1 + RCCL all-reduce of the device scalar
ref = hip_local_elbo(g)
t = torch.arange(4, dtype=torch.float64, device=dev)
r = _allreduce_scalar(t)
torch.cuda.synchronize()
print(json.dumps(dict(backend=dist.get_backend(), world=dist.get_world_size(), elbo=float(e), ref=float(ref),
                      is_cuda=bool(e.is_cuda), vec=r.tolist())))
dist.destroy_process_group()
"""


def test_rccl_world_size_one_allreduce_on_device():
    """RCCL communicator creation + an all-reduce launched on the device scalar: the branch of
    ``parallel._allreduce_scalar`` the 8-GPU job takes, executed on the box (one rank)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    code = _RCCL_CHILD.format(root=ROOT, port=_free_port())
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=420, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["backend"] == "nccl" and out["world"] == 1 and out["is_cuda"]
    assert out["elbo"] == out["ref"]                      # sum over one rank: bitwise the local value
    assert out["vec"] == [0.0, 1.0, 2.0, 3.0]


_ABI_CHILD = r"""
import sys, json, torch
sys.path.insert(0, {root!r})
from gpzoo_amd.parallel import AbiCommunicator, sharded_elbo, hip_local_elbo
from gpzoo_amd.synthetic import make_config
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
comm = AbiCommunicator(dev)                   # gpz_comm_unique_id + gpz_comm_init (one rank), no torch.distributed
t = torch.arange(5, dtype=torch.float64, device=dev) * 1.5
comm.allreduce_sum_(t)
c = make_config(2, N=3000, M=200, L=5, dtype=torch.float64)
g = {{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}}
e = sharded_elbo(g, 5, comm=comm)
ref = hip_local_elbo(g)
torch.cuda.synchronize()
comm.close()
print(json.dumps(dict(vec=t.tolist(), elbo=float(e), ref=float(ref))))
"""


def test_c_abi_collective_one_rank():
    """gpz_comm_unique_id / gpz_comm_init / gpz_allreduce_sum_f64 / gpz_comm_destroy (RCCL bound by dlopen):
    a one-rank communicator created and used from the C ABI alone, then driving parallel.sharded_elbo."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _ABI_CHILD.format(root=ROOT)], capture_output=True, text=True, timeout=420, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["vec"] == [0.0, 1.5, 3.0, 4.5, 6.0]
    assert out["elbo"] == out["ref"]


def _to_dev(c):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in c.items()}


def _eval(g, **kw):
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    spec, extra = spec_for_config(g)
    return ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], g["jitter"], g["whitened"], y=g["y"],
                            noise_sd=g["noise_sd"], want_Lu=False, **extra, **kw)


def _sharded_vs_single(cfg, L, world, per_latent):
    """Evaluate the L-latent model once, then as `world` contiguous latent blocks (what each rank of the
    sharded job evaluates, drawn per block exactly as bench.py / parallel.shard_problem do): the blocks'
    ELBOs add up to the single call's, and their q(F) means are bitwise the single call's rows."""
    from gpzoo_amd.synthetic import make_config, shard_latents
    shards = []
    for r in range(world):
        lat = shard_latents(L, world, r)
        shards.append(_to_dev(make_config(cfg, L=L, latents=lat)))
    full = dict(shards[0])
    for k in per_latent:
        full[k] = torch.cat([s[k] for s in shards], dim=0)
    a = _eval(full)
    tot, kls = 0.0, []
    for r, s in enumerate(shards):
        lat = shard_latents(L, world, r)
        o = _eval(s)
        tot += float(o["elbo"])
        kls.append(o["kl"])
        assert torch.equal(o["mean"], a["mean"][lat.start:lat.stop])
        assert torch.equal(o["scale"], a["scale"][lat.start:lat.stop])
        del o
    assert tot == pytest.approx(float(a["elbo"]), rel=1e-12)
    assert torch.equal(torch.cat(kls), a["kl"])
    assert torch.isfinite(a["mean"]).all() and (a["scale"] > 0).all()
    return a


def test_config4_256_latents_as_eight_shards_full_size():
    """BASELINE configs[3]: N=200k, M=2048, Matern-3/2 fp32, L=256 -> 8 ranks x 32 latents."""
    a = _sharded_vs_single(4, 256, 8, ("sigma", "lengthscale", "mu", "Lu_raw", "y"))
    assert a["mean"].shape == (256, 200_000)


def test_config5_mggp_fp64_as_eight_shards_full_size():
    """BASELINE configs[4]: 4 groups x 50k spots, shared M=2048, MGGP_NSF_RBF fp64, L=32 -> 8 ranks x 4 latents."""
    a = _sharded_vs_single(5, 32, 8, ("sigma", "lengthscale", "mu", "Lu_raw", "y", "group_diff"))
    assert a["mean"].dtype == torch.float64 and a["mean"].shape == (32, 200_000)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_two_ranks_is_baseline_configs3_strong_scaled(world):
    """`python bench.py --gpus N` defaults to BASELINE configs[3]: the L=256 model, 256 / N latents per rank (128 on two
    GPUs, 64 on four), strong scaling (a gloo rehearsal on this one-GPU box); the all-reduced ELBO is the single-process
    L=256 ELBO on the same spots."""
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    env = dict(os.environ, GPZ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--N", "8192", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == world and res["ranks"] == world and res["scaling"] == "strong"
    assert res["config"]["latents_total"] == 256 and res["config"]["latents_per_rank"] == [256 // world] * world
    assert "configs[3]" in res["config"]["workload"] and "L=256" in res["config"]["workload"]
    assert res["value"] == pytest.approx(1e3 / res["ms_per_step"], rel=1e-9)      # L=256-latent evaluations per second
    assert res["value_per_32_latents"] == pytest.approx(8 * res["value"], rel=1e-12)
    g = _to_dev(make_config(4, N=8192))
    assert g["mu"].shape[0] == 256
    spec, extra = spec_for_config(g, torch.device("cuda", 0))
    o = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], g["jitter"], g["whitened"], y=g["y"],
                         noise_sd=g["noise_sd"], want_Lu=False, want_moments=False, **extra)
    assert res["elbo"] == pytest.approx(float(o["elbo"]), rel=1e-12)


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` as typed (no launcher): it starts its own two ranks -- on this one-GPU box they
    share the device and rendezvous over gloo -- and prints one JSON line with n_gpus = ranks = 2."""
    env = dict(os.environ, GPZ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--N", "20000", "--M", "1024",
                        "--L", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],      # --L: latents per GPU = weak
                       capture_output=True, text=True, timeout=540, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks"] == 2 and res["backend"] == "gloo" and res["scaling"] == "weak"
    assert res["value"] > 0 and res["unit"] == "ELBO evals/s"


def test_bench_under_the_drivers_launcher_with_rccl_at_one_rank():
    """The driver's own command line -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --
    with N = 1 and GPZ_BENCH_GROUP=1: bench.main creates the nccl (= RCCL) process group bound to the device, all-reduces the
    device ELBO scalar every step, fences with dist.barrier, takes the MAX of the step time on a DEVICE tensor and tears the
    group down -- every multi-rank branch of the file, executed end to end on this one-GPU box.  (An 8-rank rehearsal is not
    possible here: the pool admits at most 6 processes on a GPU; the 8 x 32 plan itself is pinned by
    test_bench_launcher.py::test_latent_plan_is_baseline_configs3_for_several_gpus, the eight shards at full size by
    test_config4_256_latents_as_eight_shards_full_size, and 2- and 4-rank rehearsals run above.)"""
    env = dict(os.environ, GPZ_DIST_BACKEND="nccl", GPZ_BENCH_GROUP="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--N", "20000", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["ranks"] == 1 and res["backend"] == "nccl"
    assert res["value"] == pytest.approx(1e3 / res["ms_per_step"], rel=1e-9) and res["roofline"]["frac"] > 0.3
    # the all-reduced ELBO of one rank is the local one
    from gpzoo_amd import ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    g = _to_dev(make_config(3, N=20000))
    spec, extra = spec_for_config(g, torch.device("cuda", 0))
    o = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], g["jitter"], g["whitened"], y=g["y"],
                         noise_sd=g["noise_sd"], want_Lu=False, want_moments=False, **extra)
    assert res["elbo"] == pytest.approx(float(o["elbo"]), rel=1e-12)
