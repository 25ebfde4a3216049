#!/usr/bin/env python3
"""ELBO-evaluation benchmark of the SVGP/WSVGP hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]        (N > 1: starts its own N ranks, see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Typed without a launcher, `--gpus N` (N > 1) re-runs this file under torch.distributed.run in a child
process before anything touches the GPU and hands back rank 0's JSON line and the exit code.  Backend:
nccl (= RCCL over xGMI); when fewer than N GPUs are visible the ranks share the devices and the
rendezvous / scalar all-reduce fall back to gloo (a rehearsal -- the JSON says so in `backend` and
`devices`, and its `value` is not a scaling number).

One "step" = one closed-form Gaussian ELBO evaluation.  N = 1: the Slide-seq-shaped synthetic
workload of BASELINE.json configs[2] -- N=200k spots, M=2048 inducing points, L=32 latent GPs, Matern-3/2,
fp32; `value` = L=32-latent ELBO evaluations per second.  N > 1 (default `--scaling strong`): BASELINE
configs[3] as stated -- the same workload with L=256 latent GPs, block-sharded 128 / 64 / 32 per GPU on 2 / 4 / 8
GPUs (SURVEY §8d "cfg4", §8e) -- total work fixed, no data-path collective, one RCCL all-reduce of the fp64 ELBO
scalar per step; `value` = L=256-latent ELBO evaluations per second of the whole job (`value_per_32_latents`
restates it in the N = 1 line's unit).  `--scaling weak` keeps 32 latents per GPU (a 32*N-latent model) instead.
Inputs are resident in HBM before the timed region.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK = {"f32": 157.3, "f64": 78.6}  # dense MFMA TFLOP/s (MI355X_MICROARCH.md; f64 = datasheet: the guide has no fp64 row)
HBM_PEAK = 8000.0                     # GB/s, spec (MI355X_MICROARCH.md)


def gemm_source_hash() -> str:
    """sha256 (16 hex digits) of the dominant kernel's sources and of its launcher (svgp.hip fixes the strip width and the
    N-chunking): a PMC traffic file is only quoted for the code it measured."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemmw.hip", "gemmw.h", "cov.h", "gemm.hip", "gemm.h", "common.h", "svgp.hip"):
        h.update(open(os.path.join(ROOT, "gpzoo_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def latest_profile(name: str):
    """profiles/rNN/<name> of the newest round that has it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", name)))
    return files[-1] if files else None


def measured_peaks() -> dict:
    """Denominators measured on the box by tools/peaks.sh, committed per round; {} when absent."""
    f = latest_profile("peaks.json")
    try:
        d = json.load(open(f))
        d["file"] = os.path.relpath(f, ROOT)
        return d
    except Exception:
        return {}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config index (1-based): 2, 3 or 5")
    ap.add_argument("--N", type=int, default=None)
    ap.add_argument("--M", type=int, default=None)
    ap.add_argument("--L", type=int, default=None, help="latents per GPU (weak scaling / one GPU)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default=None,
                    help="N > 1: strong (default for config 3) = BASELINE configs[3], --Ltotal latents block-sharded over "
                         "the GPUs; weak = --L latents on every GPU")
    ap.add_argument("--Ltotal", type=int, default=256, help="strong scaling: latents of the whole model (configs[3]: 256)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generate-kzx", action="store_true",
                    help="stage 1 generates its covariance operand inside the product (GPZ_SVGP_GENERATE_KZX) instead of "
                         "reading a materialised Kzx: the selectable path of DESIGN.md section 5, not the default")
    ap.add_argument("--panel-products", action="store_true",
                    help="fp32, M <= 512: both products in one launch on 64-column panels held in LDS (GPZ_SVGP_PANEL_PRODUCTS, "
                         "csrc/gemmp.hip): the selectable path of DESIGN.md section 5, not the default")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="spots in the CPU-baseline sample (SURVEY §8d: 8192)")
    ap.add_argument("--with-backward", action="store_true",
                    help="(kept for old command lines: the training legs now run by default on one GPU)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the legs outside the timed region: forward + backward of this workload and the "
                         "BASELINE configs[1] record (`small_m`); they never run with more than one rank")
    return ap.parse_args()


# N-sized work of one training step since round 5, in units of L*M^2*N flops (a triangular operand counted once):
# forward Wt = Linv*Kzx and colsum((LuE^T Wt)^2): 2; backward, mu / Lu only: H += Wt diag(gv2) Wt^T (lower tiles): 1
# (rounds 1-4: P-bar and W P-bar^T, 2); all parameters add K-bar_x = [A1 Wt] diag(gv2) + a3 gm^T with the DENSE M x M
# matrix A1 = Linv^T (Sw - I): 2 (rounds 1-4: P-bar, W-bar, K-bar_x and a second accumulation, 5).  The M x M gradient of
# the factor is fp64 algebra on H (DESIGN.md section 5 "Backward").
TRAIN_PRODUCTS = {"mu_Lu": 3, "all_parameters": 5}


def train_leg(ops, spec, g, c, extra, chunk, reps=2) -> dict:
    """ms of one forward + backward pass as the gpzoo modules run a training step (gpzoo_amd/gp.py), per gradient set:
    the factor of Kzz and the q(U) operands the forward prepared are handed to the backward pass of the same call in a
    per-call buffer (never kept across calls), Wt of every chunk stays in HBM between the two.  Last of `reps` runs."""
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    for mode, kg in (("mu_Lu", False), ("all_parameters", True)):
        for _ in range(reps):
            ev0.record()
            handoff = ops.FactorCache()
            with ops.deferred_info():          # as gpzoo.utilities.train* do: `info` is read once, behind the backward's launches
                o = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"],
                                     chunk=chunk, want_Lu=False, retain_wt=1.0 / 3, cache=handoff, **extra)
                gmean = (o["mean"] - g["y"]) / c["noise_sd"] ** 2      # d(-ELBO)/dmean of the Gaussian closed form
                gscale = o["scale"] / c["noise_sd"] ** 2                 # d(-ELBO)/dscale
                ops.svgp_backward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], gmean,
                                  gscale, o["scale"], chunk=chunk, kernel_grads=kg, wt_cache=o.pop("wt_cache", None),
                                  cache=handoff, trust_cache=True, trust_qu=True, **extra)
            ev1.record()
            torch.cuda.synchronize()
            out[mode] = ev0.elapsed_time(ev1)
    return out


def train_roofline(train_ms: dict, L: int, M: int, N: int, dname: str) -> dict:
    prod = L * float(M) * M * N
    r = {}
    for mode, nprod in TRAIN_PRODUCTS.items():
        fl = nprod * prod
        r[mode] = {"bound": "mfma", "products_of_L_M2_N_flops": nprod, "algorithmic_flops": fl, "ms": train_ms[mode],
                   "achieved_TFLOPs": fl / (train_ms[mode] * 1e-3) / 1e12, "peak_TFLOPs": PEAK[dname],
                   "frac": fl / (train_ms[mode] * 1e-3) / 1e12 / PEAK[dname],
                   "mfma_floor_ms": fl / (PEAK[dname] * 1e12) * 1e3}
    return r


def small_m_record(ops, dev) -> dict:
    """BASELINE configs[1] (N=50 000, M=512, L=8, RBF, fp32) next to the headline line: ms per ELBO evaluation, of its
    factorisation (Cholesky + inverse, fp64), and of a forward + backward pass -- the regime the reference's notebooks
    train in.  Outside the timed region; about a second."""
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.synthetic import make_config
    c = make_config(2)
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, dev)
    N, M, L = c["X"].shape[0], c["Z"].shape[0], c["mu"].shape[0]

    def ev():
        return ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                                noise_sd=c["noise_sd"], want_Lu=False, **extra)
    for _ in range(5):
        ev()
    torch.cuda.synchronize()
    ops.profile_enable(True)
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        o = ev()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    prof = ops.profile_read()
    ops.profile_enable(False)
    tr = train_leg(ops, spec, g, c, extra, 0, reps=3)
    return {"workload": "BASELINE configs[1]: 2-D synthetic spatial, N=%d, M=%d, L=%d, %s, f32" % (N, M, L, c["kind"]),
            "evaluation_ms": ms, "evals_per_s": 1e3 / ms, "elbo": float(o["elbo"]),
            "evaluation_roofline": {"bound": "mfma", "algorithmic_flops": 2.0 * L * M * M * N,
                                    "achieved_TFLOPs": 2.0 * L * M * M * N / (ms * 1e-3) / 1e12, "peak_TFLOPs": PEAK["f32"],
                                    "frac": 2.0 * L * M * M * N / (ms * 1e-3) / 1e12 / PEAK["f32"]},
            "factor_ms": prof["potrf_all"][0] / steps, "products_ms": prof["stage1"][0] / steps,
            "forward_backward_ms": tr, "forward_backward_roofline": train_roofline(tr, L, M, N, "f32")}


def usable_cpus() -> tuple:
    """(threads to use, CPU model string): the cores this process may run on -- its affinity mask, cut down to the
    cgroup's CPU quota (cpu.max) and to the PHYSICAL cores among them (one thread per core: torch's CPU kernels gain
    nothing from SMT siblings and lose to oversubscription)."""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    n = len(aff)
    model, phys, cur = "unknown CPU", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur.get("processor") in aff:
                    phys.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
                cur = {}
                continue
            k, v = (t.strip() for t in line.split(":", 1))
            if k == "model name":
                model = v
            if k in ("processor", "physical id", "core id"):
                cur[k] = int(v)
        if phys:
            n = min(n, len(phys))
    except OSError:
        pass
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(f).read().split()
            quota = float(t[0]) if t[0] != "max" else -1.0
            period = float(t[1]) if len(t) > 1 else float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = max(1, min(n, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n), model


def cpu_baseline(cfg_id: int, c: dict, sample: int) -> dict:
    """Time the CPU oracle (a torch-CPU port of the reference's op sequence) on a
    bounded sample: all latents and inducing points, the first `sample` spots --
    the reference's own minibatch usage (utilities.py:605-609)."""
    from oracle import svgp_oracle as O
    n = min(sample, c["X"].shape[0])
    threads, cpu_model = usable_cpus()
    torch.set_num_threads(threads)
    kw = {}
    if "gX" in c:
        kw = dict(gX=c["gX"][:n], gZ=c["gZ"], embedding=O.embed_group_distances(
            torch.ones(c["n_groups"], c["n_groups"]) - torch.eye(c["n_groups"])).to(c["X"].dtype),
            group_diff=c["group_diff"])
    t0 = time.perf_counter()
    Kzz = O.kernel_matrix(c["kind"], c["Z"], c["Z"], c["sigma"], c["lengthscale"], gA=kw.get("gZ"), gB=kw.get("gZ"),
                          embedding=kw.get("embedding"), group_diff=kw.get("group_diff")).contiguous()
    O.add_jitter_(Kzz, c["jitter"])
    torch.linalg.cholesky(Kzz)
    t_chol = time.perf_counter() - t0
    # a short pilot decides whether the planned sample (8192 spots) fits the bench's time budget on this
    # host; if not, the sample is halved until it does and the JSON says which size ran
    pilot_n = min(512, n)
    pk = dict(kw, gX=kw["gX"][:pilot_n]) if "gX" in kw else kw
    t0 = time.perf_counter()
    O.elbo_eval(c["kind"], c["whitened"], c["X"][:pilot_n], c["y"][..., :pilot_n], c["Z"], c["sigma"], c["lengthscale"],
                c["mu"], c["Lu_raw"], c["jitter"], c["noise_sd"], **pk)
    per_spot = max(time.perf_counter() - t0 - t_chol, 1e-9) / pilot_n
    planned = n
    while n > 1024 and t_chol + per_spot * n > 60.0:
        n //= 2
    if "gX" in kw:
        kw["gX"] = kw["gX"][:n]
    t0 = time.perf_counter()
    e, _, _ = O.elbo_eval(c["kind"], c["whitened"], c["X"][:n], c["y"][..., :n], c["Z"], c["sigma"], c["lengthscale"],
                          c["mu"], c["Lu_raw"], c["jitter"], c["noise_sd"], **kw)
    t_eval = time.perf_counter() - t0
    N = c["X"].shape[0]
    t_chunk = max(t_eval - t_chol, 1e-9)
    full = t_chol + math.ceil(N / n) * t_chunk
    return {"value": 1.0 / full, "unit": "ELBO evals/s", "cores": threads, "cpu_model": cpu_model, "kind": "port",
            "sample": f"oracle/svgp_oracle.py (torch CPU, {threads} threads on {cpu_model}) on the first {n} of {N} spots, all "
                      f"{c['mu'].shape[0] if c['mu'].dim() > 1 else 1} latents, M={c['Z'].shape[0]}: {t_eval:.2f} s per "
                      f"chunk-eval incl. {t_chol:.2f} s Kzz+Cholesky; extrapolated to full N as "
                      f"1/(t_chol + ceil(N/n) * t_chunk)" +
                      (f"; {planned} spots were planned, a {pilot_n}-spot pilot predicted more than 60 s for them" if n != planned else ""),
            "sample_spots": n, "sample_elbo": float(e)}


class ClockSampler:
    """Engine clock of the GPU while the timed region runs, from the amdgpu sysfs file rocm-smi itself reads
    (pp_dpm_sclk: the level marked '*').  Host-side file reads on a side thread, 4 per second: no GPU call, no extra
    process.  The pool's boxes settle at different clocks under this load (power cap), which moves every MFMA-bound
    number by the same factor; the sampled clock lets a reader separate the box from the kernel."""

    NOMINAL_MHZ = 2400.0      # the clock the MFMA peaks of MI355X_MICROARCH.md are quoted at

    def __init__(self, device_index: int):
        import glob
        import threading
        self.path, self.samples, self._stop = None, [], threading.Event()
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        want = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        for c in cards:
            if want and want in os.path.realpath(os.path.dirname(c)):
                self.path = c
        if self.path is None and len(cards) == 1:
            self.path = cards[0]
        self._thread = threading.Thread(target=self._run, daemon=True) if self.path else None

    def _read(self):
        try:
            for line in open(self.path):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._read()
            if v:
                self.samples.append(v)
            self._stop.wait(0.25)

    def start(self):
        if self._thread:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self._thread:
            self._thread.join(timeout=2)
        if not self.samples:
            return None
        return {"sclk_MHz_mean": sum(self.samples) / len(self.samples), "sclk_MHz_min": min(self.samples),
                "sclk_MHz_max": max(self.samples), "samples": len(self.samples), "nominal_MHz": self.NOMINAL_MHZ,
                "source": self.path}


def trailing_flops(nblk: int) -> float:
    """Flops of the timed K=256 trailing SYRK launches of one matrix (csrc/factor.hip): after the two
    block columns (k, k+1) the lower tiles of the remaining (nblk-k-2)^2 blocks get a 128x128x256 update."""
    f = 0.0
    for k in range(0, nblk - 2, 2):
        rem = nblk - k - 2
        f += rem * (rem + 1) / 2 * 128.0 * 128.0 * 256.0 * 2.0
    return f


def pmc_traffic(cfg_id, N, M, L, chunk):
    """HBM-side bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC passes
    (profiles/rNN/traffic_stage1.json, FETCH_SIZE doubled per the gfx950 note) -- quoted only when this run is the
    workload those passes measured AND the kernel sources are the ones they measured (sha in the file); null
    otherwise.  bench.py itself cannot read PMC counters."""
    f = latest_profile("traffic_stage1.json")
    try:
        t = json.load(open(f))
        w = t["workload"]
        if (w["config"], w["N"], w["M"], w["L"], w["chunk"]) != (cfg_id, N, M, L, chunk):
            return None, "no PMC passes for this workload"
        if t.get("gemm_src_sha16") != gemm_source_hash():
            return None, "%s measured other kernel sources (sha %s, built %s): re-run tools/profile_round.sh" % (
                os.path.relpath(f, ROOT), t.get("gemm_src_sha16"), gemm_source_hash())
        return t["hbm_bytes_per_launch"], os.path.relpath(f, ROOT)
    except Exception:
        return None, "no committed PMC traffic file"


def plan_latents(cfg_id: int, world: int, rank: int, L, Ltotal: int, scaling):
    """Which latents of which model a rank evaluates: (model config id, scaling, latents of the whole model, this rank's
    contiguous block, latents per rank for every rank).  One GPU: configs[cfg_id - 1] as given.  Several GPUs, config 3:
    strong scaling by default -- BASELINE configs[3], `Ltotal` (256) latents block-sharded (SURVEY §8d "cfg4", §8e);
    `weak`: every rank owns `L` (default: the config's) latents of an L * world-latent model."""
    from gpzoo_amd.synthetic import CONFIGS, shard_latents
    if scaling is None:
        scaling = "strong" if (world > 1 and cfg_id == 3 and L is None) else "weak"
    model = 4 if (cfg_id == 3 and world > 1) else cfg_id
    if world > 1 and scaling == "strong":
        if L is not None:
            raise SystemExit("--L is latents per GPU (weak scaling); with --scaling strong give --Ltotal")
        if Ltotal < world:
            raise SystemExit(f"--Ltotal {Ltotal} < {world} ranks: nothing to shard")
        blocks = [shard_latents(Ltotal, world, r) for r in range(world)]
        return model, "strong", Ltotal, blocks[rank], [len(b) for b in blocks]
    Lper = L if L is not None else CONFIGS[cfg_id]["L"]
    return model, "weak", Lper * world, range(rank * Lper, (rank + 1) * Lper), [Lper] * world


def self_launch(a) -> int:
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as a child torch.distributed.run
    job (this process has not touched the GPU: device_count() does not initialise it) and return its exit
    code; rank 0 of the child prints the JSON line on the shared stdout."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    backend = os.environ.get("GPZ_DIST_BACKEND") or ("nccl" if ndev >= world else "gloo")   # nccl == RCCL over xGMI
    local = local % max(ndev, 1)          # rehearsals may run several ranks on one GPU (gloo only)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ranks = 1
    # GPZ_BENCH_GROUP=1: create the process group and run every collective branch with ONE rank too (under the driver's
    # launcher with --nproc-per-node 1): the RCCL code path of this file executed end to end on a one-GPU box
    grouped = world > 1 or (os.environ.get("GPZ_BENCH_GROUP") == "1" and "WORLD_SIZE" in os.environ)
    if grouped:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        ranks = dist.get_world_size()

    from gpzoo_amd import _lib, ops
    from gpzoo_amd.configs import spec_for_config
    from gpzoo_amd.parallel import _allreduce_scalar
    from gpzoo_amd.synthetic import CONFIGS, make_config

    cfg_id = a.config
    model, scaling, Ltot, lat, per_rank = plan_latents(cfg_id, world, rank, a.L, a.Ltotal, a.scaling)
    Lper = len(lat)
    c = make_config(model, N=a.N, M=a.M, L=Ltot, latents=lat)
    dt = c["dtype"]
    dname = "f32" if dt == torch.float32 else "f64"
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, dev)
    N, M = c["X"].shape[0], c["Z"].shape[0]

    path = {}

    def step():
        out = ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"],
                               y=g["y"], noise_sd=c["noise_sd"], chunk=a.chunk, want_Lu=False,
                               materialize_kzx=False if a.generate_kzx else None, panel_products=a.panel_products, **extra)
        path["bits"] = out["path"]
        e = out["elbo"]
        if grouped:
            e = _allreduce_scalar(e)      # RCCL on the device scalar (nccl); through the host for gloo rehearsals
        return e

    def fence():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    ops.profile_enable(True)
    clock = ClockSampler(local) if rank == 0 else None
    if clock:
        clock.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        elbo = step()
    fence()
    dt_s = time.perf_counter() - t0
    clocks = clock.stop() if clock else None
    prof = ops.profile_read()
    ops.profile_enable(False)
    tmax = torch.tensor([dt_s], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if grouped:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    t = float(tmax)
    elbo = float(elbo)

    train_ms = small_m = None
    if world == 1 and not grouped and not a.no_extra_legs:
        train_ms = train_leg(ops, spec, g, c, extra, a.chunk)
        if cfg_id == 3 and (a.N, a.M, a.L) == (None, None, None):
            small_m = small_m_record(ops, dev)

    if rank == 0:
        Mp = (M + 127) // 128 * 128
        ms1, n1 = prof["stage1"]
        ms2, n2 = prof["stage2"]
        # dominant kernel: Wt = Linv * Kzx, one launch per N-chunk -- fp32: gemmw_kernel (128 x 256 tiles, csrc/gemmw.hip),
        # fp64: gemm128_kernel<T,NN,store+colstats>.  algorithmic flops = L * M^2 * N per evaluation (SURVEY §8d TRSM count).
        flops1 = Lper * float(M) * M * N * a.steps
        # the kernel that ran, from the predicate the library itself dispatches on (gpz_svgp_forward_path)
        bits = path.get("bits", 0)
        panel = bool(bits & 4)      # fp32, M <= 512: ONE launch runs both products panel by panel (csrc/gemmp.hip): 2 x the flops
        ach1 = (2.0 if panel else 1.0) * flops1 / (ms1 * 1e-3) / 1e12 if ms1 > 0 else 0.0
        mp = measured_peaks()
        traffic, traffic_src = pmc_traffic(cfg_id, N, M, Lper, a.chunk)
        kname = ("panel_kernel<Mp/64 row blocks> (Wt = Linv*Kzx and colsum((LuE^T Wt)^2) on one 64-column panel in LDS, csrc/gemmp.hip)" if panel
                 else "gemmw_gen_kernel<512,128,generated Kzx,lower,store+colstats> (Wt = Linv*k(Z,X), csrc/gemmw.hip)" if bits & 2
                 else "gemmw_kernel<128,256,mem,lower,store+colstats> (Wt = Linv*Kzx, csrc/gemmw.hip)" if bits & 1
                 else "gemm128_kernel<%s,NN,store+colstats> (Wt = Linv*Kzx, csrc/gemm.hip)" % ("float" if dname == "f32" else "double"))
        roof = {"bound": "mfma",
                "kernel": kname,
                "achieved": ach1, "peak": PEAK[dname], "unit": "TFLOP/s", "frac": ach1 / PEAK[dname],
                "traffic": traffic, "traffic_source": traffic_src, "launches": n1,
                "avg_launch_ms": ms1 / max(n1, 1)}
        if clocks:     # the same fraction against the peak at the clock this box actually ran the timed region at
            roof["peak_at_sampled_clock"] = PEAK[dname] * clocks["sclk_MHz_mean"] / ClockSampler.NOMINAL_MHZ
            roof["frac_at_sampled_clock"] = ach1 / roof["peak_at_sampled_clock"]
        mpk = mp.get("mfma_%s_TFLOPs" % dname)
        if mpk:
            roof["measured_peak"], roof["frac_of_measured_peak"] = mpk, ach1 / mpk
        kms, kn = prof["kfill"]
        esz = 4 if dname == "f32" else 8
        kbytes = (Lper * float(Mp) * N * esz) * a.steps
        kgbs = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        tr_tf = (trailing_flops(Mp // 128) * Lper * a.steps / (prof["potrf_trailing"][0] * 1e-3) / 1e12
                 if prof["potrf_trailing"][0] > 0 else 0.0)
        sub = {
            "kuf_fill": ({"bound": "hbm", "achieved_GBps": kgbs, "peak_GBps": HBM_PEAK, "frac": kgbs / HBM_PEAK,
                          "ms_per_eval": kms / a.steps} if kn > 0 else
                         {"bound": "hbm", "achieved_GBps": None, "peak_GBps": HBM_PEAK, "frac": None, "ms_per_eval": 0.0,
                          "note": "no fill launch: panel_kernel computes its covariance panel itself and Kzx is never written"}),
            # (panel kernel: the second product runs inside the launch counted as stage 1 -- its rate is that launch's, over both)
            "stage2_LuT_Wt": ({"bound": "mfma", "achieved_TFLOPs": ach1, "peak_TFLOPs": PEAK[dname], "ms_per_eval": 0.0,
                               "note": "inside panel_kernel with the first product: achieved_TFLOPs is the launch's rate over both"}
                              if panel else
                              {"bound": "mfma", "achieved_TFLOPs": flops1 / (ms2 * 1e-3) / 1e12 if ms2 > 0 else 0.0,
                               "peak_TFLOPs": PEAK[dname], "ms_per_eval": ms2 / a.steps}),
            "stage1_ms_per_eval": ms1 / a.steps,
            "potrf_ms_per_eval": prof["potrf_all"][0] / a.steps,
            "potrf_trailing": {"bound": "mfma", "dtype": "f64", "achieved_TFLOPs": tr_tf, "peak_TFLOPs": PEAK["f64"],
                               "frac": tr_tf / PEAK["f64"],
                               "ms_per_eval": prof["potrf_trailing"][0] / a.steps,
                               "note": "K=256 SYRK launches only (90 % of the M^3/3 factorisation flops at M=2048)"},
            "trtri_ms_per_eval": prof["trtri"][0] / a.steps,
            "finalize_ms_per_eval": prof["finalize"][0] / a.steps,
        }
        potrf_flops = Lper * float(Mp) ** 3 / 3.0
        coop = bool(_lib.load().gpz_factor_path(M, 1))
        if coop:
            # one launch does the Cholesky AND the triangular inverse (csrc/coop.hip): the two profile slots bracket the
            # same interval, and there is no separate trailing-update launch to time
            fms = prof["potrf_all"][0] / a.steps
            ftf = 2.0 * potrf_flops / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
            sub["factor_path"] = "one launch: coop_factor_kernel (Cholesky + triangular inverse, tile dataflow)"
            sub["factor"] = {"bound": "mfma", "dtype": "f64", "ms_per_eval": fms,
                             "algorithmic_flops": 2.0 * potrf_flops, "achieved_TFLOPs": ftf, "peak_TFLOPs": PEAK["f64"],
                             "frac": ftf / PEAK["f64"],
                             "note": "L*M^3/3 (potrf) + L*M^3/3 (trtri) over the launch; potrf_ms_per_eval and "
                                     "trtri_ms_per_eval both report this interval"}
            del sub["potrf_trailing"]
        else:
            sub["factor_path"] = "launch per step (GPZ_FACTOR_PATH=launches or an order beyond the task list)"
            if prof["potrf_all"][0] > 0:
                sub["potrf_whole_TFLOPs"] = potrf_flops * a.steps / (prof["potrf_all"][0] * 1e-3) / 1e12
        if mp:   # fractions of the rates measured on the box (tools/peaks.sh), next to the spec / datasheet ones
            sub["measured_peaks"] = {k: v for k, v in mp.items() if k != "how"}
            if mp.get("hbm_write_GBps") and kn > 0:
                sub["kuf_fill"]["frac_of_measured_write_rate"] = kgbs / mp["hbm_write_GBps"]
            if mp.get("mfma_f64_TFLOPs") and "potrf_trailing" in sub:
                sub["potrf_trailing"]["frac_of_measured_peak"] = tr_tf / mp["mfma_f64_TFLOPs"]
        # one step = one ELBO evaluation of the WHOLE model (all ranks together): strong scaling counts evaluations of the
        # Ltot-latent model, weak scaling evaluations of the per-GPU block (Lper latents) summed over ranks
        strong = world > 1 and scaling == "strong"
        value = a.steps / t if strong else a.steps * world / t
        unit_L = Ltot if strong else Lper
        shard = ("L=%d latents block-sharded as %s per GPU on %d GPUs" % (Ltot, "/".join(map(str, per_rank)), world)
                 if strong else "L=%d latents/GPU x %d GPU(s)" % (Lper, world))
        res = {
            "metric": "ELBO evals/sec (L=%d-latent evaluations, all GPUs) at M=%d inducing, N=%d, %s"
                      % (unit_L, M, N, shard),
            "value": value, "unit": "ELBO evals/s", "n_gpus": world, "ranks": ranks,
            "backend": backend if grouped else None, "devices": ndev, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * t / a.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": dname, "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %s, N=%d spots, M=%d, %s, %s, %s"
                                   % (model - 1, {2: "2-D synthetic spatial", 5: "MGGP multi-group synthetic (4 groups)"}.get(
                                       cfg_id, "Slide-seq-shaped synthetic"), N, M, shard, c["kind"], dname),
                       "latents_total": Ltot, "latents_per_rank": per_rank,
                       "whitened": bool(c["whitened"]), "chunk": a.chunk, "factor_dtype": "f64"},
            # the same throughput in the one-GPU line's unit (L=32-latent evaluations per second)
            "value_per_32_latents": value * unit_L / 32.0,
            "elbo": elbo, "roofline": roof, "kernels": sub, "clocks": clocks,
        }
        if train_ms is not None:
            res["forward_backward_ms"] = train_ms
            res["forward_backward_roofline"] = train_roofline(train_ms, Lper, M, N, dname)
        if small_m is not None:
            res["small_m"] = small_m
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(cfg_id, c, a.cpu_sample)
            # parity on the driver-run line: the HIP path on the very slice the CPU port just evaluated
            # (outside the timed region), same dtype, all latents and inducing points
            n = cb["sample_spots"]
            ex = {k: (v[:n] if k == "gX" else v) for k, v in extra.items()}
            o = ops.svgp_forward(spec, g["X"][:n], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"],
                                 y=g["y"][..., :n].contiguous(), noise_sd=c["noise_sd"], want_Lu=False, **ex)
            cb["sample_elbo_hip"] = float(o["elbo"])
            cb["sample_rel_diff"] = abs(cb["sample_elbo_hip"] - cb["sample_elbo"]) / abs(cb["sample_elbo"])
            # ... and, where the build container ran the REFERENCE itself on this very slice (a committed fixture: data,
            # tests/golden/make_baseline_golden.py), its fp64 ELBO next to ours
            fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "baseline_cfg3_slice.npz")
            if cfg_id == 3 and (N, M, Lper, n) == (200_000, 2048, 32, 8192) and os.path.exists(fx):
                import numpy as np
                ref = float(np.load(fx, allow_pickle=False)["f64_elbo"])
                cb["sample_elbo_reference_fp64"] = ref
                cb["sample_rel_diff_vs_reference"] = abs(cb["sample_elbo_hip"] - ref) / abs(ref)
            res["cpu_baseline"] = cb
        elif not a.no_cpu_baseline:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
