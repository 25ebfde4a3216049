"""CPU: `python bench.py --gpus N` typed without a launcher re-runs itself under torch.distributed.run as a CHILD
process (one rank per GPU, 127.0.0.1 rendezvous) before anything touches the GPU; the traffic file guard and the
measured-peaks loader are pure host logic too."""
import json
import os
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_self_launch_builds_the_driver_command(monkeypatch):
    import subprocess
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    a = bench.parse()
    assert bench.self_launch(a) == 7                      # the child's exit code comes back
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_latent_plan_is_baseline_configs3_for_several_gpus():
    """N > 1 defaults to BASELINE configs[3] as stated (SURVEY §8d "cfg4"): 256 latents block-sharded 128 / 64 / 32 per
    GPU, strong scaling; one GPU stays configs[2]; `--scaling weak` keeps 32 latents on every GPU."""
    import pytest
    import bench
    assert bench.plan_latents(3, 1, 0, None, 256, None) == (3, "weak", 32, range(0, 32), [32])
    for world, per in ((2, 128), (4, 64), (8, 32)):
        seen = []
        for rank in range(world):
            model, scaling, total, lat, per_rank = bench.plan_latents(3, world, rank, None, 256, None)
            assert (model, scaling, total, per_rank) == (4, "strong", 256, [per] * world)
            seen += list(lat)
        assert seen == list(range(256))                    # contiguous blocks, every latent exactly once
    assert bench.plan_latents(3, 3, 0, None, 256, None)[4] == [86, 85, 85]
    assert bench.plan_latents(3, 4, 1, None, 256, "weak") == (4, "weak", 128, range(32, 64), [32] * 4)
    assert bench.plan_latents(3, 2, 1, 4, 256, None) == (4, "weak", 8, range(4, 8), [4, 4])   # --L: latents per GPU
    assert bench.plan_latents(5, 8, 2, None, 256, None)[:3] == (5, "weak", 256)
    with pytest.raises(SystemExit):
        bench.plan_latents(3, 2, 0, 4, 256, "strong")


def test_traffic_file_is_only_quoted_for_the_kernel_it_measured(tmp_path, monkeypatch):
    import bench
    d = tmp_path / "profiles" / "r09"
    d.mkdir(parents=True)
    rec = {"workload": {"config": 3, "N": 10, "M": 4, "L": 2, "chunk": 0}, "hbm_bytes_per_launch": 123.0,
           "gemm_src_sha16": bench.gemm_source_hash()}
    (d / "traffic_stage1.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "latest_profile", lambda name: str(d / name))
    assert bench.pmc_traffic(3, 10, 4, 2, 0)[0] == 123.0
    assert bench.pmc_traffic(3, 11, 4, 2, 0)[0] is None              # another workload
    rec["gemm_src_sha16"] = "0" * 16
    (d / "traffic_stage1.json").write_text(json.dumps(rec))
    val, why = bench.pmc_traffic(3, 10, 4, 2, 0)
    assert val is None and "other kernel sources" in why              # a stale measurement is refused


def test_committed_profiles_are_consistent_with_the_sources():
    """Whatever profiles/rNN/traffic_stage1.json is newest either matches the built GEMM sources or is refused."""
    import bench
    f = bench.latest_profile("traffic_stage1.json")
    if f is None:
        return
    t = json.load(open(f))
    val, _ = bench.pmc_traffic(t["workload"]["config"], t["workload"]["N"], t["workload"]["M"], t["workload"]["L"],
                               t["workload"]["chunk"])
    assert (val is not None) == (t.get("gemm_src_sha16") == bench.gemm_source_hash())


def test_clock_sampler_reads_the_active_sclk_level(tmp_path):
    """bench.py's `clocks` object: the level rocm-smi marks with '*' in pp_dpm_sclk; no file -> no object."""
    import time
    import bench
    f = tmp_path / "pp_dpm_sclk"
    f.write_text("0: 132Mhz \n1: 2383Mhz *\n")
    s = bench.ClockSampler.__new__(bench.ClockSampler)
    import threading
    s.path, s.samples, s._stop = str(f), [], threading.Event()
    s._thread = threading.Thread(target=s._run, daemon=True)
    assert s._read() == 2383.0
    s.start()
    time.sleep(1.0)
    out = s.stop()
    assert out["samples"] >= 2 and out["sclk_MHz_mean"] == 2383.0 and out["nominal_MHz"] == 2400.0
    none = bench.ClockSampler.__new__(bench.ClockSampler)
    none.path, none.samples, none._stop, none._thread = None, [], threading.Event(), None
    none.start()
    assert none.stop() is None
