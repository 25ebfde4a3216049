"""GPU: the C ABI without Python in the loop -- examples/c_abi_demo.c (plain C, hipMalloc'ed buffers, no
torch) is compiled against include/gpzoo_hip.h + libgpzoo_hip.so and run as a child process; its inputs are
regenerated here from the same linear-congruential stream and pushed through the Python mirror and the CPU
oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(n, state=[0x2545F4914F6CDD1D]):
    out = np.empty(n)
    s = state[0]
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        out[i] = (s >> 11) / 9007199254740992.0
    state[0] = s
    return out


def test_plain_c_program_matches_python_mirror_and_oracle(tmp_path):
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path / "c_abi_demo")
    lib = os.path.join(ROOT, "gpzoo_amd")
    subprocess.run([gcc, "-std=c11", "-O2", os.path.join(ROOT, "examples", "c_abi_demo.c"), "-D__HIP_PLATFORM_AMD__",
                    "-I" + rocm + "/include", "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lgpzoo_hip",
                    "-L" + rocm + "/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath," + rocm + "/lib", "-o", exe],
                   check=True, capture_output=True, timeout=300)
    run = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300)
    got = dict(line.split(" ", 1) for line in run.stdout.strip().splitlines())
    assert int(got["version"]) >= 100 and int(got["bad_rc"]) < 0 and len(got["bad_msg"]) > 0

    N, M, L, d = 1500, 200, 3, 2
    X = torch.from_numpy(20 * _stream(N * d) - 10).reshape(N, d)
    Z = torch.from_numpy(20 * _stream(M * d) - 10).reshape(M, d)
    mu = torch.from_numpy(_stream(L * M) - 0.5).reshape(L, M)
    Lu = torch.from_numpy(0.1 * (_stream(L * M * M) - 0.5)).reshape(L, M, M)
    y = torch.from_numpy(2 * _stream(L * N) - 1).reshape(L, N)
    sigma = torch.tensor([0.8, 1.0, 1.2], dtype=torch.float64)
    ell = torch.tensor([2.0, 3.0, 4.0], dtype=torch.float64)

    from oracle import svgp_oracle as O
    e_ref, mean_ref, scale_ref = O.elbo_eval("matern32", True, X, y, Z, sigma, ell, mu, Lu, 1e-2, 0.5)
    assert float(got["elbo"]) == pytest.approx(float(e_ref), rel=1e-9)
    for i in range(4):
        assert float(got[f"mean{i}"]) == pytest.approx(float(mean_ref.reshape(-1)[i]), rel=1e-8, abs=1e-12)
        assert float(got[f"scale{i}"]) == pytest.approx(float(scale_ref.reshape(-1)[i]), rel=1e-8)
    assert all(int(got[f"info{l}"]) == 0 for l in range(L))

    from gpzoo_amd import _lib, ops
    from gpzoo_amd.ops import KernelSpec
    out = ops.svgp_forward(KernelSpec(_lib.KERNEL_MATERN32, sigma.cuda(), ell.cuda(), True), X.cuda(), Z.cuda(), mu.cuda(),
                           Lu.cuda(), 1e-2, True, y=y.cuda(), noise_sd=0.5, chunk=512)
    assert float(got["elbo"]) == float(out["elbo"])                 # same library, same chunking: bitwise
    for l in range(L):
        assert float(got[f"kl{l}"]) == float(out["kl"][l]) and float(got[f"loglik{l}"]) == float(out["loglik"][l])


def test_plain_c_sharded_client_runs_the_rccl_exchange(tmp_path):
    """examples/c_abi_shard.c as a one-rank job: communicator from gpz_comm_unique_id / gpz_comm_init, partial ELBO
    from gpz_svgp_forward, gpz_allreduce_sum_f64 on the device scalar -- all from plain C."""
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path / "c_abi_shard")
    lib = os.path.join(ROOT, "gpzoo_amd")
    subprocess.run([gcc, "-std=gnu11", "-O2", os.path.join(ROOT, "examples", "c_abi_shard.c"), "-D__HIP_PLATFORM_AMD__",
                    "-I" + rocm + "/include", "-I" + os.path.join(ROOT, "include"), "-L" + lib, "-lgpzoo_hip",
                    "-L" + rocm + "/lib", "-lamdhip64", "-Wl,-rpath," + lib, "-Wl,-rpath," + rocm + "/lib", "-o", exe],
                   check=True, capture_output=True, timeout=300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([exe, "1", "0", str(tmp_path / "gpz.id")], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stderr[-1500:]
    got = dict(line.split(" ", 1) for line in run.stdout.strip().splitlines())
    assert got["world"] == "1" and float(got["total_elbo"]) == float(got["local_elbo"]) and float(got["local_elbo"]) < 0
