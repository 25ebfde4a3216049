import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    names = sorted(f[:-4] for f in os.listdir(GOLDEN)
                   if f.endswith(".npz") and f not in ("kernels_only.npz", "kernel_grads.npz")
                   and not f.startswith(("poisson_", "vnngp_", "ref_checkpoint_", "ref_trajectory_", "extra_", "exact_", "multiblock", "baseline_")))
    return names


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
