"""GPU: batched Cholesky / triangular solve against torch fp64 on the same inputs,
at sizes that exercise several panels, ragged (non multiple of 128) orders and the
non-positive-definite error path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def spd(batch, M, seed, jitter=0.5):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(batch, M, M + 8, generator=g, dtype=torch.float64)
    return A @ A.transpose(-1, -2) / M + jitter * torch.eye(M, dtype=torch.float64)


@pytest.mark.parametrize("M,batch", [(1, 2), (36, 3), (128, 2), (200, 2), (384, 3), (700, 2), (1024, 1), (3000, 2), (4096, 1)])
def test_cholesky_matches_lapack(M, batch):
    from gpzoo_amd import ops
    A = spd(batch, M, 7 + M)
    ref = torch.linalg.cholesky(A)
    got = ops.cholesky(A.cuda()).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-9, atol=1e-11)
    assert torch.equal(got.triu(1), torch.zeros_like(got))  # zeros above the diagonal


@pytest.mark.parametrize("M,N,batch", [(1, 1, 1), (36, 50, 2), (256, 130, 2), (640, 300, 1), (700, 129, 3), (2048, 512, 2)])
def test_trsm_matches_lapack(M, N, batch):
    from gpzoo_amd import ops
    Lc = torch.linalg.cholesky(spd(batch, M, 11 + M))
    B = torch.randn(batch, M, N, generator=torch.Generator().manual_seed(M + N), dtype=torch.float64)
    ref = torch.linalg.solve_triangular(Lc, B, upper=False)
    got = ops.solve_triangular_lower(Lc.cuda(), B.cuda()).cpu()
    torch.testing.assert_close(got, ref, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("M,N,batch", [(200, 70, 2), (513, 300, 1)])
def test_fp32_storage_factor_entries(M, N, batch):
    """dtype = GPZ_F32 on the stand-alone entries: fp32 in / out, fp64 arithmetic inside -- the result is the fp64
    answer of the fp32-representable input rounded once."""
    from gpzoo_amd import ops
    A32 = spd(batch, M, 3 + M).float()
    got = ops.cholesky(A32.cuda())
    assert got.dtype == torch.float32
    ref = torch.linalg.cholesky(A32.double())
    torch.testing.assert_close(got.cpu().double(), ref, rtol=2e-7, atol=2e-7)
    B32 = torch.randn(batch, M, N, generator=torch.Generator().manual_seed(M), dtype=torch.float32)
    L32 = ref.float()
    X = ops.solve_triangular_lower(L32.cuda(), B32.cuda())
    assert X.dtype == torch.float32
    refX = torch.linalg.solve_triangular(L32.double(), B32.double(), upper=False)
    torch.testing.assert_close(X.cpu().double(), refX, rtol=1e-6, atol=1e-6 * float(refX.abs().max()))


def test_trsm_shared_factor_broadcasts_over_the_batch():
    from gpzoo_amd import ops
    Lc = torch.linalg.cholesky(spd(1, 300, 21))[0]
    B = torch.randn(3, 300, 40, generator=torch.Generator().manual_seed(4), dtype=torch.float64)
    got = ops.solve_triangular_lower(Lc.cuda(), B.cuda()).cpu()
    torch.testing.assert_close(got, torch.linalg.solve_triangular(Lc, B, upper=False), rtol=1e-8, atol=1e-10)


def test_not_positive_definite_raises():
    """gp.py:213/270/360 callers see torch.linalg.LinAlgError with the failing minor."""
    from gpzoo_amd import ops
    A = spd(3, 200, 5)
    A[1, 150, 150] = -1.0  # leading minor of order 151 fails in batch element 1
    with pytest.raises(torch.linalg.LinAlgError, match=r"Batch element 1.*minor of order 151"):
        ops.cholesky(A.cuda())
    with pytest.raises(torch.linalg.LinAlgError):
        torch.linalg.cholesky(A)


_GIVE_UP_CHILD = r"""
import ctypes as C, json, sys, time
sys.path.insert(0, {root!r})
import torch
from gpzoo_amd import ops, _lib
lib = _lib.load()
lib.gpz_debug_coop_mute.restype = C.c_int
lib.gpz_debug_coop_mute.argtypes = [C.c_int, C.c_int]
g = torch.Generator().manual_seed(0)
B = torch.randn(2, 512, 512, generator=g, dtype=torch.float64)
A = (B @ B.transpose(-1, -2) / 512 + torch.eye(512, dtype=torch.float64)).cuda()
ref = ops.cholesky(A)
lib.gpz_debug_coop_mute(1, 0)                     # tile (1, 0) of matrix 0 is computed but its flag never stored
t0 = time.perf_counter()
try:
    ops.cholesky(A)
    msg = None
except RuntimeError as e:
    msg = str(e)
dt = time.perf_counter() - t0
lib.gpz_debug_coop_mute(-1, -1)
again = ops.cholesky(A)
print(json.dumps(dict(msg=msg, seconds=dt, healthy=bool(torch.equal(again, ref)))))
"""


def test_one_launch_factorisation_gives_up_instead_of_hanging():
    """The exit condition every wave of the one-launch factorisation reaches: with one tile's flag withheld (a debug hook)
    everything that depends on it polls until the launch's timeout (50 ms here, 5 s by default), the first poller to time
    out raises the abort word, every other poll loop reads it, all workgroups leave and info = -7 becomes a RuntimeError --
    and the next call on the same device is bitwise what it was before."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPZ_COOP_TIMEOUT_MS="50")
    env.pop("GPZ_FACTOR_PATH", None)
    p = subprocess.run([sys.executable, "-c", _GIVE_UP_CHILD.format(root=root)], capture_output=True, text=True, timeout=300,
                       env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["msg"] is not None and "timed out" in out["msg"] and "-7" in out["msg"]
    assert out["seconds"] < 5.0 and out["healthy"]


_TASK_LIST_CHILD = r"""
import ctypes as C, json, sys
sys.path.insert(0, {root!r})
import torch
from gpzoo_amd import _lib
lib = _lib.load()
lib.gpz_debug_factor_sync_words.restype = C.c_size_t
lib.gpz_debug_factor_sync_words.argtypes = [C.c_int64, C.c_int64]
lib.gpz_debug_factor_invert.restype = C.c_int
lib.gpz_debug_factor_invert.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6
out = {{}}
for L, M in ((3, 256), (5, 640), (2, 1408)):
    g = torch.Generator().manual_seed(M)
    B = torch.randn(L, M, M, generator=g, dtype=torch.float64)
    A = (B @ B.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64)).cuda()
    nblk = M // 128
    Dinv = torch.empty(L * nblk * 128 * 128, dtype=torch.float64, device="cuda")
    Linv = torch.full((L, M, M), float("nan"), dtype=torch.float64, device="cuda")
    T = torch.empty(L, M, M, dtype=torch.float64, device="cuda")
    sync = torch.empty(lib.gpz_debug_factor_sync_words(M, L), dtype=torch.int32, device="cuda")
    info = torch.empty(L, dtype=torch.int32, device="cuda")
    W = A.clone()
    rc = lib.gpz_debug_factor_invert(W.data_ptr(), M, L, Dinv.data_ptr(), Linv.data_ptr(), T.data_ptr(), sync.data_ptr(),
                                     info.data_ptr(), None)
    torch.cuda.synchronize()
    Lc = torch.tril(W)
    ref = torch.linalg.cholesky(A)
    eye = torch.eye(M, dtype=torch.float64, device="cuda")
    out[str(M)] = dict(rc=rc, info=int(info.abs().max()), factor=float((Lc - ref).abs().max() / ref.abs().max()),
                      inverse=float((torch.tril(Linv) @ Lc - eye).abs().max()), above=float(torch.triu(Linv, 1).abs().max()))
print(json.dumps(out))
"""


@pytest.mark.parametrize("unfused", [False, True])
def test_both_task_lists_of_the_one_launch_factorisation(unfused):
    """csrc/coop.hip claims tiles (j, j-1) and (j, j) as one task since round 5 (the tile below the diagonal stays in LDS for
    the diagonal tile's last update); GPZ_COOP_UNFUSED selects round 4's list, read once per process -- hence a child
    process per list.  Factor against LAPACK, inverse against the factor, nothing above the diagonal; 2, 5 and 11 block
    columns (an even, an odd count and one where every fused task has parked sums)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("GPZ_FACTOR_PATH", None)
    env.pop("GPZ_COOP_UNFUSED", None)
    if unfused:
        env["GPZ_COOP_UNFUSED"] = "1"
    p = subprocess.run([sys.executable, "-c", _TASK_LIST_CHILD.format(root=root)], capture_output=True, text=True, timeout=600,
                       env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    for M, r in out.items():
        assert r["rc"] == 0 and r["info"] == 0, (M, r)
        assert r["factor"] < 1e-13 and r["inverse"] < 1e-12 and r["above"] == 0.0, (M, r)


@pytest.mark.parametrize("k", [1, 2, 32, 33, 64, 65, 96, 127, 128, 129, 160, 255, 256, 257, 300])
def test_first_failing_minor_at_sub_block_and_block_boundaries(k):
    """LAPACK's info = the order of the first leading minor that is not positive-definite.  Since round 5 the pivot sweep
    (csrc/diag128.h factor32) does not test pivots on its way: a non-positive one becomes NaN, spreads, and the diagonal is
    looked at once per 32 columns -- so the reported order is checked here at every kind of boundary (first / last column of
    a 32-column sub-block, of a 128-column block, of the matrix), against torch's own message, for a negative pivot (and,
    where it can be made exactly, a zero one: k = 1), with a second matrix in the batch that is fine."""
    from gpzoo_amd import ops
    M = 300
    A = spd(2, M, 11)
    for bad in ((-0.5, 0.0) if k == 1 else (-0.5,)):
        B = A.clone()
        Lref = torch.linalg.cholesky(A[0])
        # make the k-th pivot exactly `bad`: a_kk <- (sum of squares of the row of the factor before it) + bad
        B[0, k - 1, k - 1] = (Lref[k - 1, :k - 1] ** 2).sum() + bad
        with pytest.raises(torch.linalg.LinAlgError) as mine:
            ops.cholesky(B.cuda())
        with pytest.raises(torch.linalg.LinAlgError) as theirs:
            torch.linalg.cholesky(B)
        if bad < 0:
            assert str(mine.value) == str(theirs.value)
        assert f"minor of order {k}" in str(mine.value) and "Batch element 0" in str(mine.value)
