#!/usr/bin/env python3
"""Timeline of the one-launch factorisation (csrc/coop.hip): per-task claim / end stamps and polling time.

    python tools/coop_trace.py L M [inverse=1]
Prints the launch's span, per-kind task counts and mean durations net of polling, and the utilisation of the clusters;
GPZ_TRACE_ALL=1: every task of matrix 0 with its claim / end times (for a diagonal block: `park` = factor, `mult` = inverse
levels); GPZ_COOP_UNFUSED=1: round 4's task list (tiles (j, j-1) and (j, j) as two tasks).
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops, _lib  # noqa: E402

L, M = int(sys.argv[1]), int(sys.argv[2])
INV = len(sys.argv) > 3 and sys.argv[3] == "1"
lib = _lib.load()
lib.gpz_debug_coop_trace.restype = C.c_int
lib.gpz_debug_coop_trace.argtypes = [C.c_void_p]
g = torch.Generator().manual_seed(0)
B = torch.randn(L, M, M, generator=g, dtype=torch.float64)
A = (B @ B.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64)).cuda()
nblk = (M + 127) // 128
ntasks = nblk * (nblk + 1) // 2 + (nblk * (nblk - 1) // 2 if INV else 0)
if not os.environ.get("GPZ_COOP_UNFUSED"):
    ntasks -= nblk - 1              # tiles (j, j-1) and (j, j) are one task (kind 2) ...
    ntasks += max(nblk - 2, 0)      # ... and the early sums of tile (j, j), j >= 2, another (kind 3)
trace = torch.zeros(L * ntasks * 8, dtype=torch.int64, device="cuda")
if INV:
    assert M % 128 == 0
    lib.gpz_debug_factor_sync_words.restype = C.c_size_t
    lib.gpz_debug_factor_sync_words.argtypes = [C.c_int64, C.c_int64]
    lib.gpz_debug_factor_invert.restype = C.c_int
    lib.gpz_debug_factor_invert.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6
    Dinv = torch.empty(L * nblk * 128 * 128, dtype=torch.float64, device="cuda")
    Linv = torch.full((L, M, M), float("nan"), dtype=torch.float64, device="cuda")
    T = torch.empty(L, M, M, dtype=torch.float64, device="cuda")
    sync = torch.empty(lib.gpz_debug_factor_sync_words(M, L), dtype=torch.int32, device="cuda")
    info = torch.empty(L, dtype=torch.int32, device="cuda")

    def run():
        W = A.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.gpz_debug_factor_invert(W.data_ptr(), M, L, Dinv.data_ptr(), Linv.data_ptr(), T.data_ptr(), sync.data_ptr(),
                                         info.data_ptr(), None)
        torch.cuda.synchronize()
        assert rc == 0 and not bool(info.any()), (rc, info)
        return W, 1e3 * (time.perf_counter() - t0)
else:
    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        W = ops.cholesky(A)
        torch.cuda.synchronize()
        return W, 1e3 * (time.perf_counter() - t0)
run()
W, ms = run()
print(f"host-timed call: {ms:.3f} ms")
lib.gpz_debug_coop_trace(trace.data_ptr())
W, _ = run()
lib.gpz_debug_coop_trace(None)
if INV:
    Lc = torch.tril(W)
    print("max |L L^T - A| =", float((Lc @ Lc.transpose(-1, -2) - A).abs().max()), " max |Linv L - I| =",
          float((Linv @ Lc - torch.eye(M, dtype=torch.float64, device="cuda")).abs().max()),
          " Linv above the diagonal:", float(torch.triu(Linv, 1).abs().max()))
t = trace.cpu().view(L, ntasks, 8)
t0 = int(t[:, :, 2].min())
span = (int(t[:, :, 3].max()) - t0) / 100.0
print(f"L={L} M={M} nblk={nblk}: span {span:.1f} us, {ntasks} tasks per matrix")
code = t[:, :, 0]
kind = code >> 12
ii = (code >> 6) & 63
jj = code & 63
dur = (t[:, :, 3] - t[:, :, 2]).double() / 100.0
poll = t[:, :, 4].double() / 100.0
net = dur - poll
diag = ((kind == 0) | (kind == 2)) & (ii == jj)
off = (kind == 0) & (ii != jj)
print(f"  diag tiles: mean dur {float(dur[diag].mean()):.1f} us, polling {float(poll[diag].mean()):.1f}")
print(f"  off tiles : mean dur {float(dur[off].mean()):.1f} us, polling {float(poll[off].mean()):.1f}")
# net time per k-block: regress net on j for off tiles
jo = jj[off].double()
no = net[off]
Amat = torch.stack([jo, torch.ones_like(jo)], 1)
sol = torch.linalg.lstsq(Amat, no[:, None]).solution
print(f"  off tiles net = {float(sol[0]):.2f} us per k-block + {float(sol[1]):.2f} us")
jd = jj[diag].double()
nd = net[diag]
Amat = torch.stack([jd, torch.ones_like(jd)], 1)
sol = torch.linalg.lstsq(Amat, nd[:, None]).solution
print(f"  diag tiles net = {float(sol[0]):.2f} us per k-block + {float(sol[1]):.2f} us")
acc_t = t[:, :, 5].double() / 100.0
segs = t[:, :, 6].double()
kb_total = float(jj[kind == 0].double().sum() + (ii - jj)[kind == 1].double().sum())
print(f"  accumulation loops: {float(acc_t.sum()):.0f} us over {kb_total:.0f} k-blocks = {float(acc_t.sum()) / kb_total:.2f} us per k-block; {float(segs.sum()) / max(1.0, float((jj > 0).sum())):.2f} segments per tile with a sum")
if INV:
    xt = kind == 1
    print(f"  inverse tiles: mean dur {float(dur[xt].mean()):.1f} us, polling {float(poll[xt].mean()):.1f}")
    jo = (ii - jj)[xt].double()
    Amat = torch.stack([jo, torch.ones_like(jo)], 1)
    sol = torch.linalg.lstsq(Amat, net[xt][:, None]).solution
    print(f"  inverse tiles net = {float(sol[0]):.2f} us per k-block + {float(sol[1]):.2f} us")
cyc = t[:, :, 7].double()
if False: print(f"  shader clock inside the loops: {float(cyc.sum()) / max(1.0, float(acc_t.sum())):.0f} cycles/us; {float(cyc.sum()) / kb_total / 8:.0f} cycles per k-tile (4096 = MFMA bound)")
e1 = t[:, :, 6].double() / 100.0
pl = t[:, :, 7].double() / 100.0
tail = dur - acc_t - e1 - pl - poll
print(f"  off tiles: park + wait {float(e1[off].mean()):.1f} us (incl. polling), multiply loop {float(pl[off].mean()):.1f} us, rest (claim, store, publish) {float((dur - acc_t - e1 - pl)[off].mean()):.1f} us")
if INV:
    print(f"  inverse tiles: park + wait {float(e1[kind == 1].mean()):.1f} us, multiply loop {float(pl[kind == 1].mean()):.1f} us, rest {float((dur - acc_t - e1 - pl)[kind == 1].mean()):.1f} us")
wg = t[:, :, 1]
nw = len(torch.unique(wg))
busy = float(net.sum())
print(f"  {nw} workgroups, sum of net task time {busy:.0f} us -> utilisation {busy / (nw * span):.2f}; polling total {float(poll.sum()):.0f} us")
# matrix 0: the diagonal chain
m0 = t[0]
for tk in range(ntasks):
    c = int(m0[tk, 0])
    i, j = (c >> 6) & 63, c & 63
    if i == j and (c >> 12) in (0, 2):
        print(f"    D({j:2d}) ticket {tk:3d} wg {int(m0[tk,1]):3d} claim {(int(m0[tk,2])-t0)/100:8.1f} end {(int(m0[tk,3])-t0)/100:8.1f} poll {int(m0[tk,4])/100:7.1f}")
if os.environ.get("GPZ_TRACE_ALL"):
    print("    every task of matrix 0 (us from the first claim): kind i j | wg | claim end | polling, accumulate, park+wait, multiply")
    rows = sorted(range(ntasks), key=lambda k: int(m0[k, 2]))
    for tk in rows:
        c = int(m0[tk, 0])
        k, i, j = c >> 12, (c >> 6) & 63, c & 63
        print(f"    {'CXFP'[k]}({i:2d},{j:2d}) wg {int(m0[tk,1]):3d} claim {(int(m0[tk,2])-t0)/100:8.1f} end {(int(m0[tk,3])-t0)/100:8.1f} | poll {int(m0[tk,4])/100:6.1f} acc {int(m0[tk,5])/100:6.1f} park {int(m0[tk,6])/100:6.1f} mult {int(m0[tk,7])/100:6.1f}")
