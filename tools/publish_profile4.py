#!/usr/bin/env python3
"""Copy the kernel-trace summaries of the OTHER measured paths (tools/profile_round4.sh) into profiles/<round>/ and write
a table, other_paths.txt, that recomputes the quoted TF / GB-s figures from those files alone.

    python tools/publish_profile4.py gpurun_out/prof_r04 profiles/r04
"""
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
PEAK32, PEAK64 = 157.3, 78.6


def stats(tag):
    f = glob.glob(f"{src}/trace_{tag}/*/*_kernel_stats.csv")[0]
    shutil.copy(f, f"{dst}/kernel_stats_{tag}.csv")
    return list(csv.DictReader(open(f)))


def jline(tag):
    ls = [l for l in open(f"{src}/trace_{tag}.log") if l.startswith("{")]
    if ls:
        open(f"{dst}/bench_under_rocprof_{tag}.json", "w").write(ls[-1])
        return json.loads(ls[-1])
    return None


def rows(st, pred):
    return [(r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6) for r in st if pred(r["Name"])]


out = []
P = out.append
# ---- forward + backward at config 3 (bench.py --with-backward: 1 warm-up + 1 timed forward, then 2 x (forward + mu/Lu
# backward) and 2 x (forward + all-parameter backward)) ----
st = stats("bwd")
b = jline("bwd")
L, M, N = 32, 2048, 200000
prod = L * float(M) * M * N
P("== forward + backward at config 3 (N=200000, M=2048, L=32, fp32) under rocprofv3 --kernel-trace --stats")
if b and "forward_backward_roofline" in b:
    for mode, v in b["forward_backward_roofline"].items():
        P(f"  {mode:15s} {v['ms']:8.1f} ms for {v['products_of_L_M2_N_flops']} products of L*M^2*N flops -> {v['achieved_TFLOPs']:.1f} TF = "
          f"{v['frac']:.3f} of {PEAK32} (floor {v['mfma_floor_ms']:.0f} ms)")
for name, calls, avg, tot in sorted(rows(st, lambda n: "gemmw" in n or "kgrad" in n or "kfill" in n), key=lambda r: -r[3])[:12]:
    P(f"  {name[:78]:78s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")
nt = rows(st, lambda n: "gemmw_nt_kernel" in n)
if nt:
    _, calls, avg, _ = nt[0]
    chunk = 12288
    fl = L * float(M) * M * chunk           # lower-tile flops of one (M x n)(n x M) accumulation over a full chunk
    P(f"  A.B^T accumulate (gemmw_nt_kernel): avg {avg:.3f} ms per launch over all chunk sizes; a full 12288-column chunk is "
      f"{fl / 1e12:.3f} Tflop of lower-tile work")
# ---- Poisson ----
for tag, E in (("poisson", 3), ("poisson20", 20)):
    st = stats(tag)
    D, Nb, Lt = 17702, 7000, 20
    P(f"== Poisson NSF step, D={D}, N_b={Nb}, {Lt} factors, E={E} (tools/poisson_step.py {E})")
    tot_ms = 0.0
    for name, calls, avg, tot in sorted(rows(st, lambda n: "mfma_kernel" in n or "expf_kernel" in n or "poisson_finish" in n), key=lambda r: -r[3]):
        P(f"  {name[:78]:78s} {calls:5d} calls  avg {avg:8.3f} ms")
        tot_ms += avg
    flops = 8.0 * Lt * E * D * Nb          # rate (both passes), dexpF, dW: 4 products of 2 Lt flops per (e, d, n)
    P(f"  one call = {tot_ms:.3f} ms of kernels -> {flops / tot_ms / 1e9:.1f} TF of {PEAK32} fp32 MFMA peak = {flops / tot_ms / 1e9 / PEAK32:.3f}")
# ---- configs[4] per rank, configs[1] ----
for tag, what in (("cfg5", "configs[4] as one of eight ranks holds it: MGGP, N=200000, M=2048, L=4, fp64"),
                  ("cfg2", "configs[1]: N=50000, M=512, L=8, RBF, fp32")):
    st = stats(tag)
    b = jline(tag)
    P(f"== {what}")
    if b:
        k = b["kernels"]
        P(f"  {b['ms_per_step']:.2f} ms per evaluation | stage 1 {b['roofline']['achieved']:.1f} TF = {b['roofline']['frac']:.3f} of "
          f"{b['roofline']['peak']} | stage 2 {k['stage2_LuT_Wt']['achieved_TFLOPs']:.1f} TF | fill {(k['kuf_fill']['achieved_GBps'] or 0):.0f} GB/s | "
          f"factor {k['potrf_ms_per_eval']:.3f} ms ({k.get('factor', {}).get('achieved_TFLOPs', 0):.1f} TF fp64)")
    for name, calls, avg, tot in sorted(rows(st, lambda n: "gpz::" in n), key=lambda r: -r[3])[:8]:
        P(f"  {name[:78]:78s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")
# ---- the panel kernel: both products in one launch ----
for tag, what, Lp, Mq, Nq in (("m256", "N=200000, M=256, L=32, Matern-3/2, fp32: the library's own choice", 32, 256, 200000),
                              ("cfg2panel", "configs[1] with --panel-products: N=50000, M=512, L=8, RBF, fp32", 8, 512, 50000)):
    if not glob.glob(f"{src}/trace_{tag}/*/*_kernel_stats.csv"):
        continue
    st = stats(tag)
    b = jline(tag)
    P(f"== panel kernel (csrc/gemmp.hip), {what}")
    if b:
        P(f"  {b['ms_per_step']:.2f} ms per evaluation | {b['roofline']['kernel'][:60]}: {b['roofline']['achieved']:.1f} TF over BOTH products "
          f"= {b['roofline']['frac']:.3f} of {b['roofline']['peak']} | fill {(b['kernels']['kuf_fill']['achieved_GBps'] or 0):.0f} GB/s (0: computed inside the kernel)")
    for name, calls, avg, tot in sorted(rows(st, lambda n: "gpz::" in n), key=lambda r: -r[3])[:5]:
        P(f"  {name[:78]:78s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")
    pk = rows(st, lambda n: "panel_kernel" in n)
    ev = rows(st, lambda n: "coop_factor_kernel" in n)          # one factorisation per evaluation
    if pk and ev:
        fl = 2.0 * Lp * float(Mq) * Mq * Nq     # two triangular products of L M^2 N flops per evaluation
        per_eval = pk[0][3] / ev[0][1]          # ms of panel_kernel launches per evaluation (one launch per chunk of columns)
        P(f"  panel_kernel: {pk[0][1] // ev[0][1]} launch(es) = {per_eval:.3f} ms per evaluation for {fl / 1e9:.1f} Gflop -> "
          f"{fl / per_eval / 1e9:.1f} TF = {fl / per_eval / 1e9 / PEAK32:.3f}  (the wall time per evaluation above is under the profiler)")
open(f"{dst}/other_paths.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
