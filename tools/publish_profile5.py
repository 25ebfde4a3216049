#!/usr/bin/env python3
"""Copy the summaries of one tools/profile_round5.sh run into profiles/<round>/ (tracked) and write other_paths.txt, a table
that recomputes the quoted TF / GB-s figures of the other measured paths from those files alone.

    python tools/publish_profile5.py traffic gpurun_out/prof_r05 profiles/r05     (after the PMC passes)
    python tools/publish_profile5.py rest    gpurun_out/prof_r05 profiles/r05     (after the kernel traces)
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

what, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(here))
PEAK32, PEAK64, HBM = 157.3, 78.6, 8000.0


def stats(tag):
    f = glob.glob(f"{src}/trace{tag}/*/*_kernel_stats.csv")[0]
    shutil.copy(f, f"{dst}/kernel_stats{tag}.csv")
    return list(csv.DictReader(open(f)))


def jline(tag):
    ls = [l for l in open(f"{src}/trace{tag}.log") if l.startswith("{")]
    if ls:
        open(f"{dst}/bench_under_rocprof{tag}.json", "w").write(ls[-1])
        return json.loads(ls[-1])
    return None


def rows(st, pred):
    return [(r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6) for r in st if pred(r["Name"])]


if what == "traffic":
    passes = [k for k in ("fetch", "write", "l2", "sq") if os.path.isdir(src + "/pmc_" + k)]
    subprocess.run([sys.executable, here + "/pmc_summary.py", *(src + "/pmc_" + k for k in passes),
                    "--json", dst + "/pmc_per_dispatch.json"], check=True, stdout=subprocess.DEVNULL)
    pmc = json.load(open(dst + "/pmc_per_dispatch.json"))
    stage1 = [k for k in pmc if "gemmw_kernel<128, 256, 0, 1, 0" in k][0]
    v = pmc[stage1]
    N, M, L = 200_000, 2048, 32
    launches = v["dispatches"] / 3                      # bench.py --steps 2 --warmup 1 --no-extra-legs: three evaluations
    algo = (2 * L * M * N * 4 + launches * L * M * M * 4 / 2) / launches    # Kzx read + Wt write + the Linv triangle per launch
    from bench import gemm_source_hash
    out = {"kernel": stage1, "gemm_src_sha16": gemm_source_hash(),
           "workload": {"config": 3, "N": N, "M": M, "L": L, "chunk": 0, "launches_per_eval": launches},
           "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
           "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
           "hbm_bytes_per_launch_undoubled": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
           "l2_requests_per_launch": v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0),
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round5.sh), averaged over "
                   "the dispatches of `python3 bench.py --steps 2 --warmup 1 --no-extra-legs`; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads; checked for this kernel's "
                   "access pattern with tools/fetch_probe.hip in round 3); counts L2->fabric requests, Infinity-Cache hits included",
           "algorithmic_bytes_per_launch": algo}
    json.dump(out, open(dst + "/traffic_stage1.json", "w"), indent=1)
    print("traffic %.2f GB / launch, algorithmic %.2f GB" % (out["hbm_bytes_per_launch"] / 1e9, algo / 1e9))
    for k in pmc:
        if "gemmw_kernel<" in k and "GRBM_GUI_ACTIVE" in pmc[k]:
            w = pmc[k]
            print(k, "MFMA busy %.3f, L2 hit %.3f" % (w["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * w["GRBM_GUI_ACTIVE"] / 8),
                                                     w["TCC_HIT_sum"] / (w["TCC_HIT_sum"] + w["TCC_MISS_sum"])))
    sys.exit(0)

out = []
P = out.append
st = stats("")
b = jline("")
P("== timed loop of the default bench command (config 3: N=200000, M=2048, L=32, fp32) under rocprofv3 --kernel-trace --stats")
if b:
    r = b["roofline"]
    P(f"  {b['ms_per_step']:.1f} ms/step; stage 1 {r['achieved']:.1f} TF = {r['frac']:.3f} (HIP events: avg launch {r['avg_launch_ms']:.3f} ms); "
      f"traffic {r['traffic'] and r['traffic'] / 1e9:.2f} GB / launch ({r['traffic_source']})")
for name, calls, avg, tot in sorted(rows(st, lambda n: "gpz::" in n), key=lambda r: -r[3])[:8]:
    P(f"  {name[:86]:86s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")

# ---- the default line with its training legs ----
st = stats("_train")
b = jline("_train")
L, M, N = 32, 2048, 200000
P("")
P("== the default bench line with its training legs (1 warm-up + 1 timed forward; 2 x (forward + mu / Lu backward), 2 x (forward +")
P("   all-parameter backward) at config 3; then the configs[1] record)")
if b:
    for mode, v in b.get("forward_backward_roofline", {}).items():
        P(f"  config 3 {mode:15s} {v['ms']:8.1f} ms for {v['products_of_L_M2_N_flops']} units of L*M^2*N flop -> {v['achieved_TFLOPs']:.1f} TF = "
          f"{v['frac']:.3f} of {PEAK32} (floor {v['mfma_floor_ms']:.0f} ms)")
    sm = b.get("small_m")
    if sm:
        P(f"  configs[1] evaluation {sm['evaluation_ms']:.3f} ms ({sm['evaluation_roofline']['frac']:.3f} of the fp32 peak over its two products), "
          f"factor {sm['factor_ms']:.3f} ms, products {sm['products_ms']:.3f} ms")
        for mode, v in sm["forward_backward_roofline"].items():
            P(f"  configs[1] {mode:15s} {v['ms']:8.3f} ms for {v['products_of_L_M2_N_flops']} units -> {v['achieved_TFLOPs']:.1f} TF = {v['frac']:.3f} "
              f"(floor {v['mfma_floor_ms']:.2f} ms)")
for name, calls, avg, tot in sorted(rows(st, lambda n: "gemmw" in n or "kgrad" in n or "kfill" in n or "panel" in n or "coop" in n), key=lambda r: -r[3])[:14]:
    P(f"  {name[:86]:86s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")
if os.path.exists(dst + "/pmc_train.json"):
    pt = json.load(open(dst + "/pmc_train.json"))
    P("  matrix pipes and L2 under the same command (separate rocprofv3 --pmc passes, averaged over a kernel's dispatches -- launches")
    P("  that leave at their gate included -- profiles/<round>/pmc_train.json): SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)")
    for k in sorted(pt, key=lambda k: -pt[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * pt[k].get("dispatches", 0)):
        w = pt[k]
        if w.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0 or not w.get("GRBM_GUI_ACTIVE"):
            continue
        hit = w.get("TCC_HIT_sum", 0.0) / max(1.0, w.get("TCC_HIT_sum", 0.0) + w.get("TCC_MISS_sum", 0.0))
        P(f"    {k[:70]:70s} {int(w['dispatches']):4d} dispatches  MFMA pipes busy {w['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * w['GRBM_GUI_ACTIVE'] / 8):.3f}  L2 hit {hit:.3f}")

# ---- configs[1] ----
st = stats("_cfg2")
b = jline("_cfg2")
P("")
P("== configs[1] (N=50000, M=512, L=8, RBF, fp32): bench.py --config 2 --steps 10 --warmup 2, with its training legs")
if b:
    P(f"  {b['ms_per_step']:.3f} ms/step; forward + backward {b.get('forward_backward_ms')}")
for name, calls, avg, tot in sorted(rows(st, lambda n: "gpz::" in n), key=lambda r: -r[3])[:14]:
    P(f"  {name[:86]:86s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")
nt = rows(st, lambda n: "gemmw_nt_kernel" in n)
if nt:
    L2, M2, N2 = 8, 512, 50048
    big = [r for r in nt if r[2] > 0.1]
    if big:
        avg = sum(r[3] for r in big) / sum(r[1] for r in big)
        P(f"  H += W diag(gv2) W^T (gemmw_nt_kernel, launches that run): avg {avg:.3f} ms for L*M^2*N = {L2 * M2 * M2 * N2 / 1e9:.0f} Gflop of lower-tile work "
          f"-> {L2 * M2 * M2 * N2 / avg / 1e9:.1f} TF = {L2 * M2 * M2 * N2 / avg / 1e9 / PEAK32:.3f} of the fp32 peak")

# ---- minibatch ----
st = stats("_minibatch")
P("")
P("== minibatch training step (N_b=7000, M=3000, L=20, fp32; WSVGP and SVGP, factor cache off / on): tools/minibatch_step.py")
for l in open(f"{src}/trace_minibatch.log"):
    if "step =" in l:
        P("  " + l.rstrip())
for name, calls, avg, tot in sorted(rows(st, lambda n: "gpz::" in n), key=lambda r: -r[3])[:10]:
    P(f"  {name[:86]:86s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")

# ---- Poisson ----
for tag, E in (("_poisson", 3), ("_poisson20", 20)):
    st = stats(tag)
    D, Nb, Lt = 17702, 7000, 20
    P("")
    P(f"== Poisson NSF step, D={D} genes, N_b={Nb} spots, {Lt} factors, E={E} (tools/poisson_step.py {E})")
    for l in open(f"{src}/trace{tag}.log"):
        if l.startswith(("fused", "floors")):
            P("  " + l.rstrip())
    tot = 0.0
    for name, calls, avg, _ in rows(st, lambda n: "gpz::" in n and ("mfma_kernel" in n or "expf" in n or "finish" in n or "lgamma_sum" in n)):
        P(f"  {name[:86]:86s} {calls:5d} calls  avg {avg:8.3f} ms")
        tot += avg
    P(f"  kernels per step {tot:.3f} ms")

# ---- VNNGP ----
st = stats("_vnngp")
P("")
Nv, Mv, Lv, Kv = 40000, 1000, 10, 10
P(f"== VNNGP, N={Nv}, M={Mv}, L={Lv}, K={Kv}, fp32 (tools/vnngp_step.py)")
for l in open(f"{src}/trace_vnngp.log"):
    if l.startswith("VNNGP"):
        P("  " + l.rstrip())
for name, calls, avg, tot in sorted(rows(st, lambda n: "gpz::" in n), key=lambda r: -r[3])[:10]:
    P(f"  {name[:86]:86s} {calls:5d} calls  avg {avg:8.3f} ms  total {tot:9.1f} ms")
# rooflines of its point kernels: what a (latent, point) thread moves -- the K (K + 1) / 2 entries of the Kzz block and the K^2
# of the S block (fp64 gathers from the two (L, Mp, Mp) matrices, L2 / Infinity-Cache resident: 2 x 84 MB), K neighbour ids,
# its coordinates and outputs; the backward adds the record it writes (3K + 2 values) and g_mean / g_scale
gat_f = (Kv * (Kv + 1) // 2 + Kv * Kv) * 8 + Kv * 8 + 2 * 4 + Kv * 4 + 2 * 4
gat_b = gat_f + (3 * Kv + 2) * 4 + 2 * 4
for pat, by, lab in (("vnngp_point_reg_kernel", gat_f, "forward point kernel"), ("vnngp_point_bwd_reg_kernel", gat_b, "backward point kernel")):
    r = rows(st, lambda n: pat in n)
    if r:
        avg = r[0][2]
        gb = Lv * Nv * by / 1e9
        P(f"  {lab}: {by} B per (latent, point) x {Lv * Nv} = {gb:.3f} GB of 8-byte gathers in {avg:.3f} ms -> {gb / avg * 1e3:.0f} GB/s = "
          f"{gb / avg * 1e3 / HBM:.3f} of the 8 TB/s HBM peak (the gathers hit L2 / the Infinity Cache: a latency / request-rate bound, not HBM)")
r = rows(st, lambda n: "vnngp_gather_kernel" in n)
if r:
    avg = r[0][2]
    ent = Lv * Nv * Kv
    by = (3 * Kv + 2) * 4 + Kv * 4 + 8 + 4
    P(f"  fixed-order gather: {ent} (latent, point, slot) entries x {by} B (record, neighbour ids, coordinates, entry) = {ent * by / 1e9:.3f} GB in {avg:.3f} ms -> "
      f"{ent * by / avg / 1e6:.0f} GB/s = {ent * by / avg / 1e6 / HBM:.3f} of the HBM peak; {avg * 1e6 / ent * 256 * 10:.0f} ns per entry and wave "
      f"(ten resident waves per CU): bound by the ~200 dependent, mostly scalar instructions per entry, see DESIGN section 8")
open(f"{dst}/other_paths.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
