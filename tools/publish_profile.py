#!/usr/bin/env python3
"""Copy the summaries of one tools/profile_round.sh run into profiles/<round>/ (tracked):
kernel_stats.csv, bench_under_rocprof.json, pmc_per_dispatch.json, traffic_stage1.json.

    python tools/publish_profile.py gpurun_out/prof_r01e profiles/r01
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(src + "/trace/*/*_kernel_stats.csv")[0], dst + "/kernel_stats.csv")
line = [l for l in open(src + "/trace.log") if l.startswith("{")][-1]
open(dst + "/bench_under_rocprof.json", "w").write(line)
bench = json.loads(line)
here = os.path.dirname(os.path.abspath(__file__))
passes = [k for k in ("fetch", "write", "l2", "sq", "lds") if os.path.isdir(src + "/pmc_" + k)]
subprocess.run([sys.executable, here + "/pmc_summary.py", *(src + "/pmc_" + k for k in passes),
                "--json", dst + "/pmc_per_dispatch.json"], check=True, stdout=subprocess.DEVNULL)
pmc = json.load(open(dst + "/pmc_per_dispatch.json"))
# stage 1 = Wt = Linv * Kzx: gemmw_kernel<128, 256, mem, lower, store+stats, ..> (fp32) or gemm128_kernel<T, NI, NN, EPI_STORE_STATS, ..>
stage1 = ([k for k in pmc if "gemmw_kernel<128, 256, 0, 1, 0" in k] or
          [k for k in pmc if "gemm128_kernel<float" in k and ", false, 1," in k])[0]
v = pmc[stage1]
cfgs = bench["config"]["workload"]
N, M, L = (int(cfgs.split(f"{t}=")[1].split()[0].rstrip(",")) for t in ("N", "M", "L"))
Mp = (M + 127) // 128 * 128
launches = v["dispatches"] / 3                      # bench.py --steps 2 --warmup 1
esz = 4
algo = (2 * L * Mp * N * esz + launches * L * Mp * Mp * esz / 2) / launches    # Kzx read + Wt write + the Linv triangle per launch
sys.path.insert(0, os.path.dirname(here))
from bench import gemm_source_hash  # noqa: E402
out = {"kernel": stage1, "gemm_src_sha16": gemm_source_hash(), "workload": {"config": 3, "N": N, "M": M, "L": L, "chunk": 0, "launches_per_eval": launches},
       "FETCH_SIZE_KB_per_launch": v["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": v["WRITE_SIZE"],
       "hbm_bytes_per_launch": (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
       # the x2 of the gfx950 note holds for this kernel's 64-byte row-segment reads too: tools/fetch_probe.hip (L2 fetches
       # whole 128-byte lines and tallies each as 64 bytes); the raw figure is kept for reference
       "hbm_bytes_per_launch_undoubled": (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024,
       "l2_requests_per_launch": v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0),
       "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_round.sh), averaged over "
               "the dispatches of `python3 bench.py --steps 2 --warmup 1`; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 reports half the bytes of wide coalesced reads); counts L2->fabric requests, Infinity-Cache hits included",
       "algorithmic_bytes_per_launch": algo}
json.dump(out, open(dst + "/traffic_stage1.json", "w"), indent=1)
print("bench under rocprof: %.1f ms/step, stage-1 avg launch %.3f ms, %.1f TF" % (
    bench["ms_per_step"], bench["roofline"]["avg_launch_ms"], bench["roofline"]["achieved"]))
for r in csv.DictReader(open(dst + "/kernel_stats.csv")):
    if "gemm128_kernel<float" in r["Name"] or "gemmw_kernel<" in r["Name"]:
        print("rocprof:", r["Name"][:60], r["Calls"], "calls, avg %.3f ms" % (float(r["AverageNs"]) / 1e6))
print("traffic %.2f GB / launch, algorithmic %.2f GB" % (out["hbm_bytes_per_launch"] / 1e9, algo / 1e9))
for k in pmc:
    if ("gemm128_kernel<float" in k or "gemmw_kernel<" in k) and "GRBM_GUI_ACTIVE" in pmc[k]:
        w = pmc[k]
        print(k, "MFMA busy %.3f, L2 hit %.3f" % (w["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * w["GRBM_GUI_ACTIVE"] / 8),
                                                 w["TCC_HIT_sum"] / (w["TCC_HIT_sum"] + w["TCC_MISS_sum"])))
