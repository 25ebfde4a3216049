#!/usr/bin/env python3
"""Kernel timeline of the LAST ELBO evaluation in a rocprofv3 --kernel-trace CSV of `python3 bench.py ...`:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ev_trace -- python3 bench.py --config 2 --steps 3 --no-cpu-baseline
   python3 tools/eval_timeline.py gpurun_out/ev_trace
An evaluation starts at the Kzz fill (the kfill launch that precedes the factorisation) and ends with elbo_sum_kernel; prints each kernel's
start, duration and the idle gap in front of it, then busy / wall totals."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
diag = [i for i, n in enumerate(names) if "diag128" in n or "coop_factor" in n]
# first factor launch of the last evaluation: the last one whose predecessor is the Kzz fill
starts = [i for i in diag if i > 0 and "kfill" in names[i - 1]]
first = starts[-1] - 1
last = max(i for i, n in enumerate(names) if "elbo_sum" in n)
t0 = int(rows[first]["Start_Timestamp"])
prev_end, busy = t0, 0
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("gpz::", "").replace("void ", "")
    busy += e - max(s, prev_end) if e > prev_end else 0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n[:90]}")
    prev_end = max(prev_end, e)
print(f"wall {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, kernels {last - first + 1}")
