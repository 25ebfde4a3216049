#!/usr/bin/env python3
"""Timeline of the LAST batched-Cholesky call in a rocprofv3 --kernel-trace CSV of tools/potrf_only.py:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/potrf_trace -- python3 tools/potrf_only.py 32 2048
   python3 tools/potrf_timeline.py gpurun_out/potrf_trace
prints start / duration / gap of every kernel between the last call's first and last diag128_kernel."""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
diag = [i for i, n in enumerate(names) if "diag128" in n]
# calls are separated by the pad/copy kernels: walk back from the last diag to the first diag of that call
last = diag[-1]
first = last
for i in reversed(diag):
    if all("pad_copy" not in names[j] for j in range(i, first)):
        first = i
    else:
        break
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
for r in rows[first:last + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = "diag" if "diag128" in n else ("gemm" if "gemm128" in n else n[:20])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  q={r.get('Queue_Id', '?'):>3}  grid={r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {short}")
    prev_end = max(prev_end, e)
print(f"total {(prev_end - t0) / 1e3:.1f} us")
