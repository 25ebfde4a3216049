#!/usr/bin/env python3
"""Turn tools/peaks.sh output into the peaks.json bench.py reads: the best measured rate of each kind.

    python3 tools/peaks_json.py gpurun_out/peaks.txt profiles/r02/peaks.json
"""
import json
import re
import sys

txt = open(sys.argv[1]).read()
f32 = [float(m.group(1)) for m in re.finditer(r"pure mfma\s+shape=16.*?([\d.]+) TF", txt)]
f64 = [float(m.group(1)) for m in re.finditer(r"f64 16x16x4 nacc\S+ \S+ blocks/CU=\d+\s+[\d.]+ ms\s+([\d.]+) TF", txt)]
# write-only: the grid-strided store loop, the contiguous-piece and multi-stream variants and hipMemsetAsync -- the best
# of them is the ceiling the Kzx fill is held against (round 3 took the first form alone: a probe below the fill itself)
wr = [float(m.group(1)) for m in re.finditer(r"write (\d+) GB/s", txt)]
wr += [float(m.group(1)) for m in re.finditer(r"hbm (?:write,|hipMemsetAsync)[^\n]*?: (\d+) GB/s", txt)]
rd = [float(m.group(1)) for m in re.finditer(r"read (\d+) GB/s", txt)]
cp = [float(m.group(1)) for m in re.finditer(r"copy (\d+) GB/s", txt)]
out = {"mfma_f32_TFLOPs": max(f32), "mfma_f64_TFLOPs": max(f64), "hbm_write_GBps": max(wr), "hbm_read_GBps": max(rd),
       "hbm_copy_GBps": max(cp),
       "mfma_f64_note": "register-resident v_mfma_f64_16x16x4_f64 loop with the accumulators tied to VGPRs (inline asm): 64 shader "
                        "cycles per MFMA per SIMD at the ~2.39 GHz the chip holds = the 78.6 TF datasheet rate.  (Round 2's 49 TF "
                        "probe used the builtin: hipcc kept the accumulators in AGPRs and emitted 16 v_accvgpr moves per MFMA, which "
                        "cost their issue time -- vector instructions are not free beside an MFMA.)",
       "how": "tools/peaks.sh on the MI355X box: register-resident v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 loops "
              "(tools/mfma_probe_f32.hip, _f64.hip) and 16-byte-per-lane streaming kernels over 4 GiB buffers (tools/hbm_probe.hip); "
              "best of the occupancies tried; the write rate is the best of the store-loop shapes and hipMemsetAsync"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
