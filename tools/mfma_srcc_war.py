#!/usr/bin/env python3
"""Scan gfx950 ISA (hipcc -S output) for a write-after-read pattern hipcc does not guard: an MFMA whose C operand is a
register range other than its D, followed within a few instructions by a memory load (ds_read / global / scratch /
buffer) that RETURNS INTO that C range.  The fp64 MFMA reads the last register pairs of C in its last passes; with the
matrix pipe backed up a returning LDS read can get there first (seen in csrc/coop.hip, round 5).
    python tools/mfma_srcc_war.py file.s [window=24]"""
import re
import sys

path = sys.argv[1]
window = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)\b")


def regs(tok):
    m = rng.fullmatch(tok.strip().rstrip(","))
    if not m:
        return None
    if m.group(1):
        return int(m.group(1)), int(m.group(2))
    return int(m.group(3)), int(m.group(3))


lines = [l.split(";")[0].strip() for l in open(path)]
ins = [(n + 1, l) for n, l in enumerate(lines) if l and not l.startswith(".") and not l.endswith(":")]
hits = 0
for k, (ln, l) in enumerate(ins):
    if not l.startswith("v_mfma"):
        continue
    ops = l.split(None, 1)[1].split(",")
    d, c = regs(ops[0]), regs(ops[3].split()[0]) if len(ops) > 3 else None
    if not d or not c or d == c:
        continue
    for ln2, l2 in ins[k + 1:k + 1 + window]:
        if l2.startswith(("ds_read", "global_load_dword", "scratch_load", "buffer_load", "flat_load")) and "lds" not in l2.split()[0]:
            w = regs(l2.split(None, 1)[1].split(",")[0])
            if w and not (w[1] < c[0] or w[0] > c[1]):
                print(f"{path}:{ln}: {l}\n    {ln2}: {l2}")
                hits += 1
                break
print(f"{path}: {hits} MFMA(s) whose C registers are reloaded within {window} instructions")
