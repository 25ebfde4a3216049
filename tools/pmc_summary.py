#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel averages per dispatch.

    python tools/pmc_summary.py gpurun_out/prof_r01/pmc_* [--json out.json]
"""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        seen[k].add(r['Dispatch_Id'])
    return {k: {c: v / len(seen[k]) for c, v in cs.items()} | {'dispatches': len(seen[k])} for k, cs in agg.items()}


def main():
    argv = sys.argv[1:]
    jout = None
    if '--json' in argv:
        i = argv.index('--json')
        jout = argv[i + 1]
        del argv[i:i + 2]
    args = argv
    out = {}
    for d in args:
        for k, v in load(d).items():
            if 'gpz::' in k:
                out.setdefault(k, {}).update(v)
    for k, v in sorted(out.items()):
        print(k)
        for c, x in sorted(v.items()):
            print(f'    {c:28s} {x:.6g}')
    if jout:
        json.dump(out, open(jout, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
