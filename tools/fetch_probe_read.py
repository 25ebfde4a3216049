#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE of tools/fetch_probe.hip from a rocprofv3 --pmc FETCH_SIZE run:  fetch_probe_read.py <dir>"""
import collections
import csv
import glob
import sys

agg, n = collections.defaultdict(float), collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            agg[r["Kernel_Name"]] += float(r["Counter_Value"])
            n[r["Kernel_Name"]] += 1
for k in agg:
    print("%-60s launches %d  FETCH_SIZE per launch %.4g KB = %.3f GiB" % (k[:60], n[k], agg[k] / n[k], agg[k] / n[k] / 1048576))
