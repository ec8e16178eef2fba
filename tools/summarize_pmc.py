#!/usr/bin/env python
"""Condense a rocprofv3 counter_collection.csv: per (kernel, counter) mean value over dispatches.
Usage: summarize_pmc.py <..._counter_collection.csv> [kernel-substring]"""
import collections
import csv
import sys

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0.0, 0])
grid = {}
with open(path) as f:
    for row in csv.DictReader(f):
        k = row.get("Kernel_Name", "")
        if flt and flt not in k:
            continue
        short = k.split("(")[0][-60:]
        key = (short, row.get("Grid_Size", ""), row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"])
        acc[key][1] += 1
print(f"{'kernel':60s} {'grid':>10s} {'counter':32s} {'mean/dispatch':>16s} {'n':>4s}")
for (k, g, c), (v, n) in sorted(acc.items()):
    print(f"{k:60s} {g:>10s} {c:32s} {v / n:16.1f} {n:4d}")
