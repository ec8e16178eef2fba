#!/usr/bin/env python
"""Per-kernel sums of a rocprofv3 --pmc pass (counter_collection.csv) as one table: kernel x counter, plus the SQ ratios
MI355X_MICROARCH.md defines (ACTIVE / WAIT_INST_ANY = issue stall / WAIT_ANY = parked in s_waitcnt or barrier, in % of
SQ_WAVE_CYCLES; MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves-per-SIMD ...) is left to the reader).
    summarize_counters.py <rocprof_dir> <out.txt>"""
import csv, glob, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short

d, out = sys.argv[1:3]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
names = []
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        if c not in names:
            names.append(c)
        agg[k][c] += float(r["Counter_Value"])
        if c == names[0]:
            cnt[k] += 1
lines = [f"{'kernel':30s} {'launches':>8s} " + " ".join(f"{n.replace('SQ_', ''):>20s}" for n in names) + "   active% issue-stall% parked%  lds-conflict%"]
key = "SQ_WAVE_CYCLES"
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:40]:
    wc = v.get(key, 0) or 1
    extra = f"   {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.1f} {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:11.1f} {100 * v.get('SQ_WAIT_ANY', 0) / wc:7.1f}"
    if "SQ_LDS_IDX_ACTIVE" in v:
        extra += f" {100 * v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v['SQ_LDS_IDX_ACTIVE'], 1):13.1f}"
    lines.append(f"{k[:30]:30s} {cnt[k]:8d} " + " ".join(f"{v.get(n, 0):20.4g}" for n in names) + extra)
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
