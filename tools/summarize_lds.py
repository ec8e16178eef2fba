#!/usr/bin/env python
"""Per-kernel LDS counter table from one `rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES` pass over bench.py.

    python tools/summarize_lds.py <counter_collection.csv> <out.txt>

SQ_LDS_IDX_ACTIVE = LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra cycles among them (MI355X_MICROARCH.md, LDS)."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1
cols = ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS", "SQ_INSTS_VMEM",
        "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
lines = [f"{'kernel':24s} {'launches':>8s} " + " ".join(f"{c.replace('SQ_', ''):>18s}" for c in cols) +
         f" {'lds_idx_active/busy':>20s} {'conflict%':>10s}"]
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    a = agg[k]
    lines.append(f"{k[:24]:24s} {max(cnt[k], 1):8d} " + " ".join(f"{a.get(c, 0):18.4g}" for c in cols) +
                 f" {a.get('SQ_LDS_IDX_ACTIVE', 0) / max(a.get('SQ_BUSY_CYCLES', 0), 1):20.3f}"
                 f" {100 * a.get('SQ_LDS_BANK_CONFLICT', 0) / max(a.get('SQ_LDS_IDX_ACTIVE', 0), 1):10.1f}")
open(sys.argv[2], "w").write("\n".join(lines[:40]) + "\n")
print("\n".join(lines[:14]))
