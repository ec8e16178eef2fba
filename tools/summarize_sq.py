#!/usr/bin/env python
"""Per-kernel SQ counter table from one `rocprofv3 --pmc <SQ counters> -- python bench.py ...` pass.

    python tools/summarize_sq.py <counter_collection.csv> <kernel_stats-or-empty> <out.txt>

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per
v_mfma_f32_32x32x16_f16).  mfma_busy_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs: divided by the kernel's duration
in cycles it is the MFMA-pipe utilisation."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402


def main(path, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
    cols = ["SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA",
            "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"]
    lines = [f"{'kernel':24s} {'launches':>8s} " + " ".join(f"{c.replace('SQ_', ''):>20s}" for c in cols) +
             f" {'active%':>8s} {'issue-stall%':>12s} {'wait%':>7s} {'VALU/MFMA':>9s} {'mfma_busy_cyc/SIMD/launch':>26s}"]
    for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
        a = agg[k]
        w = max(a.get("SQ_WAVE_CYCLES", 0), 1)
        n = max(cnt[k], 1)
        lines.append(f"{k[:24]:24s} {n:8d} " + " ".join(f"{a.get(c, 0):20.4g}" for c in cols) +
                     f" {100 * a.get('SQ_ACTIVE_INST_ANY', 0) / w:8.1f} {100 * a.get('SQ_WAIT_INST_ANY', 0) / w:12.1f}"
                     f" {100 * a.get('SQ_WAIT_ANY', 0) / w:7.1f} {a.get('SQ_INSTS_VALU', 0) / max(a.get('SQ_INSTS_MFMA', 0), 1):9.1f}"
                     f" {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / n:26.0f}")
    open(out, "w").write("\n".join(lines[:40]) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[-1])
