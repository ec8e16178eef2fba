#!/usr/bin/env python
"""Condense rocprofv3 output (kernel_stats / counter_collection CSVs) into small summaries for profiles/.

    python tools/summarize_rocprof.py <rocprof_out_dir> <summary.txt>
"""
import csv
import glob
import os
import sys
from collections import defaultdict


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import NAME_MAP  # noqa: E402  (library kernel -> HIP-event recorder name)


def short(name):
    for pat, nm in NAME_MAP:
        if pat in name:
            return f"{nm}  [{pat}]"
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0]
    return name[-70:]


def main(d, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append(f"== {os.path.relpath(f, d)} (top 25 by total time)")
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
        lines.append(f"{'kernel':72s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
        for r in rows[:25]:
            lines.append(f"{short(r['Name']):72s} {r['Calls']:>8s} {float(r['TotalDurationNs']) / 1e6:10.2f} "
                         f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
        # library kernels grouped under the names of bench.py's HIP-event recorder (template instantiations of one
        # kernel -- e.g. the compile-time GEMM epilogues -- are separate rocprof rows): call-weighted mean duration
        grp = defaultdict(lambda: [0, 0.0])
        for r in rows:
            for pat, nm in NAME_MAP:
                if pat in r["Name"]:
                    grp[nm][0] += int(r["Calls"])
                    grp[nm][1] += float(r["TotalDurationNs"])
                    break
        lines.append("-- grouped by recorder name (compare with bench.py \"kernels\"[].avg_us)")
        for nm, (n, t) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
            lines.append(f"{nm:72s} {n:8d} {t / 1e6:10.2f} {t / max(n, 1) / 1e3:10.1f}")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        lines.append(f"== {os.path.relpath(f, d)} (per-kernel counter means)")
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
        lines.append(f"{'kernel':72s} {'counter':>14s} {'dispatches':>10s} {'mean':>16s} {'sum':>18s}")
        for (k, c), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            lines.append(f"{k:72s} {c:>14s} {n:10d} {s / n:16.1f} {s:18.1f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
