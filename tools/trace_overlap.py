#!/usr/bin/env python
"""Do two HIP streams really overlap?  Reads a rocprofv3 --kernel-trace CSV of a bench step run with CTK_OVERLAP=8 and reports,
for every dispatch on the less-used queue (the auxiliary stream), how much of its duration other kernels were running and which.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-lines
    python tools/trace_overlap.py /tmp/kt"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"][:60], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
per_q = collections.Counter(q for _, _, q, _, _ in rows)
print("dispatches per queue:", dict(per_q))
if len(per_q) < 2:
    sys.exit("single queue: nothing to overlap")
aux_q = min(per_q, key=per_q.get)
aux = [r for r in rows if r[2] == aux_q]
main = [r for r in rows if r[2] != aux_q]
tot = ov = 0
conc = collections.Counter()
j0 = 0
for s, e, _, name, grid in aux:
    while j0 < len(main) and main[j0][1] <= s:
        j0 += 1
    j = j0
    o = 0
    while j < len(main) and main[j][0] < e:
        a, b = max(s, main[j][0]), min(e, main[j][1])
        if b > a:
            o += b - a
            conc[main[j][3][:40]] += b - a
        j += 1
    tot += e - s
    ov += min(o, e - s)
print(f"aux queue {aux_q}: {len(aux)} dispatches, {tot / 1e6:.2f} ms busy, of which {ov / 1e6:.2f} ms ({100.0 * ov / max(tot, 1):.1f} %) while a main-queue kernel ran")
for k, v in conc.most_common(8):
    print(f"   beside {k:42s} {v / 1e6:8.2f} ms")
names = collections.Counter(n for _, _, _, n, _ in aux)
print("aux kernels:", dict(names.most_common(4)))
d = collections.defaultdict(list)
for s, e, _, n, g in aux:
    d[n].append((e - s) / 1e3)
for n, v in d.items():
    print(f"   {n:60s} n={len(v)} avg {sum(v) / len(v):.1f} us")
