#!/usr/bin/env python
"""Where does a split-half GEMM launch spend its time?  t(M, K) sweep at N = 384 (q / to_out shape class) to separate
the per-launch cost, the per-tile cost (prologue + epilogue + pipeline fill) and the per-K-tile cost of the main loop.
Prints a table and a least-squares fit  t = a + tiles_per_slot * (T0 + KT * tk)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
R = 6464 * 16
N = int(os.environ.get("N", "384"))
rows = []
bufs = {}
for K in (128, 384, 768, 1536, 3072):
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    wp = ops.pack_weight(w)
    for M in (R // 8, R // 4, R // 2, R, 2 * R):
        a = ops.split_rows(torch.randn(M, K, device=dev))
        out = torch.empty(M, N, device=dev)
        bufs[(M, K)] = (a, w, b, wp, out)
for rd in range(6):
    for (M, K), (a, w, b, wp, out) in bufs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(a, w, bias=b, out=out, packed=wp)
        e1.record()
        e1.synchronize()
        if rd > 0:
            rows.append((M, K, e0.elapsed_time(e1) * 1e3))
med = {}
for M, K, t in rows:
    med.setdefault((M, K), []).append(t)
print(f"N={N}  CTK_GEMM_TILE={os.environ.get('CTK_GEMM_TILE', '0')}")
print(f"{'M':>8s} {'K':>6s} {'tiles':>7s} {'us(med)':>9s} {'TF/s':>8s} {'us per tile-round':>18s}")
A, y = [], []
for (M, K), ts in sorted(med.items()):
    t = float(np.median(ts))
    tiles = ((M + 127) // 128) * (N // 128)
    rounds = tiles / 512.0
    print(f"{M:8d} {K:6d} {tiles:7d} {t:9.1f} {2.0 * M * N * K / t / 1e6:8.1f} {t / max(rounds, 1e-9):18.2f}")
    A.append([1.0, rounds, rounds * (K // 32)])
    y.append(t)
sol, *_ = np.linalg.lstsq(np.array(A), np.array(y), rcond=None)
print(f"fit t = a + rounds * (T0 + KT * tk):  a = {sol[0]:.2f} us per launch, T0 = {sol[1]:.2f} us per tile-round, tk = {sol[2]:.3f} us per K-tile")
print("ideal tk (24 MFMA x 32 cyc x 2 waves per SIMD) = 0.64 us at 2.4 GHz, 0.73 us at 2.1 GHz")
