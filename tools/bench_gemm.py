#!/usr/bin/env python
"""Per-shape GEMM micro-benchmark on the exact launches of one C3 update iteration (S=16, N=6400).
Interleaved rounds in one process; reports TFLOP/s per shape and the time-weighted aggregate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import ops  # noqa: E402

P, V = 6400 * 16, 64 * 16
R = P + V
SHAPES = [  # name, M, K, N, act, resid, count per iteration, k_valid
    ("corr_fc1", 4 * P, 2432, 384, 1, False, 1, 2401),
    ("corr_fc2", 4 * P, 384, 256, 0, False, 1, 384),
    ("in_proj", P, 1120, 384, 0, False, 1, 1110),
    ("q_all", R, 384, 384, 0, False, 3, 384),
    ("kv_all", R, 384, 768, 0, False, 3, 384),
    ("out_all", R, 384, 384, 0, True, 3, 384),
    ("fc1_all", R, 384, 1536, 2, False, 3, 384),
    ("fc2_all", R, 1536, 384, 0, True, 3, 1536),
    ("kv_pts", P, 384, 768, 0, False, 3, 384),
    ("q_pts", P, 384, 384, 0, False, 3, 384),
    ("out_pts", P, 384, 384, 0, True, 3, 384),
    ("fc1_pts", P, 384, 1536, 2, False, 3, 384),
    ("fc2_pts", P, 1536, 384, 0, True, 3, 1536),
    ("fc1_virt", V, 384, 1536, 2, False, 6, 384),
    ("fc2_virt", V, 1536, 384, 0, True, 6, 1536),
    ("q_virt", V, 384, 384, 0, False, 12, 384),
]


MODES = os.environ.get("MODES", "f32,f16x3,sh,sh2sh").split(",")
if os.environ.get("SHAPES"):
    SHAPES = [s_ for s_ in SHAPES if s_[0] in os.environ["SHAPES"].split(",")]


def main():
    dev = torch.device("cuda:0")
    rounds = int(os.environ.get("ROUNDS", "5"))
    res = {}
    bufs = {}
    for name, M, K, N, act, resid, cnt, kv in SHAPES:
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        r = torch.randn(M, N, device=dev) if resid else None
        bufs[name] = (a, w, b, out, r, ops.pack_weight(w), ops.split_rows(a))
    for rd in range(rounds + 1):
        for name, M, K, N, act, resid, cnt, kv in SHAPES:
            a, w, b, out, r, wp, ash = bufs[name]
            for mode in MODES:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if mode == "f32":
                    ops.gemm(a, w, bias=b, act=act, resid=r, out=out)
                elif mode == "f16x3":
                    ops.gemm(a, w, bias=b, act=act, resid=r, out=out, packed=wp)
                elif mode == "sh":
                    ops.gemm(ash, w, bias=b, act=act, resid=r, out=out, packed=wp)
                else:  # sh2sh: SH in, SH out (no residual)
                    ops.gemm(ash, w, bias=b, act=act, packed=wp, out_split=True)
                e1.record()
                e1.synchronize()
                if rd > 0:
                    res.setdefault((name, mode), []).append(e0.elapsed_time(e1))
    for mode in MODES:
        tot_t = tot_f = 0.0
        print(f"--- back end {mode}, CTK_GEMM_TILE={os.environ.get('CTK_GEMM_TILE', '0')} (TF/s = algorithmic f32-equivalent flops / time)")
        print(f"{'shape':10s} {'M':>8s} {'K':>6s} {'N':>6s} {'ms(med)':>9s} {'TF/s':>8s} {'x/iter':>6s}")
        for name, M, K, N, act, resid, cnt, kv in SHAPES:
            ts = sorted(res[(name, mode)])
            med = ts[len(ts) // 2]
            fl = 2.0 * M * N * kv
            print(f"{name:10s} {M:8d} {K:6d} {N:6d} {med:9.3f} {fl / med / 1e9:8.1f} {cnt:6d}")
            tot_t += med * cnt
            tot_f += fl * cnt
        print(f"aggregate per iteration: {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TF/s")


if __name__ == "__main__":
    main()
