#!/usr/bin/env python
"""Micro-benchmark of the correlation sampler alone on a C3-sized window (S=16, N=6400, 96x128 4-level pyramid).
Versions 1, 2, 3 (CTK_CORR) and their bisection bits (CTK_CORR_DBG).  Env: REPS, NPTS, ONLY."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, N, H0, W0 = 16, int(os.environ.get("NPTS", "6400")), 96, 128
g = torch.Generator().manual_seed(0)
f0 = torch.randn(S, H0, W0, 128, generator=g).to(dev)
f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
pyr = ops.build_pyramid(f0)
G = int(round(N ** 0.5))
ys, xs = torch.meshgrid(torch.linspace(2, H0 - 3, G), torch.linspace(2, W0 - 3, G), indexing="ij")
q = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1)[:N]
coords = (q[None] + torch.arange(S)[:, None, None] * torch.tensor([0.13, 0.07]) + 0.3 * torch.rand(S, N, 2, generator=g)).contiguous().to(dev)
sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (coords[0] / 2 ** l).contiguous()) for l in range(4)]
win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
reps = int(os.environ.get("REPS", "10"))
V2 = {"CTK_CORR": "2"}
V3 = {"CTK_CORR": "3", "CTK_CORR_MAP": "0"}  # (the bisection rows below were taken with the point-major dealing)
for tag, env in (("warmup", {}), ("v1", {"CTK_CORR": "1"}), ("v3", V3), ("v3_nostore", dict(V3, CTK_CORR_DBG="1")), ("v3_noloads", dict(V3, CTK_CORR_DBG="2")),
                 ("v3_neither", dict(V3, CTK_CORR_DBG="3")), ("v3_nomfma", dict(V3, CTK_CORR_DBG="16")), ("v3_noblend", dict(V3, CTK_CORR_DBG="32")),
                 ("v3_nomfma_noblend", dict(V3, CTK_CORR_DBG="48")), ("v3_nostorephase", dict(V3, CTK_CORR_DBG="64")),
                 ("v3_loads_only", dict(V3, CTK_CORR_DBG="112")), ("v3_again", V3), ("v3_map1", dict(V3, CTK_CORR_MAP="1")), ("v3_map2", dict(V3, CTK_CORR_MAP="2")),
                 ("v3_map1_noloads", dict(V3, CTK_CORR_MAP="1", CTK_CORR_DBG="2")), ("v3_map0_b", V3), ("default", {}), ("v3_map1_b", dict(V3, CTK_CORR_MAP="1")), ("v3_map2_b", dict(V3, CTK_CORR_MAP="2")), ("v3_map3", dict(V3, CTK_CORR_MAP="3")), ("v3_map4", dict(V3, CTK_CORR_MAP="4")), ("v3_map3_b", dict(V3, CTK_CORR_MAP="3")),
                 ("v3_map4_b", dict(V3, CTK_CORR_MAP="4")), ("v3_map3_nostore", dict(V3, CTK_CORR_MAP="3", CTK_CORR_DBG="1")), ("v3_map3_noloads", dict(V3, CTK_CORR_MAP="3", CTK_CORR_DBG="2")), ("v1_b", {"CTK_CORR": "1"}), ("v2", V2), ("v2_nostore", dict(V2, CTK_CORR_DBG="1")),
                 ("v2_noloads", dict(V2, CTK_CORR_DBG="2")), ("v2_neither", dict(V2, CTK_CORR_DBG="3")),
                 ("v2_nt", dict(V2, CTK_CORR_DBG="4")), ("v2_nopf", dict(V2, CTK_CORR_DBG="8")), ("v1_again", {"CTK_CORR": "1"})):
    if os.environ.get("ONLY") and tag not in os.environ["ONLY"].split(","):
        continue
    for k in ("CTK_CORR", "CTK_CORR_DBG", "CTK_CORR_MAP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    out = ops.corr_volume_sh(win)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ops.corr_volume_sh(win)
    e1.record()
    e1.synchronize()
    print(f"{tag:12s} {e0.elapsed_time(e1) / reps * 1e3:9.1f} us per launch (incl. pyramid split + 4 GB torch.empty)")
