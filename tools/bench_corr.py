#!/usr/bin/env python
"""Micro-benchmark of the correlation sampler alone on a C3-sized window (S=16, N=6400, 96x128 4-level pyramid).
Versions 1 and 3 (CTK_OPT_CORR_VERSION), the workgroup dealings (CTK_OPT_CORR_MAP) and, in dev builds, the bisection bits (CTK_CORR_DBG).  Env: REPS, NPTS, ONLY."""
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
from cotracker_amd import _lib  # noqa: E402

# rows: (tag, CTK_OPT_CORR_VERSION, CTK_OPT_CORR_MAP).  The bisection bits (CTK_CORR_DBG: no stores / no loads / no MFMAs ...) exist in
# the DEV build only and are read once per process: `CTK_LIB_PATH=co-tracker_amd/libctk_hip_dev.so CTK_CORR_DBG=<bits> ONLY=v3 ...`
for tag, ver, mp in (("warmup", 3, 3), ("v1", 1, 0), ("v3", 3, 0), ("v3_map1", 3, 1), ("v3_map2", 3, 2), ("v3_map3", 3, 3), ("v3_map4", 3, 4),
                     ("default", None, None), ("v1_b", 1, 0), ("v3_map3_b", 3, 3)):
    if os.environ.get("ONLY") and tag not in os.environ["ONLY"].split(","):
        continue
    lib = _lib.load()
    lib.ctk_set_option(_lib.OPT_CORR_VERSION, 3 if ver is None else ver)
    lib.ctk_set_option(_lib.OPT_CORR_MAP, 3 if mp is None else mp)
    out = ops.corr_volume_sh(win)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ops.corr_volume_sh(win)
    e1.record()
    e1.synchronize()
    print(f"{tag:12s} {e0.elapsed_time(e1) / reps * 1e3:9.1f} us per launch (incl. pyramid split + 4 GB torch.empty)")
