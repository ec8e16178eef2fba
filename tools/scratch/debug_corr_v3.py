#!/usr/bin/env python
"""Dev-only: hunt the intermittent mismatch of sampler version 3 (stress coordinates), print where it differs from version 1.
Every repetition uses fresh allocations (cold caches / TLB) and runs other kernels in between (stale LDS of another kernel)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cotracker_amd import ops
dev = torch.device("cuda:0")
S = int(os.environ.get("S", "20"))
VER = os.environ.get("VER", "3")
H0, W0, N = 48, 64, 90
bad = 0
keep = []
for it in range(int(os.environ.get("REPS", "60"))):
    r = np.random.RandomState(1000 + it)
    f0 = torch.from_numpy(r.standard_normal((S, H0, W0, 128)).astype(np.float32)).to(dev)
    f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
    pyr = ops.build_pyramid(f0)
    c = r.uniform(-8, 8, size=(S, N, 2)) + r.uniform(0, 1, size=(S, N, 2)) * np.array([W0 - 1, H0 - 1])
    c[:, 0:20] = np.round(c[:, 0:20]); c[:, 20:30] = np.round(c[:, 20:30]) + 0.5; c[:, 30:40] = np.round(c[:, 30:40] / 8) * 8
    c[:, 40] = [0.0, 0.0]; c[:, 41] = [W0 - 1, H0 - 1]; c[:, 42] = [-50.0, 1000.0]; c[:, 43] = [W0 + 2.25, -3.5]
    coords = torch.from_numpy(c.astype(np.float32)).to(dev)
    qc = coords[0].contiguous()
    sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (qc / 2 ** l).contiguous()) for l in range(4)]
    win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
    keep.append(torch.empty(64 << 20, device=dev))  # move later allocations to fresh addresses
    os.environ["CTK_CORR"] = VER
    ref32 = ops.corr_volume(win)  # the exact-f32 sampler (as in the test; also dirties LDS)
    got = [ops.unsplit(v).clone() for v in ops.corr_volume_sh(win)]
    os.environ["CTK_CORR"] = "1"
    ref = [ops.unsplit(v).clone() for v in ops.corr_volume_sh(win)]
    for l in range(4):
        d = (got[l] - ref[l]).abs()
        d = torch.nan_to_num(d, nan=1e9)
        if float(d.max()) > 1e-4:
            bad += 1
            idx = (d > 1e-4).nonzero()
            rows = idx[:, 0].unique().tolist()
            cols = idx[:, 1]
            print(f"it {it} level {l}: {idx.shape[0]} elements, (n,t) {[(x // S, x % S) for x in rows][:10]} cols {int(cols.min())}..{int(cols.max())} "
                  f"p {sorted(set((cols // 49).tolist()))[:10]} q {sorted(set((cols % 49).tolist()))[:10]} maxdiff {float(d.max()):.4g} vs f32 {float((got[l]-ref32[l]).abs().max()):.3g}", flush=True)
print("bad (launch, level) pairs:", bad)
