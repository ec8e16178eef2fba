#!/usr/bin/env python
"""Dev-only: fast reproducer.  Fixed inputs (the stress test's, S=20); every repetition launches version PRE (dirties LDS), then
version 3 with CTK_CORR_DBG=VAR, and compares with a version-1 result computed once."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cotracker_amd import ops
dev = torch.device("cuda:0")
S = int(os.environ.get("S", "20"))
r = np.random.RandomState(S)
H0, W0, N = 48, 64, 90
f0 = torch.from_numpy(r.standard_normal((S, H0, W0, 128)).astype(np.float32)).to(dev)
f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
pyr = ops.build_pyramid(f0)
c = r.uniform(-6, 1, size=(S, N, 2))
c = r.uniform(-8, 8, size=(S, N, 2)) + r.uniform(0, 1, size=(S, N, 2)) * np.array([W0 - 1, H0 - 1])
c[:, 0:20] = np.round(c[:, 0:20]); c[:, 20:30] = np.round(c[:, 20:30]) + 0.5; c[:, 30:40] = np.round(c[:, 30:40] / 8) * 8
c[:, 40] = [0.0, 0.0]; c[:, 41] = [W0 - 1, H0 - 1]; c[:, 42] = [-50.0, 1000.0]; c[:, 43] = [W0 + 2.25, -3.5]
coords = torch.from_numpy(c.astype(np.float32)).to(dev)
qc = coords[0].contiguous()
sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (qc / 2 ** l).contiguous()) for l in range(4)]
win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
os.environ.pop("CTK_CORR_DBG", None)
os.environ["CTK_CORR"] = "1"
ref = [ops.unsplit(v).clone() for v in ops.corr_volume_sh(win)]
PRE = os.environ.get("PRE", "2")
for VAR in os.environ.get("VARS", "0").split(","):
    bad = 0
    for it in range(int(os.environ.get("REPS", "300"))):
        os.environ.pop("CTK_CORR_DBG", None)
        if PRE == "f32":
            ops.corr_volume(win)
        elif PRE != "none":
            os.environ["CTK_CORR"] = PRE
            ops.corr_volume_sh(win)
        os.environ["CTK_CORR"] = "3"
        if VAR != "0":
            os.environ["CTK_CORR_DBG"] = VAR
        got = ops.corr_volume_sh(win)
        for l in range(4):
            d = torch.nan_to_num((ops.unsplit(got[l]) - ref[l]).abs(), nan=1e9)
            if float(d.max()) > 1e-5:
                bad += 1
                if bad <= 6:
                    idx = (d > 1e-5).nonzero()
                    rows = idx[:, 0].unique().tolist()
                    cols = idx[:, 1]
                    print(f"  VAR {VAR} it {it} level {l}: {idx.shape[0]} el, (n,t) {[(x // S, x % S) for x in rows][:6]} p {sorted(set((cols // 49).tolist()))[:8]} q {sorted(set((cols % 49).tolist()))[:8]} maxdiff {float(d.max()):.3g}", flush=True)
    print(f"PRE {PRE} VAR {VAR}: bad (launch, level) pairs {bad}", flush=True)
