#!/usr/bin/env python
"""Dev-only: the body of tests/test_gpu_parity.py::test_corr_volume_sh_stress_coordinates for every (S, version) in a fresh
process; on a mismatch say where."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cotracker_amd import ops
dev = torch.device("cuda:0")
bad = 0
summary = {}
for rep in range(int(os.environ.get("REPS", "3"))):
  for S in (1, 2, 5, 20):
    for version in os.environ.get("VERS", "1,2,3").split(","):
        os.environ["CTK_CORR"] = version
        r = np.random.RandomState(S)
        H0, W0, N = 48, 64, 90
        f0 = torch.from_numpy(r.standard_normal((S, H0, W0, 128)).astype(np.float32)).to(dev)
        f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
        pyr = ops.build_pyramid(f0)
        c = r.uniform(-6, 1, size=(S, N, 2)) * np.array([W0 + 10, H0 + 10]) * np.array([-1, -1]) * -1
        c = r.uniform(-8, 8, size=(S, N, 2)) + r.uniform(0, 1, size=(S, N, 2)) * np.array([W0 - 1, H0 - 1])
        c[:, 0:20] = np.round(c[:, 0:20]); c[:, 20:30] = np.round(c[:, 20:30]) + 0.5; c[:, 30:40] = np.round(c[:, 30:40] / 8) * 8
        c[:, 40] = [0.0, 0.0]; c[:, 41] = [W0 - 1, H0 - 1]; c[:, 42] = [-50.0, 1000.0]; c[:, 43] = [W0 + 2.25, -3.5]
        coords = torch.from_numpy(c.astype(np.float32)).to(dev)
        qc = coords[0].contiguous()
        sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (qc / 2 ** l).contiguous()) for l in range(4)]
        vis, conf = torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev)
        win = ops.Window(pyr, sup, coords, vis, conf, (W0, H0), iters=1)
        ref = ops.corr_volume(win)
        got = ops.corr_volume_sh(win)
        for l in range(4):
            g = ops.unsplit(got[l])
            d = torch.nan_to_num((g - ref[l]).abs(), nan=1e9)
            if float(d.max()) >= 3e-6:
                bad += 1
                idx = (d >= 3e-6).nonzero()
                rows = idx[:, 0].unique().tolist()
                cols = idx[:, 1]
                i0 = idx[0]
                for x in idx.tolist():
                    col = x[1]
                    pp, qq = col // 49, col % 49
                    bhx_, bwy_ = pp // 7, pp % 7
                    bp_ = bwy_ * 7 + bhx_
                    tid_ = bp_ * 5 + min(qq // 12, 4)
                    key = (version, "tl", (x[0] % S) % 16, "lvl", l, "wave", tid_ // 64, "lane", tid_ % 64, "je", (qq % 12) if qq < 48 else 0)
                    summary[key] = summary.get(key, 0) + 1
                if os.environ.get("QUIET"):
                    continue
                os.environ["CTK_CORR"] = "1"
                g1 = ops.unsplit(ops.corr_volume_sh(win)[l])
                ref2 = ops.corr_volume(win)[l]
                os.environ["CTK_CORR"] = version
                g3 = ops.unsplit(ops.corr_volume_sh(win)[l])
                print(f"   outlier check: |got-v1| {float((g - g1).abs().max()):.3g} |ref-v1| {float((ref[l] - g1).abs().max()):.3g} |ref-ref_again| {float((ref[l] - ref2).abs().max()):.3g} |got-got_again| {float((g - g3).abs().max()):.3g}")
                nn, tt = rows[0] // S, rows[0] % S
                np.savez(f"gpurun_out/v3_fail_{os.getpid()}_{rep}_{S}_{l}.npz", got=g[rows[0]].cpu().numpy(), ref=ref[l][rows[0]].cpu().numpy(), v1=g1[rows[0]].cpu().numpy(),
                         n=nn, t=tt, level=l, S=S, coords=c[:, nn], fmap=pyr[l].cpu().numpy(), support=sup[l][nn].cpu().numpy(),
                         got_all=g[nn * S:(nn + 1) * S].cpu().numpy(), ref_all=ref[l][nn * S:(nn + 1) * S].cpu().numpy())
                print(f"rep {rep} S {S} v{version} level {l}: {idx.shape[0]} elements, (n,t) {[(x // S, x % S) for x in rows][:10]} cols {int(cols.min())}..{int(cols.max())} "
                      f"p {sorted(set((cols // 49).tolist()))[:12]} q {sorted(set((cols % 49).tolist()))[:12]} maxdiff {float(d.max()):.4g}; first: got {float(g[i0[0], i0[1]]):.5f} ref {float(ref[l][i0[0], i0[1]]):.5f} "
                      f"coords {c[rows[0] % S, rows[0] // S].tolist()}", flush=True)
print("bad:", bad)
import collections
for field in (2, 4, 6, 8, 10):
    c = collections.Counter()
    for k, v in summary.items():
        c[(k[0], k[field - 1], k[field])] += v
    print("  ", sorted(c.items()))
