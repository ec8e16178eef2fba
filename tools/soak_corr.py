#!/usr/bin/env python
"""Soak test of the correlation sampler versions on the stress window of tests/test_gpu_parity.py (integer / border / out-of-range
coordinates, S in {1, 2, 5, 20}): every repetition rebuilds the inputs, runs the exact-f32 sampler as the reference, then every
version in VERS (interleaved: each leaves its own LDS contents and timing behind), and reports WHERE a version differs
(frame index inside its 16-frame chunk, level, wave / lane / element of the blend thread that wrote it).

Written in round 5 to corner an intermittent mismatch of version 3 (1 launch in ~30, only in the chunk's second frame, only
lanes 48..63 of a wave, only the first register of a ds_write2_b32): a v_pk_fma_f32 with op_sel:[0,1,0] right in front of the
LDS write.  profiles/r05_sampler_v3_pk_hazard.txt has the story.  Env: REPS (3), VERS ("1,3"), QUIET."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
bad = 0
summary = {}
launches = 0
for rep in range(int(os.environ.get("REPS", "3"))):
    for S in (1, 2, 5, 20):
        for version in os.environ.get("VERS", "1,3").split(","):
            _lib.load().ctk_set_option(_lib.OPT_CORR_VERSION, int(version))
            r = np.random.RandomState(S)
            H0, W0, N = 48, 64, 90
            f0 = torch.from_numpy(r.standard_normal((S, H0, W0, 128)).astype(np.float32)).to(dev)
            f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
            pyr = ops.build_pyramid(f0)
            c = r.uniform(-6, 1, size=(S, N, 2))
            c = r.uniform(-8, 8, size=(S, N, 2)) + r.uniform(0, 1, size=(S, N, 2)) * np.array([W0 - 1, H0 - 1])
            c[:, 0:20] = np.round(c[:, 0:20])
            c[:, 20:30] = np.round(c[:, 20:30]) + 0.5
            c[:, 30:40] = np.round(c[:, 30:40] / 8) * 8
            c[:, 40] = [0.0, 0.0]
            c[:, 41] = [W0 - 1, H0 - 1]
            c[:, 42] = [-50.0, 1000.0]
            c[:, 43] = [W0 + 2.25, -3.5]
            coords = torch.from_numpy(c.astype(np.float32)).to(dev)
            qc = coords[0].contiguous()
            sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (qc / 2 ** l).contiguous()) for l in range(4)]
            win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
            ref = ops.corr_volume(win)
            got = ops.corr_volume_sh(win)
            launches += 1
            for l in range(4):
                g = ops.unsplit(got[l])
                d = torch.nan_to_num((g - ref[l]).abs(), nan=1e9)
                if float(d.max()) < 3e-6:
                    continue
                bad += 1
                idx = (d >= 3e-6).nonzero()
                for x in idx.tolist():
                    col = x[1]
                    pp, qq = col // 49, col % 49
                    bp = (pp % 7) * 7 + pp // 7  # the blend thread of versions 1 / 3: tap p dealt x-fastest, 12 q per thread
                    tid = bp * 5 + min(qq // 12, 4)
                    key = (version, "frame_in_chunk", (x[0] % S) % 16, "level", l, "wave", tid // 64, "lane", tid % 64, "element", (qq % 12) if qq < 48 else 0)
                    summary[key] = summary.get(key, 0) + 1
                if not os.environ.get("QUIET"):
                    rows = idx[:, 0].unique().tolist()
                    print(f"rep {rep} S {S} version {version} level {l}: {idx.shape[0]} elements differ, (n, t) {[(x // S, x % S) for x in rows][:8]}, max {float(d.max()):.3g}", flush=True)
print(f"launches {launches}, (launch, level) pairs with a mismatch >= 3e-6: {bad}")
for field in (2, 4, 6, 8, 10):
    cnt = collections.Counter()
    for k, v in summary.items():
        cnt[(k[0], k[field - 1], k[field])] += v
    if cnt:
        print("  ", sorted(cnt.items()))
sys.exit(1 if bad else 0)
