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
if os.environ.get("WINDOW") == "c3":
    # Production-shape soak: the C3 window -- S = 16 frames, N = 6400 grid points on the 96 x 128 level-0 map,
    # 25 600 workgroups per launch -- LAUNCHES (2000) launches of the DEFAULT sampler, each compared bit for bit with the first
    # (and the first with the exact-f32 sampler to 3e-6); other kernels (the compare, a pyramid rebuild every 50 launches) run in
    # between, as in a real step.  Exit code 1 on any mismatch.
    S, H0, W0, G = 16, 96, 128, 80
    N = G * G
    n_launch = int(os.environ.get("LAUNCHES", "2000"))
    r = np.random.RandomState(7)
    f0 = torch.from_numpy(r.standard_normal((S, H0, W0, 128)).astype(np.float32)).to(dev)
    f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
    pyr = ops.build_pyramid(f0)
    ys, xs = np.meshgrid(np.linspace(3, H0 - 4, G), np.linspace(3, W0 - 4, G), indexing="ij")
    q = np.stack([xs.reshape(-1), ys.reshape(-1)], -1).astype(np.float32)
    c = q[None] + np.arange(S)[:, None, None] * np.array([0.13, 0.07], np.float32) + 0.3 * r.uniform(size=(S, N, 2)).astype(np.float32)
    c[:, ::97] = np.round(c[:, ::97])  # some integer coordinates: 9-wide footprints
    coords = torch.from_numpy(c.astype(np.float32)).to(dev)
    sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (coords[0] / 2 ** l).contiguous()) for l in range(4)]
    win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
    first = ops.corr_volume_sh(win).clone()
    ref = ops.corr_volume(win)
    worst = max(float((ops.unsplit(first[l]) - ref[l]).abs().max()) for l in range(4))
    del ref
    mism = []
    for i in range(n_launch):
        if i % 50 == 49:
            pyr = ops.build_pyramid(f0)  # same values, fresh buffers: other kernels and allocations in between
            win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
        out = ops.corr_volume_sh(win)
        if not torch.equal(out, first):
            d = (out.view(torch.int16) != first.view(torch.int16)).nonzero()
            mism.append((i, int(d.shape[0]), d[:4].tolist()))
            print(f"launch {i}: {d.shape[0]} halves differ, first at {d[:4].tolist()}", flush=True)
        del out
    print(f"soak c3 window: S={S} N={N} ({N * 4} workgroups per launch), {n_launch} launches of the default sampler, "
          f"first launch vs exact-f32 sampler max {worst:.3g}; launches that differ bitwise from the first: {len(mism)}")
    sys.exit(1 if mism or worst >= 3e-6 else 0)
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
