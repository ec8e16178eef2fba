#!/usr/bin/env python
"""Build-container measurement behind bench.py's cpu_baseline: the UNMODIFIED reference and oracle/torch_port.py on the
same workload, same weights, same thread count -- wall time and agreement of the outputs.

    MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=68719476736 MALLOC_TOP_PAD_=1073741824 \
        python tools/time_cpu_reference.py [--frames 24] [--grid 20] [--threads 8] > profiles/r02_cpu_reference_vs_port.txt

(needs /root/reference; the malloc variables are what bench.py sets for its own CPU leg, see oracle/torch_port.py)"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from cotracker.predictor import CoTrackerPredictor  # noqa: E402  (the reference)

from cotracker_amd.synthetic import synthetic_video  # noqa: E402
from cotracker_amd.weights import fill_synthetic_  # noqa: E402
from oracle import torch_port as TP  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--grid", type=int, default=20)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--threads", type=int, default=8)
a = ap.parse_args()
torch.set_num_threads(a.threads)
video = synthetic_video(a.frames, a.size, a.size, seed=1234)
print(f"workload: sliding window S=16, {a.size}x{a.size} video, T={a.frames}, N={a.grid ** 2}, 6 iterations, {a.threads} threads of "
      f"{os.cpu_count()} logical CPUs, torch {torch.__version__}, malloc tuned: "
      f"{all(os.environ.get(k) == v for k, v in TP.MALLOC_ENV.items())}")
rows = {}
for kind, offline, wl in (("sliding", False, 16), ("offline", True, 60)):
    torch.manual_seed(0)
    p = CoTrackerPredictor(checkpoint=None, offline=offline, window_len=wl)
    fill_synthetic_(p.model, seed=0)
    with torch.no_grad():
        t0 = time.time()
        ref_tracks, ref_vis = p(video, grid_size=a.grid)
        t_ref = time.time() - t0
    fnet, params = TP.build(offline, wl)
    t0 = time.time()
    tracks, vis, *_ = TP.predictor_forward(fnet, params, video, a.grid, wl, offline)
    t_port = time.time() - t0
    n = a.grid ** 2 * a.frames
    print(f"{kind:8s} reference {t_ref:7.1f} s = {n / t_ref:7.1f} pf/s | torch_port {t_port:7.1f} s = {n / t_port:7.1f} pf/s | "
          f"ratio port/reference {t_port / t_ref:.2f} | max |tracks diff| {float((tracks - ref_tracks).abs().max()):.2e} px, "
          f"visibility flips {int((vis != ref_vis).sum())}")
