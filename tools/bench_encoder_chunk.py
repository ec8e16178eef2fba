#!/usr/bin/env python
"""Encoder time for 120 frames of 384x512 as a function of the frames-per-call chunk (per-frame network: results are
identical up to MIOpen's algorithm choice per batch size; smaller chunks keep the activations in the 256 MB Infinity Cache)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd.model import CoTrackerThreeOnline  # noqa: E402
from cotracker_amd.weights import fill_synthetic_  # noqa: E402

dev = torch.device("cuda:0")
m = CoTrackerThreeOnline(window_len=16).eval()
fill_synthetic_(m, seed=0)
m = m.to(dev)
m.encoder_chunk = 10 ** 9  # lift the model's internal cap: this tool sweeps the chunk itself
x = torch.rand(120, 3, 384, 512, device=dev) * 255
ref = None
for chunk in (200, 60, 40, 24, 16, 12, 8, 4):
    with torch.no_grad():
        y = m._encode(x, chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = m._encode(x, chunk)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    if ref is None:
        ref = y.clone()
    print(f"chunk {chunk:4d}: {ms:7.1f} ms for 120 frames   max |diff| vs chunk 200: {float((y - ref).abs().max()):.2e}", flush=True)
