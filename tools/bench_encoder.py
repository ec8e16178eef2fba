#!/usr/bin/env python
"""Encoder-only timing (BasicEncoder on MIOpen, T=120 frames of 384x512): default vs cudnn.benchmark vs channels_last."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd.encoder import BasicEncoder  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = BasicEncoder().to(dev).eval()
x = torch.randn(120, 3, 384, 512, device=dev)


def run(tag, fn, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        y = fn()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
        for _ in range(reps):
            y = fn()
    torch.cuda.synchronize()
    print(f"{tag:28s} first call {first * 1e3:9.1f} ms, steady {(time.perf_counter() - t0) / reps * 1e3:8.1f} ms  out {tuple(y.shape)}", flush=True)
    return y


y0 = run("default", lambda: enc(x))
xc = x.contiguous(memory_format=torch.channels_last)
encc = BasicEncoder().to(dev).eval()
encc.load_state_dict(enc.state_dict())
encc = encc.to(memory_format=torch.channels_last)
y1 = run("channels_last", lambda: encc(xc))
print("   max diff vs default", float((y1 - y0).abs().max()))
torch.backends.cudnn.benchmark = True
y2 = run("cudnn.benchmark", lambda: enc(x))
print("   max diff vs default", float((y2 - y0).abs().max()))
y3 = run("benchmark+channels_last", lambda: encc(xc))
