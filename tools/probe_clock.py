#!/usr/bin/env python
"""Is the split-half GEMM power/clock limited?  Runs one shape in a long loop with (a) random operands and
(b) zero operands, sampling rocm-smi clocks/power from a side thread, and reports TF/s for both."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N = 409600, 2432, 384
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l)]
            samples.append(" | ".join(keep))
        except Exception as e:  # noqa: BLE001
            samples.append(f"rocm-smi failed: {e}")
        time.sleep(0.3)


for label, fill in (("random", None), ("zeros", 0.0), ("random2", None)):
    a = torch.randn(M, K, device=dev) if fill is None else torch.full((M, K), fill, device=dev)
    w = (torch.randn(N, K, device=dev) / K ** 0.5) if fill is None else torch.full((N, K), fill, device=dev)
    wp = ops.pack_weight(w)
    ash = ops.split_rows(a)
    out = torch.empty(M, N, device=dev)
    del a
    for _ in range(3):
        ops.gemm(ash, w, packed=wp, out=out)
    torch.cuda.synchronize()
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    reps = 600
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(ash, w, packed=wp, out=out)
    e1.record()
    e1.synchronize()
    stop = True
    th.join()
    ms = e0.elapsed_time(e1) / reps
    print(f"{label}: {ms:.3f} ms/launch, {2.0 * M * N * 2401 / ms / 1e9:.1f} TF/s f32-equivalent ({3 * 2.0 * M * N * K / ms / 1e9:.0f} TF/s of f16 MFMA issued)")
    for s_ in samples[:6]:
        print("   ", s_)
    del ash, out
