#!/usr/bin/env python
"""Is the correlation sampler power/clock limited?  Runs corr_volume_sh in a long loop for several kernel
variants (CTK_CORR / CTK_CORR_DBG bisection bits), sampling rocm-smi clocks/power from a side thread."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
S, N, H0, W0 = 16, 6400, 96, 128
g = torch.Generator().manual_seed(0)
f0 = torch.randn(S, H0, W0, 128, generator=g).to(dev)
f0 = (f0 / f0.norm(dim=-1, keepdim=True)).contiguous()
pyr = ops.build_pyramid(f0)
ys, xs = torch.meshgrid(torch.linspace(2, H0 - 3, 80), torch.linspace(2, W0 - 3, 80), indexing="ij")
q = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1)
coords = (q[None] + torch.arange(S)[:, None, None] * torch.tensor([0.13, 0.07]) + 0.3 * torch.rand(S, N, 2, generator=g)).contiguous().to(dev)
sup = [ops.sample_support(pyr[l], torch.zeros(N, device=dev), (coords[0] / 2 ** l).contiguous()) for l in range(4)]
win = ops.Window(pyr, sup, coords, torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev), (W0, H0), iters=1)
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.split(":")[-1].strip() for l in out.splitlines() if ("sclk" in l or "Power" in l)]
            samples.append(" | ".join(keep))
        except Exception as e:  # noqa: BLE001
            samples.append(f"rocm-smi failed: {e}")
        time.sleep(0.25)


ops.corr_volume_sh(win)
torch.cuda.synchronize()
for tag, env in (("v2", {"CTK_CORR": "2"}), ("v2_neither", {"CTK_CORR": "2", "CTK_CORR_DBG": "3"}),
                 ("v2_nostore", {"CTK_CORR": "2", "CTK_CORR_DBG": "1"}), ("v1", {})):
    for k in ("CTK_CORR", "CTK_CORR_DBG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for _ in range(3):
        ops.corr_volume_sh(win)
    torch.cuda.synchronize()
    samples.clear()
    stop = False
    th = threading.Thread(target=sampler)
    th.start()
    reps = 500
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.corr_volume_sh(win)
    e1.record()
    e1.synchronize()
    stop = True
    th.join()
    print(f"{tag}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us/launch")
    for s_ in samples[1:6]:
        print("   ", s_)
