#!/usr/bin/env python
"""Can the CNN encoder (MIOpen, ~100 ms of a 1.5 s C3 step) run in half precision?  For fp32 / fp16 / bf16 autocast:
encoder time on the C3 video (120 frames of 384x512) and the BASELINE configs[1] end-to-end error against the
unmodified reference's CPU run (tests/golden/scale_c2.npz).  The bar is 1e-3 px / 1e-4 logit."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cotracker_amd.predictor import CoTrackerPredictor  # noqa: E402
from cotracker_amd.synthetic import synthetic_video  # noqa: E402
from cotracker_amd.weights import fill_synthetic_  # noqa: E402

dev = torch.device("cuda:0")
p = CoTrackerPredictor(checkpoint=None, offline=True, window_len=60)
fill_synthetic_(p.model, seed=0)
p = p.to(dev)
video = synthetic_video(48, 256, 256, seed=1234).to(dev)
big = torch.rand(120, 3, 384, 512, device=dev) * 255
for name, dt in (("fp32", torch.float32), ("fp16", torch.float16), ("bf16", torch.bfloat16)):
    p.model.encoder_dtype = dt
    cap = {}
    fwd = type(p.model).forward

    def tap(*a, **k):
        out = fwd(p.model, *a, **k)
        cap["coords"] = out[0].clone()
        return out

    p.model.forward = tap
    p(video, grid_size=20)
    vl, cl = p.model.last_logits
    par = bench.golden_parity("c2", cap["coords"][0], vl[0], cl[0])
    del p.model.forward
    with torch.no_grad():
        p.model._encode(big, 200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            p.model._encode(big, 200)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"encoder": name, "encode_120_frames_ms": round(ms, 1), "c2_coords_px": par["coords_px"],
                      "c2_vis_logit": par["vis_logit"], "c2_conf_logit": par["conf_logit"]}), flush=True)
