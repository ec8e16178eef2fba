#!/usr/bin/env python
"""Dev probe: is the streaming model bit-reproducible run-to-run (direct launches) and graph vs direct?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd.model import CoTrackerThreeOnline
from cotracker_amd.weights import fill_synthetic_

dev = torch.device("cuda:0")
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model_online.npz"))
for prec in ("f16x3", "f32"):
    m = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=8, model_resolution=(64, 96)).eval()
    m.precision = prec
    fill_synthetic_(m, seed=1)
    m = m.to(dev)
    video, q = torch.from_numpy(g["on_video"]).to(dev), torch.from_numpy(g["on_queries"]).to(dev)

    def run(use_graph):
        m.hip_graph = use_graph
        m.init_video_online_processing()
        per_call = []
        for ind in range(0, video.shape[1] - 4, 4):
            cs, vs, fs, _ = m(video[:, ind:ind + 8], q, iters=4, is_online=True)
            per_call.append(cs.clone())
        return per_call

    x = 2 * (video[0, :8].float() / 255.0) - 1.0
    f1 = m.fnet(x).clone(); f2 = m.fnet(x).clone(); f3 = m.fnet(x).clone()
    print(prec, "fnet run-to-run:", float((f1 - f2).abs().max()), float((f2 - f3).abs().max()))
    runs = {"A1": run(False), "A2": run(False), "G1": run(True), "G2": run(True), "A3": run(False)}
    for a, b in (("A1", "A2"), ("A2", "A3"), ("A2", "G1"), ("G1", "G2"), ("A3", "G2")):
        print(prec, a, b, [float((x_ - y_).abs().max()) for x_, y_ in zip(runs[a], runs[b])])
