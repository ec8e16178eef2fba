"""The reference CoTracker2 at BASELINE scale (512x512, window 8, 6 iterations) with damped synthetic weights: its own 8-vs-3-thread
spread per (head_scale, updater_scale, residual_scale) setting.  Build container only (imports /root/reference).
    python tools/explore_v2_chaos.py 24 20 0.25,0.1,1.0 0.05,0.1,1.0 0.02,0.1,1.0
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "/root/reference")
import numpy as np, torch
from cotracker.predictor import CoTrackerPredictor
from cotracker_amd.weights import fill_synthetic_
from cotracker_amd.synthetic import synthetic_video

def run(T, G, hs, upd, threads, res=1.0):
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    video = synthetic_video(T, 512, 512, seed=1234)
    p = CoTrackerPredictor(checkpoint=None, v2=True, window_len=8)
    fill_synthetic_(p.model, seed=0, head_scale=hs)
    with torch.no_grad():
        p.model.track_feat_updater[0].weight.mul_(upd); p.model.track_feat_updater[0].bias.mul_(upd)
        if res != 1.0:
            for blk in list(p.model.updateformer.time_blocks) + list(p.model.updateformer.space_virtual_blocks) + list(p.model.updateformer.space_point2virtual_blocks) + list(p.model.updateformer.space_virtual2point_blocks):
                for lin in (blk.attn.to_out if hasattr(blk, "attn") else blk.cross_attn.to_out, blk.mlp.fc2):
                    lin.weight.mul_(res); lin.bias.mul_(res)
    cap = {}
    mf = p.model.forward
    def tap(*a, **k):
        out = mf(*a, **k); cap["c"] = out[0].clone(); cap["v"] = out[1].clone(); return out
    p.model.forward = tap
    t0 = time.time()
    with torch.no_grad():
        tr, vi = p(video, grid_size=G)
    return cap["c"][0].numpy(), cap["v"][0].numpy(), time.time() - t0

T, G = int(sys.argv[1]), int(sys.argv[2])
for cfg in sys.argv[3:]:
    hs, upd, res = map(float, cfg.split(","))
    a, av, dt = run(T, G, hs, upd, 8, res)
    b, bv, _ = run(T, G, hs, upd, 3, res)
    lg = lambda p: np.log(p.astype(np.float64) / (1 - p.astype(np.float64)))
    motion = np.abs(a - a[:1]).max()
    print(f"cfg hs={hs} upd={upd} res={res}: {dt:.0f}s spread coords {np.abs(a-b).max():.2e} vis-logit {np.abs(lg(av)-lg(bv)).max():.2e} motion {motion:.2f}px med {np.median(np.abs(a-a[:1])):.2f} vis range {lg(av).min():.2f}..{lg(av).max():.2f}", flush=True)
