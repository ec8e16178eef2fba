"""PP on/off A/B of the dense-mode predictor run of tests/test_sharding.py (single process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import _lib
from cotracker_amd.predictor import CoTrackerPredictor
from cotracker_amd.synthetic import synthetic_video
from cotracker_amd.weights import fill_synthetic_
lib = _lib.load()
dev = torch.device("cuda:0")
p = CoTrackerPredictor(checkpoint=None, offline=False, window_len=8)
fill_synthetic_(p.model, seed=0)
p = p.to(dev)
video = synthetic_video(12, 96, 160, seed=3).to(dev)
outs = {}
modes = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 1, 0, 1]
for mode in modes:
    lib.ctk_gemm_pp_mode(mode)
    t, v = p(video[:, :8])
    torch.cuda.synchronize()
    key = (mode, len([k for k in outs if k[0] == mode]))
    outs[key] = (t.clone(), v.clone())
    print(mode, t.shape, float(t.abs().max()), bool(torch.isfinite(t).all()))
keys = list(outs)
for b in keys[1:]:
    a = keys[0]
    print(a, b, "max |dt| =", float((outs[a][0] - outs[b][0]).abs().max()), "vis flips", int((outs[a][1] != outs[b][1]).sum()))
print("range_fallbacks", getattr(p.model, "range_fallbacks", None))
