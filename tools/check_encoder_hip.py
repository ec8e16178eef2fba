"""HIP encoder (co-tracker_amd/encoder_hip.py) against the torch BasicEncoder (MIOpen fp32) on the GPU: stage by stage, then
the normalised features, then timing.  usage: check_encoder_hip.py [frames] [H] [W]"""
import os, sys, time, torch
import torch.nn.functional as Fn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import ops
from cotracker_amd.encoder_hip import HipEncoder
from cotracker_amd.model import CoTrackerThreeOnline
from cotracker_amd.synthetic import synthetic_video
from cotracker_amd.weights import fill_synthetic_

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = int(sys.argv[2]) if len(sys.argv) > 2 else 384
W = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda:0")
m = CoTrackerThreeOnline(window_len=16).eval()
fill_synthetic_(m, seed=0)
m = m.to(dev)
video = synthetic_video(nf, H, W, seed=1234)[0].to(dev).float().contiguous()  # [T,3,H,W] 0..255
enc = HipEncoder(m.fnet, dev)
enc.trace = {}
out = enc(video)
torch.cuda.synchronize()
f = m.fnet
with torch.no_grad():
    x = 2 * (video / 255.0) - 1.0
    ref = {}
    c1 = f.conv1(x); ref["conv1"] = c1
    h = Fn.relu(Fn.instance_norm(c1, eps=1e-5))
    feats = []
    for i, layer in enumerate((f.layer1, f.layer2, f.layer3, f.layer4)):
        h = layer(h); ref[f"layer{i + 1}"] = h
        feats.append(Fn.interpolate(h, (H // 4, W // 4), mode="bilinear", align_corners=True))
    cat = torch.cat(feats, 1); ref["fused"] = cat
    c2 = f.conv2(cat); ref["conv2"] = c2
    c3 = f.conv3(Fn.relu(Fn.instance_norm(c2, eps=1e-5))); ref["conv3"] = c3
    fin = c3.permute(0, 2, 3, 1)
    fin = fin / torch.sqrt(torch.maximum((fin * fin).sum(-1, keepdim=True), torch.tensor(1e-12, device=dev)))
for k in ("conv1", "layer1", "layer2", "layer3", "layer4", "fused", "conv2", "conv3"):
    a, b = enc.trace[k], ref[k].permute(0, 2, 3, 1)
    d = (a.double() - b.double()).abs()
    print(f"{k:8s} shape {tuple(a.shape)} max|ref| {float(b.abs().max()):9.3f} max err {float(d.max()):.3e} mean err {float(d.mean()):.3e}")
d = (out.double() - fin.double()).abs()
print(f"features max err {float(d.max()):.3e} mean {float(d.mean()):.3e} (|f| <= 1)")
ours = m._encode(video, 16)
print("model._encode (torch) vs ref", float((ours - fin).abs().max()))
enc.trace = None
for name, fn in (("hip", lambda: enc(video)), ("torch", lambda: m._encode(video, 16))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms for {nf} frames of {H}x{W}")
ops.profile_enable(True)
enc(video)
torch.cuda.synchronize()
rows = ops.profile_read()
ops.profile_enable(False)
rows.sort(key=lambda r: -r["total_ms"])
tot = sum(r["total_ms"] for r in rows)
print(f"profiled kernels: {tot:.2f} ms")
for r in rows:
    print(f"  {r['name']:40s} n={r['launches']:3d} {r['total_ms']:8.3f} ms  {r['flops'] / max(r['total_ms'], 1e-9) / 1e9:8.1f} TF/s  {r['bytes'] / max(r['total_ms'], 1e-9) / 1e6:8.1f} GB/s")
