"""torch-CPU port of the reference's predictor path -- the CPU baseline of bench.py (``cpu_baseline.kind = "port-torch"``).

THIS IS TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/cotracker_oracle.py): only tests/ and
bench.py's cpu_baseline leg may use it.

The reference itself cannot travel to the GPU box, and the numpy oracle is >= 4x slower than the reference because it
restates ATen's kernels in numpy.  This port calls the SAME ATen CPU kernels the reference calls -- conv2d /
instance_norm for the encoder, ``F.grid_sample`` 5-D for both samplers, ``einsum`` for the 49x49 correlation,
``F.linear`` / ``F.layer_norm`` / ``F.gelu`` / ``softmax`` for corr_mlp and EfficientUpdateFormer -- in the same order and
on tensors of the same shapes, so its wall time on a host is what the reference's would be (tools/time_cpu_reference.py
measures both in the build container: profiles/r02_cpu_reference_vs_port.txt).  It is validated against the reference's
goldens in tests/test_oracle_golden.py.

Follows: predictor.py:100-190 (resize, grid queries, query fix-up), cotracker3_online.py:266-541 (sliding windows),
cotracker3_offline.py:62-233, cotracker3_online.py:94-264 (support / correlation / token assembly / iteration),
cotracker.py:483-531 + blocks.py:379-438 + cotracker.py:559-577 (update former), model_utils.py:191-255 (sampler).

    python -m oracle.torch_port --bench sliding --frames 32 --grid 20 --threads 0     # prints one JSON line
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# glibc malloc returns every freed multi-MB tensor to the kernel and page-faults it back in: on this workload that is
# 6x of the reference's CPU wall time (measured: C2 360 s -> 56 s).  A fair CPU baseline runs with these set.
MALLOC_ENV = {"MALLOC_MMAP_MAX_": "0", "MALLOC_TRIM_THRESHOLD_": "68719476736", "MALLOC_TOP_PAD_": "1073741824"}


def sample5d(inp, coords):
    """bilinear_sampler, 5-D branch (model_utils.py:234-255): inp [B,C,T,H,W], coords [B,R1,R2,R3,3]=(t,x,y)."""
    T, H, W = inp.shape[-3:]
    c = coords[..., [1, 2, 0]]
    c = c * torch.tensor([2 / max(W - 1, 1), 2 / max(H - 1, 1), 2 / max(T - 1, 1)], dtype=c.dtype)
    c = c - 1
    return F.grid_sample(inp, c, align_corners=True, padding_mode="border")


def support_lattice(coords, r=3):
    """get_support_points (cotracker3_online.py:94-111): coords [B,1,N,3] -> [B,N,7,7,3]; first lattice index = x."""
    d = torch.linspace(-r, r, 2 * r + 1)
    gx, gy = torch.meshgrid(d, d, indexing="ij")
    delta = torch.stack([torch.zeros_like(gx), gx, gy], dim=-1)
    B, _, N, _ = coords.shape
    return coords.reshape(B, N, 1, 1, 3) + delta.view(1, 1, 2 * r + 1, 2 * r + 1, 3)


def track_support(fmaps, qframes, qcoords, r=3):
    """get_track_feat (cotracker3_online.py:113-128): fmaps [B,T,C,H,W] -> support features [B,49,N,C]."""
    B, N = qframes.shape
    pts = torch.cat([qframes[:, None, :, None].float(), qcoords[:, None]], dim=-1)  # [B,1,N,3]
    lat = support_lattice(pts, r).reshape(B, N, 49, 1, 3).permute(0, 2, 1, 3, 4).reshape(B, 49, N, 1, 3)
    out = sample5d(fmaps.permute(0, 2, 1, 3, 4), lat)  # [B,C,49,N,1]
    return out[..., 0].permute(0, 2, 3, 1)


def correlation_feat(fmaps, coords, r=3):
    """get_correlation_feat (cotracker3_online.py:130-143): fmaps [B,S,C,H,W], coords [B*S,N,2] -> [B,S,N,7,7,C]."""
    B, S, C, H, W = fmaps.shape
    N = coords.shape[1]
    pts = torch.cat([torch.zeros_like(coords[..., :1]), coords], dim=-1)[:, None]
    lat = support_lattice(pts, r)  # [B*S,N,7,7,3]
    out = sample5d(fmaps.reshape(B * S, C, 1, H, W), lat)
    return out.view(B, S, C, N, 2 * r + 1, 2 * r + 1).permute(0, 1, 3, 4, 5, 2)


def posenc(x, lo=0, hi=10):
    """cotracker3_online.py:19-39."""
    scales = 2.0 ** torch.arange(lo, hi, dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * torch.pi], dim=-1))], dim=-1)


def attention(x, ctx, p, pre, heads=8):
    """Attention.forward (blocks.py:379-398)."""
    B, N1, C = x.shape
    q = F.linear(x, p[pre + "to_q.weight"], p[pre + "to_q.bias"]).reshape(B, N1, heads, C // heads).permute(0, 2, 1, 3)
    k, v = F.linear(ctx, p[pre + "to_kv.weight"], p[pre + "to_kv.bias"]).chunk(2, dim=-1)
    k = k.reshape(B, -1, heads, C // heads).permute(0, 2, 1, 3)
    v = v.reshape(B, -1, heads, C // heads).permute(0, 2, 1, 3)
    a = ((q @ k.transpose(-2, -1)) * (48 ** -0.5)).softmax(dim=-1)
    return F.linear((a @ v).transpose(1, 2).reshape(B, N1, C), p[pre + "to_out.weight"], p[pre + "to_out.bias"])


def mlp(x, p, pre, tanh=True):
    h = F.gelu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"]), approximate="tanh" if tanh else "none")
    return F.linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def self_block(x, p, pre):
    """AttnBlock.forward (blocks.py:426-438)."""
    n = F.layer_norm(x, (384,), eps=1e-6)
    x = x + attention(n, n, p, pre + "attn.")
    return x + mlp(F.layer_norm(x, (384,), eps=1e-6), p, pre + "mlp.")


def cross_block(x, ctx, p, pre):
    """CrossAttnBlock.forward (cotracker.py:559-577)."""
    c = F.layer_norm(ctx, (384,), p[pre + "norm_context.weight"], p[pre + "norm_context.bias"], eps=1e-5)
    x = x + attention(F.layer_norm(x, (384,), eps=1e-6), c, p, pre + "cross_attn.")
    return x + mlp(F.layer_norm(x, (384,), eps=1e-6), p, pre + "mlp.")


def update_former(x, p, u="updateformer.", depth=3, V=64):
    """EfficientUpdateFormer.forward (cotracker.py:483-531): x [B,N,S,1110] -> [B,N,S,4]."""
    t = F.linear(x, p[u + "input_transform.weight"], p[u + "input_transform.bias"])
    B, _, S, _ = t.shape
    t = torch.cat([t, p[u + "virual_tracks"].repeat(B, 1, S, 1)], dim=1)
    N = t.shape[1]
    for i in range(depth):
        t = self_block(t.contiguous().view(B * N, S, -1), p, f"{u}time_blocks.{i}.").view(B, N, S, -1)
        s = t.permute(0, 2, 1, 3).contiguous().view(B * S, N, -1)
        pt, vt = s[:, : N - V], s[:, N - V:]
        vt = cross_block(vt, pt, p, f"{u}space_virtual2point_blocks.{i}.")
        vt = self_block(vt, p, f"{u}space_virtual_blocks.{i}.")
        pt = cross_block(pt, vt, p, f"{u}space_point2virtual_blocks.{i}.")
        t = torch.cat([pt, vt], dim=1).view(B, S, N, -1).permute(0, 2, 1, 3)
    t = t[:, : N - V]
    return torch.cat([F.linear(t, p[u + "flow_head.weight"], p[u + "flow_head.bias"]),
                      F.linear(t, p[u + "vis_conf_head.weight"], p[u + "vis_conf_head.bias"])], dim=-1)


def time_embed(p, S):
    """interpolate_time_embed (cotracker3_online.py:145-156)."""
    te = p["time_emb"]
    if S == te.shape[1]:
        return te
    return F.interpolate(te.permute(0, 2, 1), size=S, mode="linear").permute(0, 2, 1)


def forward_window(pyr, coords, support, vis, conf, p, iters, res=(384, 512), stride=4):
    """CoTrackerThreeOnline.forward_window (cotracker3_online.py:171-264); coords in level-0 units, returns last iterate."""
    B, S = pyr[0].shape[:2]
    N = coords.shape[2]
    for _ in range(iters):
        embs = []
        for l in range(4):
            cf = correlation_feat(pyr[l], coords.reshape(B * S, N, 2) / 2 ** l)
            sup = support[l].view(B, 7, 7, N, -1).permute(0, 3, 1, 2, 4)
            vol = torch.einsum("btnhwc,bnijc->btnhwij", cf, sup)
            embs.append(mlp(vol.reshape(B * S * N, 2401), p, "corr_mlp.", tanh=False))
        embs = torch.cat(embs, dim=-1).view(B, S, N, -1)
        fwd = F.pad(coords[:, :-1] - coords[:, 1:], (0, 0, 0, 0, 0, 1))
        bwd = F.pad(coords[:, 1:] - coords[:, :-1], (0, 0, 0, 0, 1, 0))
        scale = torch.tensor([res[1], res[0]], dtype=coords.dtype) / stride
        rel = posenc(torch.cat([fwd / scale, bwd / scale], dim=-1))
        x = torch.cat([vis, conf, embs, rel], dim=-1).permute(0, 2, 1, 3).reshape(B * N, S, -1)
        x = x + time_embed(p, S)
        d = update_former(x.view(B, N, S, -1), p).permute(0, 2, 1, 3)
        coords = coords + d[..., :2]
        vis = vis + d[..., 2:3]
        conf = conf + d[..., 3:4]
    return coords * stride, vis[..., 0], conf[..., 0]


def encode(fnet, video, chunk=200):
    """cotracker3_online.py:320,362-409: 2*(v/255)-1 -> fnet -> channel L2 normalise -> 4-level avg_pool pyramid."""
    B, T = video.shape[:2]
    v = 2 * (video / 255.0) - 1.0
    f = torch.cat([fnet(v[0, t0:t0 + chunk]) for t0 in range(0, T, chunk)], dim=0)[None]
    f = f / torch.sqrt(torch.maximum(torch.sum(f * f, dim=2, keepdim=True), torch.tensor(1e-12)))
    pyr = [f]
    for _ in range(3):
        g = F.avg_pool2d(pyr[-1].reshape(B * T, 128, *pyr[-1].shape[-2:]), 2, stride=2)
        pyr.append(g.reshape(B, T, 128, *g.shape[-2:]))
    return pyr


@torch.no_grad()
def model_forward(fnet, p, video, queries, iters=6, window_len=16, offline=False, stride=4):
    """CoTrackerThreeOnline.forward, is_online=False (cotracker3_online.py:266-541) / CoTrackerThreeOffline.forward
    (cotracker3_offline.py:62-233): video [1,T,3,H,W] at model resolution, queries [1,N,3] -> coords px, vis/conf logits."""
    B, T, _, H, W = video.shape
    N = queries.shape[1]
    S = T if offline else window_len
    step = S // 2
    qf = queries[:, :, 0].long()
    qc = queries[..., 1:3] / stride
    pad = 0 if offline else (S - T % S) % S
    if pad:
        video = torch.cat([video, video[:, -1:].expand(-1, pad, -1, -1, -1)], dim=1)
    pyr = encode(fnet, video)
    support = [track_support(pyr[l], qf, qc / 2 ** l) for l in range(4)]
    out_c = torch.zeros(B, T, N, 2)
    out_v = torch.zeros(B, T, N)
    out_f = torch.zeros(B, T, N)
    c0 = qc.reshape(B, 1, N, 2).expand(B, S, N, 2)
    v0 = torch.zeros(B, S, N, 1)
    f0 = torch.zeros(B, S, N, 1)
    if offline:
        c, v, f = forward_window(pyr, c0, support, v0, f0, p, iters, (H, W), stride)
        return c, v, f
    nwin = (T - S + step - 1) // step + 1
    for ind in range(0, step * nwin, step):
        if ind > 0:  # cotracker3_online.py:457-482
            ov = S - step
            copy = (qf < ind + ov)[:, None, :, None]
            cp = torch.cat([out_c[:, ind:ind + ov] / stride, (out_c[:, ind + ov - 1:ind + ov] / stride).expand(-1, step, -1, -1)], 1)
            vp = torch.cat([out_v[:, ind:ind + ov], out_v[:, ind + ov - 1:ind + ov].expand(-1, step, -1)], 1)[..., None]
            fp = torch.cat([out_f[:, ind:ind + ov], out_f[:, ind + ov - 1:ind + ov].expand(-1, step, -1)], 1)[..., None]
            c0 = torch.where(copy, cp, c0)
            v0 = torch.where(copy, vp, v0)
            f0 = torch.where(copy, fp, f0)
        am = (qf < ind + S).float()[:, None, :, None]  # attention_mask: zero the support of not-yet-queried tracks (:493-496)
        c, v, f = forward_window([q[:, ind:ind + S] for q in pyr], c0, [s * am for s in support], v0, f0, p, iters, (H, W), stride)
        tr = min(T - ind, S)
        out_c[:, ind:ind + S] = c[:, :tr]
        out_v[:, ind:ind + S] = v[:, :tr]
        out_f[:, ind:ind + S] = f[:, :tr]
    return out_c, out_v, out_f


@torch.no_grad()
def predictor_forward(fnet, p, video, grid_size, window_len=16, offline=False, interp=(384, 512)):
    """CoTrackerPredictor._compute_sparse_tracks with grid queries (predictor.py:100-190)."""
    from cotracker_amd.predictor import get_points_on_a_grid
    B, T, C, H, W = video.shape
    v = F.interpolate(video.reshape(B * T, C, H, W), tuple(interp), mode="bilinear", align_corners=True).reshape(B, T, 3, *interp)
    pts = get_points_on_a_grid(grid_size, interp)
    q = torch.cat([torch.zeros_like(pts[:, :, :1]), pts], dim=2)
    coords, vl, fl = model_forward(fnet, p, v, q, 6, window_len, offline)
    vis = torch.sigmoid(vl) > 0.9
    n = torch.arange(q.shape[1])
    coords[0, q[0, :, 0].long(), n] = q[0, :, 1:]
    vis[0, q[0, :, 0].long(), n] = True
    tracks = coords * coords.new_tensor([(W - 1) / (interp[1] - 1), (H - 1) / (interp[0] - 1)])
    return tracks, vis, coords, vl, fl


def build(offline, window_len, seed=0):
    """Encoder module + flat parameter dict with bench.py's synthetic weights."""
    from cotracker_amd.model import CoTrackerThreeOffline, CoTrackerThreeOnline
    from cotracker_amd.weights import fill_synthetic_
    m = (CoTrackerThreeOffline if offline else CoTrackerThreeOnline)(window_len=window_len).eval()
    fill_synthetic_(m, seed=seed)
    return m.fnet, {k: v for k, v in m.state_dict().items() if not k.startswith("fnet.")}


def bench(kind, frames, grid, size, threads, window_len=16):
    from cotracker_amd.synthetic import synthetic_video
    if threads > 0:
        torch.set_num_threads(threads)
    offline = kind == "offline"
    fnet, p = build(offline, 60 if offline else window_len)
    video = synthetic_video(frames, size, size, seed=1234)
    t0 = time.time()
    tracks, *_ = predictor_forward(fnet, p, video, grid, window_len, offline)
    dt = time.time() - t0
    assert torch.isfinite(tracks).all()
    return {"seconds": round(dt, 2), "points": grid * grid, "frames": frames, "video": [size, size], "kind": kind,
            "threads": torch.get_num_threads(), "tracked_point_frames_per_s": round(grid * grid * frames / dt, 2),
            "malloc_tuned": all(os.environ.get(k) == v for k, v in MALLOC_ENV.items())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bench", default="sliding", choices=["sliding", "offline"])
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--grid", type=int, default=20)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--threads", type=int, default=0, help="0 = torch default (all cores)")
    a = ap.parse_args()
    print(json.dumps(bench(a.bench, a.frames, a.grid, a.size, a.threads)))
