"""Generate golden vectors by executing the UNMODIFIED reference on CPU fp32.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference cannot travel to the GPU box, so its outputs are committed here
as small .npz fixtures.  Weights are NOT stored: both sides regenerate them with
``cotracker_amd.weights.fill_synthetic_`` (name+shape keyed, numpy RandomState).
torch version used is recorded in every file.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from cotracker.models.core.model_utils import bilinear_sampler  # noqa: E402
from cotracker.models.core.cotracker.cotracker3_online import CoTrackerThreeOnline, posenc  # noqa: E402
from cotracker.models.core.cotracker.cotracker3_offline import CoTrackerThreeOffline  # noqa: E402
from cotracker.models.core.cotracker.blocks import CorrBlock  # noqa: E402
from cotracker.models.core.cotracker.cotracker import CoTracker2  # noqa: E402
from cotracker.predictor import CoTrackerPredictor, CoTrackerOnlinePredictor  # noqa: E402

from cotracker_amd.weights import fill_synthetic_  # noqa: E402
from cotracker_amd.synthetic import synthetic_video  # noqa: E402

META = dict(torch_version=torch.__version__, reference="facebookresearch/co-tracker@2025-03-04")


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    out["meta"] = np.array(str(META))
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def rand_pyramid(g, B, S, C, H, W, levels=4):
    f = torch.randn(B, S, C, H, W, generator=g)
    f = f / f.norm(dim=2, keepdim=True)
    pyr = [f]
    for _ in range(levels - 1):
        x = torch.nn.functional.avg_pool2d(pyr[-1].reshape(B * S, C, *pyr[-1].shape[-2:]), 2, stride=2)
        pyr.append(x.reshape(B, S, C, *x.shape[-2:]))
    return pyr


@torch.no_grad()
def gen_sampler():
    g = torch.Generator().manual_seed(11)
    out = {}
    for tag, (D, H, W) in dict(d1=(1, 24, 32), d1b=(1, 96, 128), d5=(5, 12, 16)).items():
        inp = torch.randn(2, 6, D, H, W, generator=g)
        co = torch.rand(2, 40, 7, 1, 3, generator=g) * torch.tensor([D + 2.0, W + 8.0, H + 8.0]) - torch.tensor(
            [1.0, 4.0, 4.0])
        co[:, :12] = co[:, :12].round()  # integer coordinates (round-trip is not identity)
        co[:, 12:16, ..., 1] = co[:, 12:16, ..., 1].round() + 0.5
        out[f"{tag}_input"] = inp
        out[f"{tag}_coords"] = co
        out[f"{tag}_output"] = bilinear_sampler(inp, co.clone())
    save("sampler.npz", **out)


@torch.no_grad()
def gen_ops():
    """Stage-level goldens from the reference's own modules (small sizes)."""
    torch.manual_seed(0)
    model = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=8, model_resolution=(96, 128)).eval()
    fill_synthetic_(model, seed=3)
    g = torch.Generator().manual_seed(5)
    B, S, N, C = 1, 8, 12, 128
    pyr = rand_pyramid(g, B, S, C, 24, 32)
    # coordinates: inside, near border, outside, integer
    coords = torch.rand(B, S, N, 2, generator=g) * torch.tensor([36.0, 28.0]) - 2.0
    coords[:, :, :3] = coords[:, :, :3].round()
    qf = torch.randint(0, S, (B, N), generator=g)
    qc = torch.rand(B, N, 2, generator=g) * torch.tensor([31.0, 23.0])
    out = dict(coords=coords, queried_frames=qf, queried_coords=qc)
    sup = []
    for i in range(4):
        out[f"fmaps{i}"] = pyr[i]
        _, s = model.get_track_feat(pyr[i], qf, qc / 2**i, support_radius=3)
        sup.append(s)
        out[f"support{i}"] = s
        cf = model.get_correlation_feat(pyr[i], coords.view(B * S, N, 2) / 2**i)
        tfs = s.view(B, 7, 7, N, C).permute(0, 3, 1, 2, 4)
        vol = torch.einsum("btnhwc,bnijc->btnhwij", cf, tfs)
        if i in (0, 3):  # keep the fixture small: volumes for the finest and coarsest level only
            out[f"corr_volume{i}"] = vol.reshape(B, S, N, 2401)
        out[f"corr_emb{i}"] = model.corr_mlp(vol.reshape(B * S * N, 2401)).reshape(B, S, N, 256)
    # posenc / time embedding
    x4 = torch.randn(5, 7, 4, generator=g) * 0.3
    out["posenc_in"] = x4
    out["posenc_out"] = posenc(x4, 0, 10)
    out["time_emb"] = model.time_emb
    for t in (5, 8, 12, 24):
        out[f"time_emb_interp{t}"] = model.interpolate_time_embed(torch.zeros(1), t)
    # updateformer on a random token tensor
    x = torch.randn(B, N, S, 1110, generator=g) * 0.5
    out["uf_x"] = x
    out["uf_delta"] = model.updateformer(x, add_space_attn=True)
    # full forward_window, 3 iterations, with non-zero vis/conf init
    vis = torch.randn(B, S, N, 1, generator=g) * 0.1
    conf = torch.randn(B, S, N, 1, generator=g) * 0.1
    cinit = qc.reshape(B, 1, N, 2).expand(B, S, N, 2).contiguous()
    cp, vp, fp = model.forward_window(pyr, cinit, [s.unsqueeze(1) for s in sup], vis=vis, conf=conf,
                                      attention_mask=None, iters=3, add_space_attn=True)
    out["fw_vis_init"], out["fw_conf_init"] = vis, conf
    for it in range(3):
        out[f"fw_coords{it}"] = cp[it]  # already * stride
        out[f"fw_vis{it}"] = vp[it]
        out[f"fw_conf{it}"] = fp[it]
    save("ops.npz", **out)


def _queries(g, N, T, H, W):
    q = torch.zeros(1, N, 3)
    q[0, :, 0] = torch.randint(0, T, (N,), generator=g).float()
    q[0, : N // 2, 0] = 0
    q[0, :, 1] = torch.rand(N, generator=g) * (W - 1)
    q[0, :, 2] = torch.rand(N, generator=g) * (H - 1)
    return q


@torch.no_grad()
def gen_models():
    """Model-level goldens incl. encoder: small frames, window_len 8."""
    H, W = 64, 96
    g = torch.Generator().manual_seed(21)
    out = {}
    # online-weights model, sliding windows (T=20 -> pad 24, 5 windows) + streaming equivalence
    torch.manual_seed(0)
    m = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=8, model_resolution=(H, W)).eval()
    fill_synthetic_(m, seed=1)
    video = synthetic_video(20, H, W, seed=1234)
    q = _queries(g, 10, 14, H, W)
    c, v, f, _ = m(video, q, iters=4)
    out.update(on_video=video, on_queries=q, on_coords=c, on_vis=v, on_conf=f)
    # fmaps of the same video (so the oracle can be pinned without an encoder)
    fm = m.fnet(2 * (video[0] / 255.0) - 1.0)
    out["on_fnet"] = fm
    # streaming: 8-frame chunks advancing by 4
    m.init_video_online_processing()
    for ind in range(0, 20 - 4, 4):
        cs, vs, fs, _ = m(video[:, ind: ind + 8], q, iters=4, is_online=True)
    out.update(on_stream_coords=cs, on_stream_vis=vs, on_stream_conf=fs)
    save("model_online.npz", **out)

    out = {}
    torch.manual_seed(0)
    m = CoTrackerThreeOffline(stride=4, corr_radius=3, window_len=8, model_resolution=(H, W)).eval()
    fill_synthetic_(m, seed=2)
    video = synthetic_video(12, H, W, seed=99)
    q = _queries(g, 9, 12, H, W)
    c, v, f, _ = m(video, q, iters=4)
    out.update(off_video=video, off_queries=q, off_coords=c, off_vis=v, off_conf=f)
    out["off_fnet"] = m.fnet(2 * (video[0] / 255.0) - 1.0)
    save("model_offline.npz", **out)


@torch.no_grad()
def gen_predictors():
    """Predictor-level goldens at the real model resolution (384x512 internally)."""
    out = {}
    video = synthetic_video(10, 96, 128, seed=7)  # predictor resizes to 384x512
    out["video"] = video
    torch.manual_seed(0)
    p = CoTrackerPredictor(checkpoint=None, offline=True, window_len=60)
    fill_synthetic_(p.model, seed=4)
    tr, vi = p(video, grid_size=4)
    out.update(offline_grid_tracks=tr, offline_grid_vis=vi)
    q = torch.tensor([[[0.0, 30.0, 20.0], [3.0, 100.0, 70.0], [5.0, 64.0, 48.0]]])
    tr, vi = p(video, queries=q)
    out.update(queries=q, offline_q_tracks=tr, offline_q_vis=vi)
    tr, vi = p(video, queries=q, backward_tracking=True)
    out.update(offline_qb_tracks=tr, offline_qb_vis=vi)

    torch.manual_seed(0)
    p = CoTrackerPredictor(checkpoint=None, offline=False, window_len=8)
    fill_synthetic_(p.model, seed=5)
    tr, vi = p(video, grid_size=4)
    out.update(sliding_grid_tracks=tr, sliding_grid_vis=vi)

    torch.manual_seed(0)
    p = CoTrackerOnlinePredictor(checkpoint=None, window_len=8)
    fill_synthetic_(p.model, seed=5)
    p(video_chunk=video, is_first_step=True, grid_size=4)
    for ind in range(0, video.shape[1] - p.step, p.step):
        tr, vi = p(video_chunk=video[:, ind: ind + p.step * 2])
    out.update(online_grid_tracks=tr, online_grid_vis=vi)
    save("predictor.npz", **out)


@torch.no_grad()
def gen_predictor_modes():
    """Dense mode (predictor.py:70-98: queries=None, grid_size=0 -> grid_step^2 independently tracked chunks) and the
    segm_mask grid filter (predictor.py:131-140)."""
    import contextlib
    import io
    out = {}
    video = synthetic_video(6, 48, 160, seed=17)  # W=160 -> grid_step 2: 4 chunks of 80 x 24 points
    torch.manual_seed(0)
    p = CoTrackerPredictor(checkpoint=None, offline=False, window_len=8)
    fill_synthetic_(p.model, seed=5)
    with contextlib.redirect_stdout(io.StringIO()):  # the reference prints "step i / n" per chunk
        tr, vi = p(video)
    out.update(dense_video=video, dense_tracks=tr, dense_vis=vi)
    mask = torch.zeros(1, 1, 48, 160)
    mask[:, :, 8:40, 30:120] = 1.0
    tr, vi = p(video, grid_size=12, segm_mask=mask)
    out.update(segm_mask=mask, segm_tracks=tr, segm_vis=vi)
    save("predictor_modes.npz", **out)


@torch.no_grad()
def gen_corrblock():
    """CoTracker2's CorrBlock (blocks.py:284-362) and the 4-D bilinear_sampler under it."""
    g = torch.Generator().manual_seed(31)
    out = {}
    # (i) 4-D sampler on single-channel maps: in range, out of range, integer and half-integer coordinates
    for tag, (H, W) in dict(a=(16, 24), b=(3, 2), c=(1, 5)).items():
        inp = torch.randn(6, 1, H, W, generator=g)
        co = torch.rand(6, 7, 7, 2, generator=g) * torch.tensor([W + 6.0, H + 6.0]) - 3.0
        co[:2] = co[:2].round()
        co[2, ..., 0] = co[2, ..., 0].round() + 0.5
        out[f"s4_{tag}_input"], out[f"s4_{tag}_coords"] = inp, co
        out[f"s4_{tag}_output"] = bilinear_sampler(inp, co.clone(), padding_mode="border")
    # (ii) CorrBlock as CoTracker2 builds it (cotracker.py:119-124): 4 levels, radius 3, border padding
    B, S, N, C, H, W = 1, 3, 12, 128, 16, 24
    fmaps = torch.randn(B, S, C, H, W, generator=g)
    targets = torch.randn(B, S, N, C, generator=g)
    coords = torch.rand(B, S, N, 2, generator=g) * torch.tensor([W + 4.0, H + 4.0]) - 2.0
    coords[:, :, :3] = coords[:, :, :3].round()
    coords[:, :, 3] = torch.tensor([0.0, 0.0])
    coords[:, :, 4] = torch.tensor([W - 1.0, H - 1.0])
    cb = CorrBlock(fmaps, num_levels=4, radius=3, padding_mode="border")
    cb.corr(targets)
    out.update(cb_fmaps=fmaps, cb_targets=targets, cb_coords=coords, cb_out=cb.sample(coords))
    for i in range(4):
        out[f"cb_corrs{i}"] = cb.corrs_pyramid[i]
    save("corrblock.npz", **out)


@torch.no_grad()
def gen_cotracker2():
    """CoTracker2 (cotracker.py:29-384): update former with masks, forward_window, sliding and streaming forwards."""
    H, W = 64, 96
    g = torch.Generator().manual_seed(41)
    out = {}
    torch.manual_seed(0)
    m = CoTracker2(stride=4, window_len=8, model_resolution=(H, W)).eval()
    # head_scale 1 (not the CoTracker3 "stress" 8): CoTracker2 feeds the updated track feature back into the
    # correlation, so large per-iteration moves make the map chaotic (fp32 noise grows ~40x per iteration at 8)
    fill_synthetic_(m, seed=6, head_scale=1.0)
    # (i) update former alone: B=1, N=10 (3 masked), S=8, input 456, attention mask per (frame, point)
    B, N, S = 1, 10, 8
    x = torch.randn(B, N, S, 456, generator=g)
    mask = torch.ones(B * S, N, dtype=torch.bool)
    mask[:, 7:] = False
    out.update(uf_x=x, uf_mask=mask, uf_delta=m.updateformer(x, mask))
    # (ii) forward_window: 3 iterations on random (unnormalised) feature maps
    fm = torch.nn.functional.interpolate(torch.randn(B * S, 128, H // 16, W // 16, generator=g), size=(H // 4, W // 4),
                                         mode="bilinear", align_corners=True).reshape(B, S, 128, H // 4, W // 4)  # smooth
    qc = torch.rand(B, N, 2, generator=g) * torch.tensor([W / 4 - 1.0, H / 4 - 1.0])
    qc[:, :2] = qc[:, :2].round()
    coords = qc[:, None].expand(B, S, N, 2) + 0.3 * torch.randn(B, S, N, 2, generator=g)
    tf = torch.randn(B, 1, N, 128, generator=g).repeat(1, S, 1, 1)
    amask = mask.reshape(B, S, N)
    vis = torch.ones(B, S, N, 1) * 10
    tmask = torch.ones(B, S, N, 1, dtype=torch.bool)
    tmask[:, :3, 2:5] = False
    cps, vp = m.forward_window(fmaps=fm, coords=coords.clone(), track_feat=amask.unsqueeze(-1) * tf, vis=vis,
                               track_mask=tmask, attention_mask=amask, iters=3)
    out.update(fw_fmaps=fm, fw_coords=coords, fw_track_feat=tf, fw_vis=vis, fw_track_mask=tmask, fw_attention_mask=amask,
               fw_out_coords=cps[-1], fw_out_vis=vp)
    # (iii) full forwards incl. encoder: sliding windows (T=20 -> 5 windows) and streaming.  ONE iteration per window:
    # with random weights the 6+6-layer CoTracker2 iteration is chaotic -- the reference itself differs by 0.32 px
    # between 1 and 8 CPU threads after 4 iterations (6e-5 px after 1) -- so only single iterations can be pinned.
    video = synthetic_video(20, H, W, seed=4321)
    q = _queries(g, 9, 14, H, W)
    c, v, _ = m(video, q, iters=1)
    out.update(video=video, queries=q, coords=c, vis=v)
    m.init_video_online_processing()
    for ind in range(0, video.shape[1] - 4, 4):
        cs, vs, _ = m(video[:, ind:ind + 8], q, iters=1, is_online=True)
    out.update(stream_coords=cs, stream_vis=vs)
    out["fmaps"] = m.fnet(2 * (video[0] / 255.0) - 1.0)[None]
    save("cotracker2.npz", **out)


@torch.no_grad()
def gen_eval_predictor():
    """EvaluationPredictor (evaluation_predictor.py:25-213), the TAP-Vid protocol front end: single-point mode (one model
    call per query with its local 8x8 grid + the global 5x5 grid) and joint mode, on the offline model.
    evaluation_predictor.py imports torchvision.transforms.Compose (unused; torchvision is absent here): a stub module
    is registered for the import only -- the reference source is not touched."""
    import types
    tv, tvt = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")
    tvt.Compose = object
    tv.transforms = tvt
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tvt)
    from cotracker.models.evaluation_predictor import EvaluationPredictor
    out = {}
    video = synthetic_video(8, 96, 128, seed=23)
    q = torch.tensor([[[0.0, 30.0, 20.0], [2.0, 100.0, 70.0], [5.0, 64.0, 48.0]]])
    torch.manual_seed(0)
    m = CoTrackerThreeOffline(stride=4, corr_radius=3, window_len=60, model_resolution=(384, 512)).eval()
    fill_synthetic_(m, seed=4)
    for single in (True, False):
        ev = EvaluationPredictor(m, grid_size=5, local_grid_size=8, single_point=single, n_iters=6)
        tr, vi = ev(video, q)
        out["single_tracks" if single else "joint_tracks"] = tr
        out["single_vis" if single else "joint_vis"] = vi
    save("eval_predictor.npz", video=video, queries=q, **out)


@torch.no_grad()
def gen_cotracker2_damped():
    """CoTracker2 full forwards with a DAMPED feedback loop and 4 iterations per window.  With random weights the
    6+6-layer iteration is chaotic (see gen_cotracker2) and damping the heads alone does not help: the feature update
    goes through GroupNorm (cotracker.py:167), which renormalises whatever the head emits (head_scale 0.25 alone: the
    reference's own 8-vs-1-thread spread is 7.6e-4 px / 2.9e-3 logit).  So the track_feat_updater Linear is scaled by
    0.1 as well, which makes four iterations pinnable.  The reference's own spread is stored next to the outputs."""
    H, W = 64, 96
    g = torch.Generator().manual_seed(43)
    video = synthetic_video(20, H, W, seed=4321)
    q = _queries(g, 9, 14, H, W)
    res = {}
    for threads in (8, 1):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        m = CoTracker2(stride=4, window_len=8, model_resolution=(H, W)).eval()
        fill_synthetic_(m, seed=6, head_scale=0.25)
        m.track_feat_updater[0].weight.mul_(0.1)
        m.track_feat_updater[0].bias.mul_(0.1)
        c, v, _ = m(video, q, iters=4)
        m.init_video_online_processing()
        for ind in range(0, video.shape[1] - 4, 4):
            cs, vs, _ = m(video[:, ind:ind + 8], q, iters=4, is_online=True)
        res[threads] = (c, v, cs, vs)
    torch.set_num_threads(8)
    c, v, cs, vs = res[8]
    lg = lambda p: torch.log(p.double() / (1 - p.double()))  # noqa: E731
    noise = dict(noise_coords=(res[8][0] - res[1][0]).abs().max(), noise_vis_logit=(lg(res[8][1]) - lg(res[1][1])).abs().max(),
                 noise_stream_coords=(res[8][2] - res[1][2]).abs().max())
    print("cotracker2 damped: reference 8 vs 1 threads", {k: float(x) for k, x in noise.items()},
          "track motion", float((c - q[:, None, :, 1:]).abs().max()))
    save("cotracker2_damped.npz", video=video, queries=q, coords=c, vis=v, stream_coords=cs, stream_vis=vs, **noise)


if __name__ == "__main__":
    torch.set_num_threads(8)
    if sys.argv[1:] == ["predictor_modes"]:
        gen_predictor_modes()
        sys.exit(0)
    if sys.argv[1:] == ["eval_predictor"]:
        gen_eval_predictor()
        sys.exit(0)
    if sys.argv[1:] == ["cotracker2_damped"]:
        gen_cotracker2_damped()
        sys.exit(0)
    gen_sampler()
    gen_ops()
    gen_models()
    gen_predictors()
    gen_predictor_modes()
    gen_corrblock()
    gen_cotracker2()
    gen_cotracker2_damped()
    gen_eval_predictor()
