"""CoTracker2 / CoTracker2.1 on the MI355X primitives.

Host-side mirror of ``CoTracker2`` (cotracker/models/core/cotracker/cotracker.py:29-384): same constructor kwargs,
attributes, ``forward`` signature / 3-tuple return, online-state methods and the same 321 ``state_dict`` keys
(``time_emb``, ``pos_emb``, ``fnet.*``, ``updateformer.*`` with 6 time + 6 space layers and a 130-wide ``flow_head``,
``norm.*``, ``track_feat_updater.0.*``, ``vis_predictor.0.*``), so reference checkpoints load unchanged.
"""
import torch
import torch.nn as nn

from .encoder import BasicEncoder
from .model import _Lin, _UpdateFormerParams, sincos_time_embed


def sincos_pos_embed_2d(dim: int, h: int, w: int) -> torch.Tensor:
    """get_2d_sincos_pos_embed (embeddings.py:11-55) -> [1, dim, h, w]: first half encodes the x (column) index,
    second half the y (row) index, each as [sin(pos*omega), cos(pos*omega)] with omega_k = 10000^(-k/(dim/4))."""
    def emb1d(d, pos):
        omega = torch.arange(d // 2, dtype=torch.double) / (d / 2.0)
        omega = 1.0 / 10000 ** omega
        out = torch.einsum("m,d->md", pos.reshape(-1).double(), omega)
        return torch.cat([torch.sin(out), torch.cos(out)], dim=1)
    gw, gh = torch.meshgrid(torch.arange(w, dtype=torch.float), torch.arange(h, dtype=torch.float), indexing="xy")
    emb = torch.cat([emb1d(dim // 2, gw), emb1d(dim // 2, gh)], dim=1).float()  # (h*w, dim): grid[0] = x index
    return emb.reshape(1, h, w, dim).permute(0, 3, 1, 2).contiguous()


class _Affine128(nn.Module):  # nn.GroupNorm(1, 128) parameters (cotracker.py:79)
    def __init__(self, dim=128):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class CoTracker2(nn.Module):
    """Constructor mirrors cotracker.py:30-84."""

    def __init__(self, window_len=8, stride=4, add_space_attn=True, num_virtual_tracks=64, model_resolution=(384, 512)):
        super().__init__()
        if num_virtual_tracks != 64 or not add_space_attn:
            raise NotImplementedError("HIP path is specialised to 64 virtual tracks with space attention on")
        self.window_len = window_len
        self.stride = stride
        self.hidden_dim = 256
        self.latent_dim = 128
        self.add_space_attn = add_space_attn
        self.num_virtual_tracks = num_virtual_tracks
        self.model_resolution = model_resolution
        self.input_dim = 456
        self.fnet = BasicEncoder(input_dim=3, output_dim=self.latent_dim, stride=stride)
        self.updateformer = _UpdateFormerParams(self.input_dim, 384, 6, num_virtual_tracks,
                                                flow_out=self.latent_dim + 2, vis_conf_head=False)
        self.register_buffer("time_emb", sincos_time_embed(self.input_dim, window_len))
        self.register_buffer("pos_emb", sincos_pos_embed_2d(self.input_dim, model_resolution[0] // stride,
                                                           model_resolution[1] // stride))
        self.norm = _Affine128(self.latent_dim)
        self.track_feat_updater = nn.Sequential(_Lin(self.latent_dim, self.latent_dim))  # + nn.GELU() (no parameters)
        self.vis_predictor = nn.Sequential(_Lin(self.latent_dim, 1))
        self._packed = None
        self.hip_graph = False   # streaming (is_online=True): replay the captured window graph; not a reference kwarg
        self._graphs = {}
        from . import model as _m
        self.precision = _m.DEFAULT_PRECISION  # Linear back end: "f16x3" (split-half MFMA) | "f32"; not a reference kwarg
        # f16 range guard of the split-half back end, as CoTrackerThreeBase (model.py): every forward checks its outputs once;
        # a non-finite result re-runs that forward on the exact-f32 back end (offline / sliding: immediately; graph streaming:
        # the flag is examined at the start of the next call and raises -- see _guarded).  range_fallbacks counts hits.
        self.range_guard = True
        self.range_fallbacks = 0
        self.stream_range_check = "deferred"  # or "immediate" (see CoTrackerThreeBase)
        self._pending_range = None
        # "hip" (default): BasicEncoder on the library's split-half implicit-GEMM convolutions (encoder_hip.py), without the
        # final L2 normalisation CoTracker3 applies; "torch": nn.Conv2d on PyTorch-ROCm / MIOpen (A/B).  Not a reference kwarg.
        self.encoder_backend = "hip"
        self.encoder_chunk = 16
        self._hip_encoder = None


# ------------------------------------------------------------------------------------------
# device-side packed weights (ctk_former_weights of include/ctk.h)
# ------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402

from . import _lib as L  # noqa: E402
from . import ops  # noqa: E402

IN_LD, OUT_LD, DEPTH_V2 = 480, 192, 6  # input_dim 456 / output_dim 130 padded for the GEMM tiles


class PackedWeightsV2:
    def __init__(self, model: "CoTracker2", device, precision: str = "f16x3"):
        if precision not in ("f16x3", "f32"):
            raise ValueError("precision must be 'f16x3' (split-half MFMA, default) or 'f32' (exact-f32 MFMA)")
        self.precision, self.device = precision, device
        split = precision == "f16x3"
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in model.state_dict().items()}
        self.keep = []

        def hold(t):
            t = t.contiguous()
            self.keep.append(t)
            return t.data_ptr()

        def pack(t):
            if not split:
                return None
            blob = ops.pack_weight(t.contiguous())
            self.keep.append(blob)
            return blob.data_ptr()

        u = "updateformer."
        fw = L.FormerWeights()
        fw.depth, fw.in_dim, fw.in_ld, fw.out_dim, fw.out_ld = DEPTH_V2, model.input_dim, IN_LD, model.latent_dim + 2, OUT_LD
        in_w = torch.zeros(384, IN_LD, device=device)
        in_w[:, : model.input_dim] = sd[u + "input_transform.weight"]
        fw.in_w, fw.in_p = hold(in_w), pack(in_w)
        fw.in_b = hold(sd[u + "input_transform.bias"])
        # time embedding folded into per-frame bias rows: W (x + e_t) + b = W x + (W e_t + b)   (cotracker.py:150,484)
        te = sd["time_emb"][0].double()
        bias_t = (te @ sd[u + "input_transform.weight"].double().t() + sd[u + "input_transform.bias"].double()).float()
        fw.in_bias_t = hold(bias_t)
        fw.virtual_tokens = hold(sd[u + "virual_tracks"].reshape(64, 384))
        head_w = torch.zeros(OUT_LD, 384, device=device)
        head_w[: model.latent_dim + 2] = sd[u + "flow_head.weight"]
        head_b = torch.zeros(OUT_LD, device=device)
        head_b[: model.latent_dim + 2] = sd[u + "flow_head.bias"]
        # flow_head (384 -> 128 feature delta + 2 coordinates) always runs on the exact-f32 MFMA kernel (head_p = null): its output
        # re-enters the track features six times per window, and with split-half products this ONE Linear took the visibility logit
        # of the BASELINE-scale run from 5.1e-5 to 1.04e-4 against the reference (bar 1e-4; round-6 bisect, profiles/
        # r06_v2_flow_head_bisect.txt: track_feat_updater in f32 changes nothing).  It is 0.3 % of the update's flops.
        fw.head_w, fw.head_p, fw.head_b = hold(head_w), None, hold(head_b)

        def block(prefix, attn_name, cross):
            b = L.BlockWeights()
            a = f"{prefix}{attn_name}."
            b.wq, b.bq = hold(sd[a + "to_q.weight"]), hold(sd[a + "to_q.bias"])
            b.wkv, b.bkv = hold(sd[a + "to_kv.weight"]), hold(sd[a + "to_kv.bias"])
            b.wo, b.bo = hold(sd[a + "to_out.weight"]), hold(sd[a + "to_out.bias"])
            b.w1, b.b1 = hold(sd[prefix + "mlp.fc1.weight"]), hold(sd[prefix + "mlp.fc1.bias"])
            b.w2, b.b2 = hold(sd[prefix + "mlp.fc2.weight"]), hold(sd[prefix + "mlp.fc2.bias"])
            b.wq_p, b.wkv_p, b.wo_p = pack(sd[a + "to_q.weight"]), pack(sd[a + "to_kv.weight"]), pack(sd[a + "to_out.weight"])
            b.w1_p, b.w2_p = pack(sd[prefix + "mlp.fc1.weight"]), pack(sd[prefix + "mlp.fc2.weight"])
            if cross:
                b.ctx_gamma, b.ctx_beta = hold(sd[prefix + "norm_context.weight"]), hold(sd[prefix + "norm_context.bias"])
            return b

        Arr = L.BlockWeights * DEPTH_V2
        self.arrays = [Arr(*[block(f"{u}{name}.{i}.", attn, cross) for i in range(DEPTH_V2)])
                       for name, attn, cross in (("time_blocks", "attn", False), ("space_virtual2point_blocks", "cross_attn", True),
                                                 ("space_virtual_blocks", "attn", False), ("space_point2virtual_blocks", "cross_attn", True))]
        fw.time_blocks, fw.virtual2point, fw.virtual_self, fw.point2virtual = self.arrays
        self.former = fw
        self.split = split
        self.pos_hwc = sd["pos_emb"][0].permute(1, 2, 0).contiguous()  # [H/4, W/4, 456]
        self.norm_w, self.norm_b = sd["norm.weight"].contiguous(), sd["norm.bias"].contiguous()
        self.upd_w, self.upd_b = sd["track_feat_updater.0.weight"].contiguous(), sd["track_feat_updater.0.bias"].contiguous()
        self.upd_p = ops.pack_weight(self.upd_w) if split else None
        self.vis_w, self.vis_b = sd["vis_predictor.0.weight"].reshape(128).contiguous(), sd["vis_predictor.0.bias"].contiguous()
        st = L.V2Weights()  # ctk_v2_weights of the one-call window driver (ctk_forward_window_v2)
        st.former = fw
        st.pos_hwc, st.pos_h, st.pos_w = self.pos_hwc.data_ptr(), self.pos_hwc.shape[0], self.pos_hwc.shape[1]
        st.norm_w, st.norm_b = self.norm_w.data_ptr(), self.norm_b.data_ptr()
        st.upd_w, st.upd_b = self.upd_w.data_ptr(), self.upd_b.data_ptr()
        st.upd_p = self.upd_p.data_ptr() if self.upd_p is not None else None
        st.vis_w, st.vis_b = self.vis_w.data_ptr(), self.vis_b.data_ptr()
        self.struct = st


def _v2_forward_window(self, pyr, coords, track_feat, vis, track_mask, point_mask, iters, pw):
    """CoTracker2.forward_window (cotracker.py:86-173) for one batch element: ONE C call (ctk_forward_window_v2), or one
    hipGraph replay of it in streaming mode.  pyr: 4 x [S,H_l,W_l,128] (NOT normalised), coords [S,N,2] feature units,
    track_feat [S,N,128] (already masked), vis [S,N], track_mask [S,N] float 0/1, point_mask [N] uint8.
    Returns (coords [S,N,2] feature units, vis logits [S,N])."""
    if getattr(self, "hip_graph", False) and getattr(self, "_online_active", False):
        return self._graphed_window(pyr, coords, track_feat, vis, track_mask, point_mask, iters, pw)
    win = ops.V2Window(pyr, coords.clone(), track_feat.clone(), vis.contiguous(), track_mask, point_mask, iters)
    ops.forward_window_v2(win, pw)
    return win.keep[1], win.vis_out


def _v2_graphed_window(self, pyr, coords, track_feat, vis, track_mask, point_mask, iters, pw):
    """Streaming: the whole window (iters x (5 + ~390) launches) is captured once per shape and replayed per chunk."""
    key = (tuple(tuple(f.shape) for f in pyr), coords.shape[1], int(iters), id(pw), coords.device.index)
    g = self._graphs.get(key)
    if g is None:
        if self._graphs:
            torch.cuda.current_stream().synchronize()  # a replay of the graph being dropped may still be in flight
        win = ops.V2Window([f.clone() for f in pyr], coords.clone(), track_feat.clone(), vis.clone(), track_mask.clone(),
                           point_mask.clone(), iters)
        g = ops.V2WindowGraph(win, pw)
        self._graphs = {key: g}
    else:
        st_pyr, c_, tf_, v_, tm_, pm_ = g.win.keep
        for d, s_ in zip(st_pyr, pyr):
            d.copy_(s_)
        c_.copy_(coords)
        tf_.copy_(track_feat)
        v_.copy_(vis)
        tm_.copy_(track_mask)
        pm_.copy_(point_mask)
    g.launch()
    return g.win.keep[1].clone(), g.win.vis_out.clone()


def _v2_init_online(self):  # cotracker.py:187-191
    self._resolve_deferred_range_check()  # the last chunk of the previous stream (graph streaming defers its check by one call)
    self._online_batch = None  # B > 1: one state tuple per batch element
    self.online_ind = 0
    self.online_track_feat = None
    self.online_coords_predicted = None
    self.online_vis_predicted = None


@torch.no_grad()
def _v2_forward(self, video, queries, iters=4, is_train=False, is_online=False):
    """CoTracker2.forward (cotracker.py:193-384): returns (coords [B,T,N,2] px, vis [B,T,N] post-sigmoid, None)."""
    if is_train:
        raise NotImplementedError("inference-only implementation (training is out of scope)")
    if not video.is_cuda:
        raise RuntimeError("cotracker_amd runs on an MI355X GPU only: move the model and inputs to 'cuda'. There is no CPU path.")
    B, T, C_, H, W = video.shape
    S = self.window_len
    assert S >= 2
    if is_online:
        assert T <= S, "Online mode: video chunk must be <= window size."
        assert getattr(self, "online_ind", None) is not None, "Call model.init_video_online_processing() first."
    self._online_active = bool(is_online)
    # graph streaming: the chunk stream never waits for the GPU (stream_range_check = "immediate": one sync per chunk, transparent re-run)
    deferred = bool(is_online and self.hip_graph and self.stream_range_check == "deferred" and B == 1)

    def snapshot():
        return (self.online_ind, self.online_track_feat, self.online_coords_predicted, self.online_vis_predicted)

    def restore(st):
        if st is not None:  # (offline / sliding: no online state to put back before the exact-f32 re-run)
            self.online_ind, self.online_track_feat, self.online_coords_predicted, self.online_vis_predicted = st

    def run(b):
        return self._guarded(lambda prec: self._forward_one(video[b], queries[b], iters, is_online, prec),
                             snapshot() if is_online else None, restore, deferred)

    if is_online and B > 1:
        # the reference carries the batch inside its online state tensors (cotracker.py:233-259); here every batch element owns a
        # state tuple that is swapped in around its (independent) window, as CoTrackerThreeOnline does
        states = getattr(self, "_online_batch", None) or [snapshot()] * B
        assert len(states) == B, "batch size changed between online calls"
        outs = []
        for b in range(B):
            restore(states[b])
            outs.append(run(b))
            states[b] = snapshot()
        self._online_batch = states
    else:
        outs = [run(b) for b in range(B)]
    self.last_logits = (torch.stack([o[1] for o in outs]),)  # pre-sigmoid visibility [B,T,N] (parity tests compare logits)
    return torch.stack([o[0] for o in outs]), torch.sigmoid(self.last_logits[0]), None


def _v2_encode(self, frames):
    """frames [T,3,H,W] in 0..255 -> NHWC level-0 features [T,H/4,W/4,128], NOT normalised (cotracker.py:273-275)."""
    if self.encoder_backend == "hip":
        enc = self._hip_encoder
        if enc is None or enc.device != frames.device:
            from .encoder_hip import HipEncoder
            enc = self._hip_encoder = HipEncoder(self.fnet, frames.device, normalize=False)
        T, _, H, W = frames.shape
        out = torch.empty(T, H // self.stride, W // self.stride, self.latent_dim, device=frames.device, dtype=torch.float32)
        for t0 in range(0, T, self.encoder_chunk):
            enc(frames[t0:t0 + self.encoder_chunk].float().contiguous(), out=out[t0:t0 + self.encoder_chunk])
        return out
    return self.fnet(2 * (frames.float() / 255.0) - 1.0).float().permute(0, 2, 3, 1).contiguous()


def _v2_forward_one(self, video, queries, iters, is_online, precision=None):
    T, N = video.shape[0], queries.shape[0]
    S, step, dev = self.window_len, self.window_len // 2, video.device
    pw = self.packed(dev, precision)
    queries = queries.float()
    qframes = queries[:, 0].long()
    qcoords = (queries[:, 1:3] / self.stride).contiguous()
    coords_pred = torch.zeros(T, N, 2, device=dev)
    vis_pred = torch.zeros(T, N, device=dev)
    if is_online and self.online_coords_predicted is not None:  # :247-259
        p = min(step, T - step)
        coords_pred = F.pad(self.online_coords_predicted, (0, 0, 0, 0, 0, p))
        vis_pred = F.pad(self.online_vis_predicted, (0, 0, 0, p))
    # encoder; padding the video with its last frame (:264-270) == repeating the last feature map (fnet is per-frame)
    pad = (S - T) if is_online else (S - T % S) % S
    f0 = self._encode(video)  # NHWC, not normalised
    if pad > 0:
        f0 = torch.cat([f0, f0[-1:].expand(pad, -1, -1, -1)], dim=0).contiguous()
    pyr = ops.build_pyramid(f0, 4)  # CorrBlock pyramid (blocks.py:300-307) for every frame at once
    # get_track_feat (:175-185): trilinear sample at (t, x, y) = the centre tap of the support sampler
    frames_rel = (qframes - self.online_ind if is_online else qframes).float().contiguous()
    tf0 = ops.sample_support(f0, frames_rel, qcoords)[:, 24].contiguous()  # [N,128]
    track_feat = tf0[None].expand(S, N, 128)
    if is_online:  # :286-295
        left = 0 if self.online_ind == 0 else self.online_ind + step
        right = self.online_ind + S
        smask = ((qframes >= left) & (qframes < right)).float()[None, :, None]
        if self.online_track_feat is None:
            self.online_track_feat = torch.zeros(S, N, 128, device=dev)
        self.online_track_feat = self.online_track_feat + track_feat * smask
        track_feat = self.online_track_feat
    num_windows = (T - S + step - 1) // step + 1
    indices = [self.online_ind] if is_online else range(0, step * num_windows, step)
    coords_init = qcoords[None].expand(S, N, 2).contiguous()
    vis_init = torch.full((S, N), 10.0, device=dev)
    for ind in indices:
        overlap = S - step
        if ind > 0:  # :306-327
            copy_over = (qframes < ind + overlap)[None, :]
            cprev = coords_pred[ind:ind + overlap] / self.stride
            cprev = torch.cat([cprev, cprev[-1:].expand(step, -1, -1)], dim=0)
            vprev = vis_pred[ind:ind + overlap]
            vprev = torch.cat([vprev, vprev[-1:].expand(step, -1)], dim=0)
            coords_init = torch.where(copy_over[..., None], cprev, coords_init).contiguous()
            vis_init = torch.where(copy_over, vprev, vis_init).contiguous()
        amask = qframes < ind + S                                                        # attention_mask, :331-333
        tmask = qframes[None, :] <= torch.arange(ind, ind + S, device=dev)[:, None]      # track_mask, :338-344
        if ind > 0:
            tmask = tmask.clone()
            tmask[:overlap] = False
        win_pyr = pyr if is_online else [p_[ind:ind + S] for p_ in pyr]
        coords, vis = self.forward_window(win_pyr, coords_init, (track_feat * amask.float()[None, :, None]).contiguous(), vis_init,
                                          tmask.float().contiguous(), amask.to(torch.uint8).contiguous(), iters, pw)
        S_trim = T if is_online else min(T - ind, S)
        coords_pred[ind:ind + S] = (coords * float(self.stride))[:S_trim]
        vis_pred[ind:ind + S] = vis[:S_trim]
    if is_online:
        self.online_ind += step
        self.online_coords_predicted = coords_pred
        self.online_vis_predicted = vis_pred
    return coords_pred, vis_pred  # (visibility LOGITS: forward applies the sigmoid, cotracker.py:373)


def _v2_packed(self, device, precision=None):
    precision = precision or self.precision
    if not isinstance(self._packed, dict):
        self._packed = {}
    pw = self._packed.get(precision)
    if pw is None or pw.device != device:
        pw = self._packed[precision] = PackedWeightsV2(self, device, precision)
    return pw


def _v2_invalidate(self):
    self._packed = None
    self._hip_encoder = None
    if getattr(self, "_graphs", None):
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        self._graphs = {}  # captured graphs hold pointers into the old packed weights


def _v2_load_state_dict(self, *args, **kwargs):
    _v2_invalidate(self)
    return nn.Module.load_state_dict(self, *args, **kwargs)


def _v2_apply(self, fn, *args, **kwargs):
    _v2_invalidate(self)
    return nn.Module._apply(self, fn, *args, **kwargs)


_V2_TRANSIENT = {"_packed": type(None), "_graphs": dict, "_hip_encoder": type(None), "_pending_range": type(None)}  # (online state incl. _online_batch is ordinary tensors: copied)


def _v2_getstate(self):  # the packed-weight cache holds ctypes structs with raw pointers: never pickled / deep-copied
    self._resolve_deferred_range_check()  # may wait for the last streamed chunk and raise FloatingPointError (as model.py)
    st = self.__dict__.copy()
    for k, mk in _V2_TRANSIENT.items():
        if k in st:
            st[k] = mk()
    return st


def _v2_deepcopy(self, memo):
    import copy
    self._resolve_deferred_range_check()  # before the copy is registered in memo: a raise leaves nothing half-built behind
    new = self.__class__.__new__(self.__class__)
    memo[id(self)] = new
    for k, v in self.__dict__.items():
        new.__dict__[k] = _V2_TRANSIENT[k]() if k in _V2_TRANSIENT else copy.deepcopy(v, memo)
    return new


CoTracker2.__getstate__ = _v2_getstate
CoTracker2.__deepcopy__ = _v2_deepcopy
CoTracker2.forward_window = _v2_forward_window
CoTracker2._graphed_window = _v2_graphed_window
CoTracker2.init_video_online_processing = _v2_init_online
CoTracker2.forward = _v2_forward
CoTracker2._forward_one = _v2_forward_one
CoTracker2._encode = _v2_encode
from .model import CoTrackerThreeBase as _Base  # noqa: E402  (the range guard is the same code for both model families)
CoTracker2._guarded = _Base._guarded
CoTracker2._resolve_deferred_range_check = _Base._resolve_deferred_range_check
CoTracker2.packed = _v2_packed
CoTracker2.invalidate_packed_weights = _v2_invalidate
CoTracker2.load_state_dict = _v2_load_state_dict
CoTracker2._apply = _v2_apply
