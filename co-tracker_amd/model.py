"""CoTracker3 tracker models on the MI355X hot path.

Host-side mirror of the reference's model interface:
  CoTrackerThreeOnline   <- cotracker/models/core/cotracker/cotracker3_online.py:159-541
  CoTrackerThreeOffline  <- cotracker/models/core/cotracker/cotracker3_offline.py:15-233
Same constructor kwargs, attributes (model_resolution, window_len, stride), ``forward``
signature / return tuple, online-state methods and -- crucially -- the same ``state_dict`` key
set, so reference checkpoints load unchanged and a reference ``CoTrackerPredictor`` can have its
``.model`` swapped for one of these.

Everything numeric runs in the HIP library through ``cotracker_amd.ops`` / ``encoder_hip``: the CNN
encoder (split-half implicit-GEMM convolutions, round 3; the PyTorch-ROCm / MIOpen encoder stays selectable
with ``encoder_backend = "torch"`` for A/B), feature normalisation + pyramid, support sampling, and the 6x
iterative update (correlation sampling, corr MLP, token assembly, EfficientUpdateFormer, state update).
Window scheduling and online state are Python glue, as in the reference.
Inference only (``is_train`` must be False).
"""
import warnings
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .encoder import BasicEncoder


# ------------------------------------------------------------------------------------------
# parameter containers: same names / shapes as the reference modules, no forward of their own
# ------------------------------------------------------------------------------------------
class _Lin(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.zeros(fout))
        nn.init.xavier_uniform_(self.weight)  # cotracker.py:465-469


class _Mlp(nn.Module):  # blocks.py:40-76
    def __init__(self, fin, hidden, fout):
        super().__init__()
        self.fc1 = _Lin(fin, hidden)
        self.fc2 = _Lin(hidden, fout)


class _Attn(nn.Module):  # blocks.py:365-377
    def __init__(self, dim=384):
        super().__init__()
        self.to_q = _Lin(dim, dim)
        self.to_kv = _Lin(dim, 2 * dim)
        self.to_out = _Lin(dim, dim)


class _AttnBlock(nn.Module):  # blocks.py:401-424 (both LayerNorms are parameter-free)
    def __init__(self, dim=384, ratio=4):
        super().__init__()
        self.attn = _Attn(dim)
        self.mlp = _Mlp(dim, dim * ratio, dim)


class _Affine(nn.Module):  # nn.LayerNorm(384) parameters (cotracker.py:540)
    def __init__(self, dim=384):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _CrossBlock(nn.Module):  # cotracker.py:534-557
    def __init__(self, dim=384, ratio=4):
        super().__init__()
        self.norm_context = _Affine(dim)
        self.cross_attn = _Attn(dim)
        self.mlp = _Mlp(dim, dim * ratio, dim)


class _UpdateFormerParams(nn.Module):  # cotracker.py:387-462
    def __init__(self, input_dim=1110, hidden=384, depth=3, num_virtual_tracks=64, flow_out=2, vis_conf_head=True, space_attn=True):
        """CoTracker3: flow_head(2) + vis_conf_head(2) (linear_layer_for_vis_conf=True) or one flow_head(4) (False);
        CoTracker2: one flow_head of output_dim = 130 and no vis_conf_head (cotracker.py:410-414).  space_attn=False
        (constructor add_space_attn=False): the three space block lists do not exist, as in cotracker.py:432-460."""
        super().__init__()
        self.input_transform = _Lin(input_dim, hidden)
        self.flow_head = _Lin(hidden, flow_out)
        nn.init.trunc_normal_(self.flow_head.weight, std=0.001)
        if vis_conf_head:
            self.vis_conf_head = _Lin(hidden, 2)
            nn.init.trunc_normal_(self.vis_conf_head.weight, std=0.001)
        self.virual_tracks = nn.Parameter(torch.randn(1, num_virtual_tracks, 1, hidden))  # (sic) reference key
        self.time_blocks = nn.ModuleList([_AttnBlock(hidden) for _ in range(depth)])
        if not space_attn:
            return
        self.space_virtual_blocks = nn.ModuleList([_AttnBlock(hidden) for _ in range(depth)])
        self.space_point2virtual_blocks = nn.ModuleList([_CrossBlock(hidden) for _ in range(depth)])
        self.space_virtual2point_blocks = nn.ModuleList([_CrossBlock(hidden) for _ in range(depth)])


DEFAULT_PRECISION = "f16x3"  # Linear back end of newly built models ("f16x3" | "f32"), see CoTrackerThreeBase.precision


def sincos_time_embed(dim: int, window_len: int) -> torch.Tensor:
    """get_1d_sincos_pos_embed_from_grid on linspace(0, W-1, W) (embeddings.py:59-84) -> [1,W,dim]."""
    omega = torch.arange(dim // 2, dtype=torch.double) / (dim / 2.0)
    omega = 1.0 / 10000 ** omega
    pos = torch.linspace(0, window_len - 1, window_len).double()
    out = torch.einsum("m,d->md", pos, omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)[None].float()


# ------------------------------------------------------------------------------------------
# device-side packed weights (the ctk_model_weights struct of include/ctk.h)
# ------------------------------------------------------------------------------------------
class PackedWeights:
    """Contiguous fp32 device copies in the layouts the C-ABI wants, plus the ctypes struct."""

    def __init__(self, model: "CoTrackerThreeBase", device, precision: str = "f16x3"):
        if precision not in ("f16x3", "f32"):
            raise ValueError("precision must be 'f16x3' (split-half MFMA, default) or 'f32' (exact-f32 MFMA)")
        self.precision = precision
        split = precision == "f16x3"
        sd = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in model.state_dict().items()}
        self.device = device
        self.keep: List[torch.Tensor] = []
        self.time_emb = sd["time_emb"]  # [1,W,1110], reference column order
        st = L.ModelWeights()

        def hold(t: torch.Tensor) -> int:
            t = t.contiguous()
            self.keep.append(t)
            return t.data_ptr()

        def pack(t: torch.Tensor):
            """ctk_pack_weight blob of a Linear weight (None in exact-f32 mode)."""
            if not split:
                return None
            blob = ops.pack_weight(t.contiguous())
            self.keep.append(blob)
            return blob.data_ptr()

        fc1 = torch.zeros(384, L.CORR_LD, device=device)
        fc1[:, : L.CORR_K] = sd["corr_mlp.fc1.weight"]
        st.corr_fc1_w = hold(fc1)
        st.corr_fc1_p = pack(fc1)
        st.corr_fc2_p = pack(sd["corr_mlp.fc2.weight"])
        st.corr_fc1_b = hold(sd["corr_mlp.fc1.bias"])
        st.corr_fc2_w = hold(sd["corr_mlp.fc2.weight"])
        st.corr_fc2_b = hold(sd["corr_mlp.fc2.bias"])

        u = "updateformer."
        w_ref = sd[u + "input_transform.weight"]  # columns: [vis, conf, corr(1024), posenc(84)]
        self.in_w_ref = w_ref
        self.in_b = sd[u + "input_transform.bias"]
        in_w = torch.zeros(384, L.X_LD, device=device)
        in_w[:, 0:1024] = w_ref[:, 2:1026]
        in_w[:, 1024:1026] = w_ref[:, 0:2]
        in_w[:, 1026:1110] = w_ref[:, 1026:1110]
        st.in_w = hold(in_w)
        st.in_p = pack(in_w)
        st.virtual_tokens = hold(sd[u + "virual_tracks"].reshape(64, 384))
        if u + "vis_conf_head.weight" in sd:  # linear_layer_for_vis_conf=True: two heads, concatenated (cotracker.py:526-529)
            st.head_w = hold(torch.cat([sd[u + "flow_head.weight"], sd[u + "vis_conf_head.weight"]], dim=0))
            st.head_b = hold(torch.cat([sd[u + "flow_head.bias"], sd[u + "vis_conf_head.bias"]], dim=0))
        else:                                  # linear_layer_for_vis_conf=False: ONE Linear(384, 4) = the same [4,384] matrix
            assert sd[u + "flow_head.weight"].shape == (4, 384)
            st.head_w, st.head_b = hold(sd[u + "flow_head.weight"]), hold(sd[u + "flow_head.bias"])

        def block(prefix: str, attn_name: str, cross: bool) -> L.BlockWeights:
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
                b.ctx_gamma = hold(sd[prefix + "norm_context.weight"])
                b.ctx_beta = hold(sd[prefix + "norm_context.bias"])
            return b

        has_space = f"{u}space_virtual_blocks.0.attn.to_q.weight" in sd
        for i in range(L.DEPTH):
            st.time_blocks[i] = block(f"{u}time_blocks.{i}.", "attn", False)
            if not has_space:
                # constructor add_space_attn=False: the model has no space blocks and every window runs with
                # CTK_WINDOW_NO_SPACE_ATTN, which never dereferences these slots; ctk_forward_window still validates them, so
                # they alias the time block (norm_context of the cross slots: any 384 floats)
                st.virtual_self[i] = st.virtual2point[i] = st.point2virtual[i] = st.time_blocks[i]
                st.virtual2point[i].ctx_gamma = st.point2virtual[i].ctx_gamma = st.time_blocks[i].bq
                st.virtual2point[i].ctx_beta = st.point2virtual[i].ctx_beta = st.time_blocks[i].bq
                continue
            st.virtual_self[i] = block(f"{u}space_virtual_blocks.{i}.", "attn", False)
            st.virtual2point[i] = block(f"{u}space_virtual2point_blocks.{i}.", "cross_attn", True)
            st.point2virtual[i] = block(f"{u}space_point2virtual_blocks.{i}.", "cross_attn", True)
        self.struct = st
        self._bias_t = {}

    def time_embed(self, S: int) -> torch.Tensor:
        """interpolate_time_embed (cotracker3_online.py:145-156) -> [S,1110] reference column order."""
        te = self.time_emb
        if S != te.shape[1]:
            te = F.interpolate(te.permute(0, 2, 1), size=S, mode="linear").permute(0, 2, 1)
        return te[0]

    def struct_for(self, S: int) -> L.ModelWeights:
        """Struct whose in_bias_t folds the S-frame time embedding into the input projection:
        input_transform(x + e_t) = W x + (W e_t + b)   (cotracker3_online.py:247 + cotracker.py:484)."""
        if S not in self._bias_t:
            te = self.time_embed(S).double()
            bias_t = (te @ self.in_w_ref.double().t() + self.in_b.double()).float().contiguous()
            self._bias_t[S] = bias_t
        self.struct.in_bias_t = self._bias_t[S].data_ptr()
        return self.struct


# ------------------------------------------------------------------------------------------
# models
# ------------------------------------------------------------------------------------------
def tail_aliases(prev, cur, dim, step):
    """True iff `cur` is the window of the SAME live allocation as `prev`, advanced by `step` entries along `dim` (so that
    cur[..., :n-step, ...] and prev[..., step:, ...] are the same bytes) and nothing wrote to that allocation through a
    tensor sharing its version counter in between.  Host-side metadata only: no kernel, no synchronisation.  The caller
    must have kept `prev` alive since it was recorded (a freed allocation could be handed out again at the same address)."""
    if prev is None or cur is None or prev.shape != cur.shape or prev.dtype != cur.dtype or prev.device != cur.device:
        return False
    if prev.stride() != cur.stride() or cur.shape[dim] <= step:
        return False
    try:
        same = prev.untyped_storage().data_ptr() == cur.untyped_storage().data_ptr()
    except Exception:  # storage-less tensors
        return False
    if not (same and cur.storage_offset() == prev.storage_offset() + step * prev.stride(dim)):
        return False
    # Tensors created under torch.inference_mode() do not track a version counter (reading `_version` raises): writes to them
    # cannot be seen from the host, so the overlap cannot be PROVEN -> not an alias, the caller re-encodes (advisor, round 4).
    try:
        recorded = prev._ctk_version if hasattr(prev, "_ctk_version") else prev._version
        return bool(recorded == cur._version)
    except Exception:
        return False


class CoTrackerThreeBase(nn.Module):
    """Constructor mirrors cotracker3_online.py:43-92."""

    def __init__(self, window_len=8, stride=4, corr_radius=3, corr_levels=4, num_virtual_tracks=64,
                 model_resolution=(384, 512), add_space_attn=True, linear_layer_for_vis_conf=True):
        super().__init__()
        if (corr_radius, corr_levels, num_virtual_tracks) != (3, 4, 64):
            raise NotImplementedError("the HIP kernels are specialised to corr_radius=3 (7x7 taps), corr_levels=4 and 64 virtual "
                                      "tracks -- the shapes every released model has (build_cotracker.py:31-38); see INTEGRATION.md")
        # constructor add_space_attn=False (cotracker.py:432: no space blocks at all) and linear_layer_for_vis_conf=False
        # (cotracker.py:413-414: one flow_head of width 4) are supported since round 4: same kernels, different parameter set
        self.add_space_attn = bool(add_space_attn)
        self.window_len = window_len
        self.stride = stride
        self.corr_radius = corr_radius
        self.corr_levels = corr_levels
        self.hidden_dim = 256
        self.latent_dim = 128
        self.num_virtual_tracks = num_virtual_tracks
        self.model_resolution = model_resolution
        self.input_dim = 1110
        self.linear_layer_for_vis_conf = linear_layer_for_vis_conf
        self.fnet = BasicEncoder(input_dim=3, output_dim=self.latent_dim, stride=stride)
        self.updateformer = _UpdateFormerParams(self.input_dim, 384, 3, num_virtual_tracks, flow_out=2 if linear_layer_for_vis_conf else 4,
                                                vis_conf_head=bool(linear_layer_for_vis_conf), space_attn=bool(add_space_attn))
        self.corr_mlp = _Mlp(49 * 49, 384, 256)
        self.register_buffer("time_emb", sincos_time_embed(self.input_dim, window_len))
        self._packed = {}  # precision -> PackedWeights
        self._hip_encoder = None  # encoder_hip.HipEncoder for the current device (repacked convolution weights)
        self.max_corr_rows = 262144  # (point,frame) rows of correlation volume resident at once (~10 GB)
        # arithmetic of the Linear layers: "f16x3" = split-half MFMA (3 f16 MFMAs per product, f32 accumulate,
        # fp32-class accuracy at 5.3x the f32-MFMA ceiling), "f32" = exact-f32 MFMA.  Not a reference kwarg.
        self.precision = DEFAULT_PRECISION
        # hipGraph replay of the streaming window (BASELINE.json configs[3]): with is_online=True the whole
        # window (iters x ~190 launches) is captured once per (S, N, iters) and replayed with one graph launch
        # per chunk.  Not a reference kwarg; CoTrackerOnlinePredictor switches it on.
        self.hip_graph = False
        self._graphs = {}
        # f16 range guard of the split-half back end (include/ctk.h "numeric range"): an activation beyond +-65504
        # becomes inf/NaN in its hi/lo halves and reaches the window state as a non-finite value (nothing on the path
        # clamps or masks it), so every forward checks its outputs once and, on a hit, re-runs that forward on the
        # exact-f32 MFMA back end of the same library (fp32 range, as the reference).  range_fallbacks counts hits.
        self.range_guard = True
        self.range_fallbacks = 0
        # graph streaming only: "deferred" (default) examines a chunk's finiteness flag at the start of the NEXT call, so the
        # stream of chunk calls never waits for the GPU -- a hit then raises (the chunk was already returned); "immediate" waits
        # for the flag inside the call (one device-to-host sync per chunk) and, on a hit, restores the online state and re-runs
        # that chunk on the exact-f32 back end before returning it -- the behaviour of every non-streaming path.
        self.stream_range_check = "deferred"
        self.encoder_dtype = torch.float32  # fp32 as the reference; see tools/probe_encoder_precision.py for why not lower
        # "hip" (default): the CNN runs on the library's split-half implicit-GEMM convolutions (encoder_hip.py, csrc/conv_pp.hip,
        # csrc/encoder.hip) -- fp32-class accuracy at 2.2x the speed of MIOpen's fp32 convolutions; "torch": nn.Conv2d on
        # PyTorch-ROCm / MIOpen (the round-1/2 path, kept for A/B and for `encoder_dtype` experiments).  Not a reference kwarg.
        self.encoder_backend = "hip"
        # streaming: reuse the previous chunk's features for the overlapping frames (see _encode_online).  Not a reference
        # kwarg and OFF by default (the reference re-encodes whatever chunk it is given, predictor.py:288-290); opt in for
        # streams whose chunks are overlapping VIEWS of one resident video (see _encode_online: the overlap is proven on the host
        # by storage aliasing; bench.py --workload c4_online --feature-cache)
        self.online_feature_cache = False
        self.encoder_chunk = 16  # frames per CNN call (see _encode); not a reference kwarg
        # pre-sigmoid (visibility, confidence) of the last forward, [B,T,N] each -- parity tests compare logits
        self.last_logits = None

    # -- weights ------------------------------------------------------------------------
    def load_state_dict(self, *args, **kwargs):
        self.invalidate_packed_weights()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed_weights()
        return super()._apply(fn, *args, **kwargs)

    def packed(self, device, precision: Optional[str] = None) -> PackedWeights:
        precision = precision or self.precision
        pw = self._packed.get(precision)
        if pw is None or pw.device != device:
            pw = self._packed[precision] = PackedWeights(self, device, precision)
        return pw

    def invalidate_packed_weights(self):
        """Call after mutating parameters in place (e.g. weights.fill_synthetic_)."""
        self._packed = {}
        self._hip_encoder = None
        self._drop_graphs()  # captured graphs hold pointers into the old packed weights

    def _drop_graphs(self):
        if getattr(self, "_graphs", None):
            # a replay may still be in flight on the current stream: destroying the exec / freeing its private
            # workspace under it is undefined, so drain first
            if torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()
            self._graphs = {}

    # device-side caches (ctypes structs with raw pointers) are rebuilt on demand: keep them out of pickles / deep copies
    _TRANSIENT = {"_packed": dict, "_graphs": dict, "_hip_encoder": type(None), "_pending_range": type(None), "_pending_overlap": type(None),
                  "online_f0_tail": type(None), "_online_prev_frames": type(None), "_overlap_hint": type(None), "_hint_now": type(None)}

    def __getstate__(self):
        # a pending (pinned flag, cuda Event) pair cannot be pickled, and must not be lost: pickling (like deepcopy) a model in
        # the middle of a graph stream waits for the last chunk and may raise FloatingPointError
        self._resolve_deferred_range_check()
        st = self.__dict__.copy()
        for k, mk in self._TRANSIENT.items():
            if k in st:
                st[k] = mk()
        return st

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        # may synchronise and raise FloatingPointError (a pending deferred range check of graph streaming): do it BEFORE the
        # half-built copy is registered in memo, so a raise leaves no partial object behind
        self._resolve_deferred_range_check()
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = self._TRANSIENT[k]() if k in self._TRANSIENT else copy.deepcopy(v, memo)
        return new

    def _guarded(self, run, snapshot=None, restore=None, deferred=False):
        """run(precision) -> (coords, vis_logit, conf_logit) under the f16 range guard described in __init__.
        deferred=True (streaming): the finiteness flag of this call is copied to the host asynchronously and examined at
        the START of the next call, so the stream of chunk calls never waits for the GPU; a hit then raises (the chunk
        that overflowed has already been returned, and the online state is poisoned: the stream must be re-run with
        ``precision="f32"``)."""
        self._resolve_deferred_range_check()
        out = run(self.precision)
        if self.precision == "f16x3" and self.range_guard:
            finite = torch.stack([torch.isfinite(o).all() for o in out]).all()
            if deferred:
                flag = torch.empty((), dtype=torch.bool, pin_memory=True)
                flag.copy_(finite, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._pending_range = (flag, ev)
                return out
            if not bool(finite):
                self.range_fallbacks += 1
                warnings.warn("cotracker_amd: non-finite tracks from the split-half (f16x3) back end -- an activation left "
                              "the f16 range (|x| < 65504) or the input is non-finite; re-running this forward on the "
                              "exact-f32 MFMA back end", RuntimeWarning, stacklevel=3)
                if restore is not None:
                    restore(snapshot)
                out = run("f32")
        return out

    def _resolve_deferred_range_check(self):
        pending, self._pending_range = getattr(self, "_pending_range", None), None
        if pending is not None:
            flag, ev = pending
            ev.synchronize()
            if not bool(flag):
                self.range_fallbacks += 1
                raise FloatingPointError("cotracker_amd: the previous streaming chunk produced non-finite tracks on the "
                                         "split-half (f16x3) back end (an activation left the f16 range |x| < 65504, or the "
                                         "input was non-finite); restart the stream with model.precision = 'f32'")

    def _graphed_window(self, fm, support, coords, vis, conf, mask, iters, pw):
        """Run one window through its captured hipGraph: static buffers are created (and the graph captured) on
        first use of this (shapes, iters, weights) combination, then only refreshed in place and replayed.
        Returns the static coords/vis/conf tensors (overwritten by the next call)."""
        key = (tuple(tuple(f.shape) for f in fm), coords.shape[1], int(iters), id(pw), coords.device.index,
               int(self.max_corr_rows), tuple(self.model_resolution), int(self.stride), bool(getattr(self, "_space_attn", True)))
        g = self._graphs.get(key)
        if g is None:
            st_fm = [f.clone() for f in fm]
            st_sup = [s_.clone() for s_ in support]
            win = ops.Window(st_fm, st_sup, coords.clone(), vis.clone(), conf.clone(), self._scale_xy(), iters=iters,
                             point_mask=mask.clone(), max_corr_rows=self.max_corr_rows, space_attn=getattr(self, "_space_attn", True))
            self._drop_graphs()  # one live graph per model: a new shape replaces the old one (frees its workspace)
            g = ops.WindowGraph(win, pw)
            self._graphs = {key: g}
        else:
            st_fm, st_sup, c_, v_, f_, m_ = g.win.keep
            for d, s_ in zip(st_fm, fm):
                d.copy_(s_)
            for d, s_ in zip(st_sup, support):
                d.copy_(s_)
            c_.copy_(coords)
            v_.copy_(vis)
            f_.copy_(conf)
            m_.copy_(mask)
        g.launch()
        return g.win.keep[2], g.win.keep[3], g.win.keep[4]

    # -- shared pieces ------------------------------------------------------------------
    def _scale_xy(self):
        return (self.model_resolution[1] / self.stride, self.model_resolution[0] / self.stride)

    def _encode(self, frames: torch.Tensor, chunk: int) -> torch.Tensor:
        """frames [T,3,H,W] in 0..255 -> L2-normalised NHWC level-0 features [T,H/4,W/4,128].
        The CNN is per-frame, so the frames go through it `encoder_chunk` at a time whatever `fmaps_chunk_size` the caller
        asked for: 16 frames keep the layer activations (150 MB at 384x512) inside the 256 MB Infinity Cache -- the same
        speed as one 120-frame batch in situ (C3 step 1 521 vs 1 517-1 533 ms; 100 vs 102-130 ms in
        tools/bench_encoder_chunk.py) at a tenth of the activation memory; every chunk is normalised straight into its
        frame range of the output (no torch.cat of the 755 MB feature tensor)."""
        T, _, H, W = frames.shape
        step = max(1, min(int(chunk), int(self.encoder_chunk)))
        out = torch.empty(T, H // self.stride, W // self.stride, self.latent_dim, device=frames.device, dtype=torch.float32)
        if self.encoder_backend == "hip" and self.encoder_dtype == torch.float32:
            enc = getattr(self, "_hip_encoder", None)
            if enc is None or enc.device != frames.device:
                from .encoder_hip import HipEncoder
                enc = self._hip_encoder = HipEncoder(self.fnet, frames.device)
            for t0 in range(0, T, step):
                enc(frames[t0:t0 + step].float().contiguous(), out=out[t0:t0 + step])
            return out
        for t0 in range(0, T, step):
            x = 2 * (frames[t0:t0 + step] / 255.0) - 1.0  # cotracker3_online.py:320
            if self.encoder_dtype == torch.float32:
                f = self.fnet(x)
            else:  # experiment knob (tools/probe_encoder_precision.py): MIOpen convolutions in half precision
                with torch.autocast("cuda", dtype=self.encoder_dtype):
                    f = self.fnet(x)
            ops.normalize_to_nhwc(f.float().contiguous(), out=out[t0:t0 + step])
        return out

    def _support(self, pyr, frames_f: torch.Tensor, qcoords: torch.Tensor):
        return [ops.sample_support(pyr[l], frames_f, (qcoords / 2 ** l).contiguous()) for l in range(self.corr_levels)]

    def _check_inputs(self, video, queries, is_train, add_space_attn=True):
        if is_train:
            raise NotImplementedError("inference-only implementation (training is out of scope)")
        # forward-time flag of the reference (cotracker.py:496-502: `add_space_attn and hasattr(self, "space_virtual_blocks")`)
        self._space_attn = bool(add_space_attn) and getattr(self, "add_space_attn", True)
        if not video.is_cuda:
            raise RuntimeError("cotracker_amd runs on an MI355X GPU only: move the model and inputs to 'cuda'. "
                               "There is no CPU path.")
        B, T, C, H, W = video.shape
        assert H % self.stride == 0 and W % self.stride == 0
        assert queries.shape[0] == B and queries.shape[2] == 3
        return B, T, H, W


class CoTrackerThreeOnline(CoTrackerThreeBase):
    """Sliding-window / streaming tracker (cotracker3_online.py:159-541)."""

    def init_video_online_processing(self):  # cotracker3_online.py:163-169
        self._resolve_deferred_range_check()  # the last chunk of the previous stream (graph streaming defers its check by one call)
        self.online_ind = 0
        self.online_track_feat = [None] * self.corr_levels  # unused by v3, kept for API parity
        self.online_track_support = [None] * self.corr_levels
        self.online_coords_predicted = None
        self.online_vis_predicted = None
        self.online_conf_predicted = None
        self.online_f0_tail = None        # level-0 features of the frames the NEXT chunk starts with (feature cache)
        self._online_prev_frames = None   # the previous chunk itself (a reference), to prove the overlap on the host
        self._overlap_hint = None         # the predictor's verdict for the NEXT call (it resizes chunks into fresh tensors)
        self._pending_overlap = None
        self._pending_range = None
        self._online_batch = None  # B > 1: one state tuple per batch element (the attributes above hold the last one run)

    @torch.no_grad()
    def forward(self, video, queries, iters=4, is_train=False, add_space_attn=True, fmaps_chunk_size=200,
                is_online=False):
        B, T, H, W = self._check_inputs(video, queries, is_train, add_space_attn)
        S = self.window_len
        assert S >= 2
        if is_online:
            assert T <= S, "Online mode: video chunk must be <= window size."
            assert getattr(self, "online_ind", None) is not None, "Call model.init_video_online_processing() first."
        self._hint_now, self._overlap_hint = getattr(self, "_overlap_hint", None), None
        # streaming with the window graph (CoTrackerOnlinePredictor): deferred range check, the chunk stream stays asynchronous
        deferred = bool(is_online and self.hip_graph and B == 1 and self.stream_range_check == "deferred")
        run = lambda b: self._guarded(  # noqa: E731
            lambda prec: self._forward_one(video[b], queries[b], iters, fmaps_chunk_size, is_online, prec),
            self._online_snapshot() if is_online else None, self._online_restore, deferred)
        if is_online and B > 1:
            # the reference carries the batch inside its state tensors; here every batch element owns a state tuple
            # that is swapped in around its (independent) window
            states = self._online_batch if self._online_batch is not None else [self._online_snapshot()] * B
            assert len(states) == B, "batch size changed between online calls"
            outs = []
            for b in range(B):
                self._online_restore(states[b])
                outs.append(run(b))
                states[b] = self._online_snapshot()
            self._online_batch = states
        else:
            outs = [run(b) for b in range(B)]
        coords = torch.stack([o[0] for o in outs])
        vis = torch.stack([o[1] for o in outs])
        conf = torch.stack([o[2] for o in outs])
        self.last_logits = (vis, conf)
        return coords, torch.sigmoid(vis), torch.sigmoid(conf), None

    def _online_snapshot(self):
        return (self.online_ind, list(self.online_track_support), self.online_coords_predicted, self.online_vis_predicted,
                self.online_conf_predicted, self.online_f0_tail, self._online_prev_frames)

    def _online_restore(self, snap):
        if snap is not None:
            (self.online_ind, sup, self.online_coords_predicted, self.online_vis_predicted,
             self.online_conf_predicted, self.online_f0_tail, self._online_prev_frames) = snap
            self.online_track_support = list(sup)

    def _encode_online(self, video, chunk, S, step):
        """Streaming: consecutive chunks overlap by S - step frames (predictor.py:225,288-290 feeds the last 2*step frames
        every step), and the encoder is per-frame, so the overlapping frames' level-0 features are the ones computed one
        call ago.  With ``online_feature_cache`` (opt-in) only the `step` NEW frames go through the CNN (half the encoder
        time of a streaming call) -- when the overlap is PROVEN ON THE HOST, with no device work and no synchronisation: the
        new chunk's first S - step frames must be the very memory of the previous chunk's last S - step frames (same live
        storage, offset advanced by `step` frames, same strides / dtype, tensor version unchanged: `tail_aliases`), which is
        what slicing a resident video gives (``video[:, i:i + S]``, then ``video[:, i + step:i + step + S]``).
        CoTrackerOnlinePredictor applies the same test to the chunk it is handed and passes the verdict down
        (``_overlap_hint``), because it resizes every chunk into a fresh tensor.  A chunk whose overlap cannot be proven
        that way is simply encoded in full, exactly what the reference does with it (round 3 compared the frames on the
        device with torch.equal: a blocking device-to-host wait in every streaming call -- ADVICE r3)."""
        T = video.shape[0]
        ov = S - step
        tail, prev = self.online_f0_tail, self._online_prev_frames
        hint = getattr(self, "_hint_now", None)  # the predictor's verdict for this forward call (all batch elements)
        proven = hint if hint is not None else tail_aliases(prev, video, 0, step)
        if self.online_feature_cache and T == S and tail is not None and tail.shape[0] == ov and proven:
            f0 = torch.cat([tail, self._encode(video[ov:].float(), chunk)], dim=0)
        else:
            f0 = self._encode(video.float(), chunk)
        if self.online_feature_cache and T == S:
            self.online_f0_tail = f0[step:]
            self._online_prev_frames = video  # a reference, not a copy: keeps the storage alive, so "same address" means "same allocation"
            try:
                video._ctk_version = video._version  # the version the cached features were computed from
            except Exception:
                pass
        else:
            self.online_f0_tail = self._online_prev_frames = None
        return f0

    def _forward_one(self, video, queries, iters, chunk, is_online, precision=None):
        T = video.shape[0]
        N = queries.shape[0]
        S = self.window_len
        step = S // 2
        dev = video.device
        pw = self.packed(dev, precision)
        queries = queries.float()
        qframes = queries[:, 0].long()                      # cotracker3_online.py:333
        qcoords = (queries[:, 1:3] / self.stride).contiguous()  # :335-336

        # encoder + pyramid.  The reference pads the *video* by repeating its last frame
        # (:321-328); the encoder is per-frame, so repeating the last feature map is identical.
        pad = (S - T) if is_online else (S - T % S) % S
        f0 = self._encode_online(video, chunk, S, step) if is_online else self._encode(video.float(), chunk)
        if pad > 0:
            f0 = torch.cat([f0, f0[-1:].expand(pad, -1, -1, -1)], dim=0).contiguous()
        pyr = ops.build_pyramid(f0, self.corr_levels)

        coords_pred = torch.zeros(T, N, 2, device=dev)
        vis_pred = torch.zeros(T, N, device=dev)
        conf_pred = torch.zeros(T, N, device=dev)
        if is_online:
            if self.online_coords_predicted is not None:  # :349-360
                p = min(step, T - step)
                coords_pred = F.pad(self.online_coords_predicted, (0, 0, 0, 0, 0, p))
                vis_pred = F.pad(self.online_vis_predicted, (0, 0, 0, p))
                conf_pred = F.pad(self.online_conf_predicted, (0, 0, 0, p))
            left = 0 if self.online_ind == 0 else self.online_ind + step
            right = self.online_ind + S
            sample_mask = ((qframes >= left) & (qframes < right)).float()[:, None, None]  # :411-414
            frames_rel = (qframes - self.online_ind).float().contiguous()
        else:
            frames_rel = qframes.float().contiguous()

        support = self._support(pyr, frames_rel, qcoords)
        if is_online:  # :424-440 -- accumulate only the tracks whose query frame is in this chunk
            for l in range(self.corr_levels):
                if self.online_track_support[l] is None:
                    self.online_track_support[l] = torch.zeros_like(support[l])
                self.online_track_support[l] = self.online_track_support[l] + support[l] * sample_mask
                support[l] = self.online_track_support[l]

        coords_init = qcoords[None].expand(S, N, 2).contiguous()
        vis_init = torch.zeros(S, N, device=dev)
        conf_init = torch.zeros(S, N, device=dev)

        num_windows = (T - S + step - 1) // step + 1
        indices = [self.online_ind] if is_online else range(0, step * num_windows, step)
        for ind in indices:
            if ind > 0:  # carry-over from the previous window, :457-482
                overlap = S - step
                copy_over = (qframes < ind + overlap)[None, :]
                cprev = coords_pred[ind:ind + overlap] / self.stride
                cprev = torch.cat([cprev, cprev[-1:].expand(step, -1, -1)], dim=0)
                vprev = vis_pred[ind:ind + overlap]
                vprev = torch.cat([vprev, vprev[-1:].expand(step, -1)], dim=0)
                fprev = conf_pred[ind:ind + overlap]
                fprev = torch.cat([fprev, fprev[-1:].expand(step, -1)], dim=0)
                coords_init = torch.where(copy_over[..., None], cprev, coords_init).contiguous()
                vis_init = torch.where(copy_over, vprev, vis_init).contiguous()
                conf_init = torch.where(copy_over, fprev, conf_init).contiguous()
            mask = (qframes < ind + S).to(torch.uint8).contiguous()  # attention_mask :484, used as :493-496
            fm = pyr if is_online else [p_[ind:ind + S] for p_ in pyr]
            if is_online and self.hip_graph:
                coords, vis, conf = self._graphed_window(fm, support, coords_init, vis_init, conf_init, mask, iters, pw)
            else:
                coords = coords_init.clone()
                vis = vis_init.clone()
                conf = conf_init.clone()
                win = ops.Window(fm, support, coords, vis, conf, self._scale_xy(), iters=iters, point_mask=mask,
                                 max_corr_rows=self.max_corr_rows, space_attn=getattr(self, "_space_attn", True))
                ops.forward_window(win, pw)
            S_trim = T if is_online else min(T - ind, S)
            coords_pred[ind:ind + S] = (coords * float(self.stride))[:S_trim]
            vis_pred[ind:ind + S] = vis[:S_trim]
            conf_pred[ind:ind + S] = conf[:S_trim]
        if is_online:
            self.online_ind += step
            self.online_coords_predicted = coords_pred
            self.online_vis_predicted = vis_pred
            self.online_conf_predicted = conf_pred
        return coords_pred, vis_pred, conf_pred


class CoTrackerThreeOffline(CoTrackerThreeBase):
    """Single-window tracker over all T frames (cotracker3_offline.py:15-233)."""

    @torch.no_grad()
    def forward(self, video, queries, iters=4, is_train=False, add_space_attn=True, fmaps_chunk_size=200):
        B, T, H, W = self._check_inputs(video, queries, is_train, add_space_attn)
        assert T >= 1
        outs = [self._guarded(lambda prec, b=b: self._forward_one(video[b], queries[b], iters, fmaps_chunk_size, prec))
                for b in range(B)]
        vis, conf = torch.stack([o[1] for o in outs]), torch.stack([o[2] for o in outs])
        self.last_logits = (vis, conf)
        return torch.stack([o[0] for o in outs]), torch.sigmoid(vis), torch.sigmoid(conf), None

    def _forward_one(self, video, queries, iters, chunk, precision=None):
        T = video.shape[0]
        N = queries.shape[0]
        dev = video.device
        pw = self.packed(dev, precision)
        queries = queries.float()
        qframes = queries[:, 0].long()
        qcoords = (queries[:, 1:3] / self.stride).contiguous()
        pyr = ops.build_pyramid(self._encode(video.float(), chunk), self.corr_levels)
        support = self._support(pyr, qframes.float().contiguous(), qcoords)
        coords = qcoords[None].expand(T, N, 2).contiguous()
        vis = torch.zeros(T, N, device=dev)
        conf = torch.zeros(T, N, device=dev)
        win = ops.Window(pyr, support, coords, vis, conf, self._scale_xy(), iters=iters, point_mask=None,
                         max_corr_rows=self.max_corr_rows, space_attn=getattr(self, "_space_attn", True))
        ops.forward_window(win, pw)
        return coords * float(self.stride), vis, conf
