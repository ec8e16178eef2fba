"""Torch-tensor front end of the C-ABI ops (device memory + stream plumbing only).

Every function enqueues HIP kernels from libctk_hip.so on the current torch stream; nothing here
computes on the CPU and nothing falls back to PyTorch ops.
"""
import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expected contiguous float32 CUDA(HIP) tensors")


# ------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """Split a torch-layout Linear weight [N,K] (K % 32 == 0) into the two-half blob of the split-half
    GEMM back end (ctk_pack_weight).  Returns a uint8 device tensor that must outlive its users."""
    _chk_f32(w)
    N, K = w.shape
    nbytes = C.c_size_t(0)
    L.check(L.load().ctk_pack_weight_bytes(N, K, C.byref(nbytes)), "ctk_pack_weight_bytes")
    blob = torch.empty(nbytes.value, device=w.device, dtype=torch.uint8)
    L.check(L.load().ctk_pack_weight(_ptr(w), K, N, K, _ptr(blob), _stream()), "ctk_pack_weight")
    return blob


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, act: int = L.ACT_NONE, resid=None, bias_rows=None,
         out: Optional[torch.Tensor] = None, packed: Optional[torch.Tensor] = None, out_split: bool = False) -> torch.Tensor:
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias + bias_rows[m % period]) + resid.
    packed = pack_weight(w) selects the split-half (3 x f16 MFMA) back end; None the exact-f32 one.
    With the split-half back end, a may be an SH tensor [M,K/32,2,32] float16 (split_rows) and out_split
    returns the result in SH form."""
    a_split = a.dtype == torch.float16
    if a_split:
        assert packed is not None and a.is_cuda and a.is_contiguous() and a.dim() == 4
        _chk_f32(w, bias, bias_rows)
        M, K = a.shape[0], a.shape[1] * 32
    else:
        _chk_f32(a, w, bias, bias_rows)
        M, K = a.shape
    for t_ in (resid, out):  # row-strided views are fine (leading dimension is passed explicitly)
        if t_ is not None and not (t_.is_cuda and t_.dtype == torch.float32 and t_.stride(1) == 1):
            raise ValueError("out/resid must be float32 device tensors with unit column stride")
    N = w.shape[0]
    if out_split:
        assert packed is not None and resid is None and out is None
        out = torch.empty(M, N // 32, 2, 32, device=a.device, dtype=torch.float16)
    elif out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    g = L.GemmArgs()
    g.A, g.lda, g.M = _ptr(a), (2 * K if a_split else K), M
    g.a_split, g.c_split = int(a_split), int(out_split)
    g.W, g.ldw, g.N, g.K = _ptr(w), w.shape[1], N, K
    g.Wp = _ptr(packed)
    g.C, g.ldc = _ptr(out), (2 * N if out_split else out.stride(0))
    g.bias = _ptr(bias)
    g.bias_rows = _ptr(bias_rows)
    g.bias_period = bias_rows.shape[0] if bias_rows is not None else 0
    g.resid, g.ldr = _ptr(resid), (resid.stride(0) if resid is not None else 0)
    g.act = act
    g.batch, g.a_bs, g.c_bs, g.k_valid = 1, 0, 0, 0
    L.check(L.load().ctk_gemm(C.byref(g), _stream()), "ctk_gemm")
    return out


def split_rows(x: torch.Tensor) -> torch.Tensor:
    """f32 [M,K] (K % 32 == 0) -> SH format: float16 tensor [M, K/32, 2, 32] (hi plane, lo plane), x = hi + lo."""
    _chk_f32(x)
    M, K = x.shape
    out = torch.empty(M, K // 32, 2, 32, device=x.device, dtype=torch.float16)
    L.check(L.load().ctk_split_rows(_ptr(x), K, M, K, _ptr(out), _stream()), "ctk_split_rows")
    return out


def unsplit(sh: torch.Tensor) -> torch.Tensor:
    """SH tensor [M, K/32, 2, 32] float16 -> f32 [M,K] (test helper: hi + lo)."""
    M, KT = sh.shape[0], sh.shape[1]
    return (sh[:, :, 0].float() + sh[:, :, 1].float()).reshape(M, KT * 32)


def layernorm(x: torch.Tensor, gamma=None, beta=None, eps: float = 1e-6, out_split: bool = False) -> torch.Tensor:
    _chk_f32(x, gamma, beta)
    assert x.shape[-1] == L.HID
    R = x.numel() // L.HID
    y = torch.empty(R, L.HID // 32, 2, 32, device=x.device, dtype=torch.float16) if out_split else torch.empty_like(x)
    L.check(L.load().ctk_layernorm(_ptr(x), _ptr(y), R, _ptr(gamma), _ptr(beta), float(eps), int(out_split), _stream()),
            "ctk_layernorm")
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, splits: int = 1, out_split: bool = False,
              key_mask: Optional[torch.Tensor] = None, query_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [B,N1,384], k/v [B,N2,384] (8 heads x 48, heads contiguous in the last dim) -> [B,N1,384]
    (or its SH form [B*N1, 12, 2, 32] float16 when out_split)."""
    _chk_f32(q, k, v)
    B, N1, _ = q.shape
    N2 = k.shape[1]
    out = torch.empty(B * N1, L.HID // 32, 2, 32, device=q.device, dtype=torch.float16) if out_split else torch.empty_like(q)
    a = L.AttnArgs()
    a.q, a.q_ld, a.q_bs, a.q_is = _ptr(q), L.HID, N1, 1
    a.k, a.v, a.kv_ld, a.kv_bs, a.kv_is = _ptr(k), _ptr(v), L.HID, N2, 1
    a.out, a.o_ld, a.o_bs, a.o_is = _ptr(out), (2 * L.HID if out_split else L.HID), N1, 1
    a.o_split = int(out_split)
    a.nbatch, a.n1, a.n2 = B, N1, N2
    a.splits = splits
    part = None
    if splits > 1:
        part = torch.empty(splits * B * 8 * N1 * 50, device=q.device, dtype=torch.float32)
    a.partial = _ptr(part)
    a.key_mask, a.query_mask = _ptr(key_mask), _ptr(query_mask)  # uint8 [N2] / [N1] (CoTracker2 attention mask)
    L.check(L.load().ctk_attention(C.byref(a), _stream()), "ctk_attention")
    return out


# ------------------------------------------------------------------------------------------
# pyramid / samplers
# ------------------------------------------------------------------------------------------
def normalize_to_nhwc(fmaps_nchw: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[F,128,H,W] -> channel-L2-normalised NHWC [F,H,W,128] (cotracker3_online.py:384-394); `out` = a contiguous
    [F,H,W,128] destination (e.g. a frame range of a preallocated feature tensor)."""
    _chk_f32(fmaps_nchw)
    F_, Cc, H, W = fmaps_nchw.shape
    assert Cc == 128
    if out is None:
        out = torch.empty(F_, H, W, Cc, device=fmaps_nchw.device, dtype=torch.float32)
    else:
        _chk_f32(out)
        assert out.shape == (F_, H, W, Cc)
    L.check(L.load().ctk_normalize_to_nhwc(_ptr(fmaps_nchw), F_, H, W, _ptr(out), _stream()), "ctk_normalize_to_nhwc")
    return out


def avg_pool2_nhwc(x: torch.Tensor) -> torch.Tensor:
    _chk_f32(x)
    F_, H, W, Cc = x.shape
    out = torch.empty(F_, H // 2, W // 2, Cc, device=x.device, dtype=torch.float32)
    L.check(L.load().ctk_avg_pool2_nhwc(_ptr(x), F_, H, W, _ptr(out), _stream()), "ctk_avg_pool2_nhwc")
    return out


def build_pyramid(level0_nhwc: torch.Tensor, levels: int = L.LEVELS) -> List[torch.Tensor]:
    pyr = [level0_nhwc]
    for _ in range(levels - 1):
        pyr.append(avg_pool2_nhwc(pyr[-1]))
    return pyr


def sample_support(fmap_nhwc: torch.Tensor, frames: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """get_track_feat (cotracker3_online.py:113-128): fmap [T,H,W,128], frames [N] float, coords [N,2] -> [N,49,128]."""
    _chk_f32(fmap_nhwc, frames, coords)
    T, H, W, _ = fmap_nhwc.shape
    N = coords.shape[0]
    out = torch.empty(N, 49, 128, device=coords.device, dtype=torch.float32)
    L.check(L.load().ctk_sample_support(_ptr(fmap_nhwc), T, H, W, _ptr(frames), _ptr(coords), N, _ptr(out), _stream()),
            "ctk_sample_support")
    return out


def sample_patches(fmap_nhwc: torch.Tensor, coords: torch.Tensor, level: int) -> torch.Tensor:
    """get_correlation_feat (cotracker3_online.py:130-143): fmap [S,H,W,128], coords [S,N,2] level-0 -> [S,N,49,128]."""
    _chk_f32(fmap_nhwc, coords)
    S, H, W, _ = fmap_nhwc.shape
    N = coords.shape[1]
    out = torch.empty(S, N, 49, 128, device=coords.device, dtype=torch.float32)
    L.check(L.load().ctk_sample_patches(_ptr(fmap_nhwc), S, H, W, _ptr(coords), N, level, _ptr(out), _stream()),
            "ctk_sample_patches")
    return out


def corrblock_sample(pyr_nhwc: Sequence[torch.Tensor], targets: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """CorrBlock.corr + .sample fused (blocks.py:309-362): pyr_nhwc 4 x [S,H_l,W_l,128], targets [S,N,128],
    coords [S,N,2] (level-0 units) -> [N,S,196].  The correlation volume is never materialised."""
    _chk_f32(*pyr_nhwc, targets, coords)
    assert len(pyr_nhwc) == L.LEVELS
    S, N = coords.shape[0], coords.shape[1]
    assert targets.shape == (S, N, 128) and coords.shape == (S, N, 2)
    fm = (C.c_void_p * L.LEVELS)(*[_ptr(f) for f in pyr_nhwc])
    Hs = (C.c_int32 * L.LEVELS)(*[f.shape[1] for f in pyr_nhwc])
    Ws = (C.c_int32 * L.LEVELS)(*[f.shape[2] for f in pyr_nhwc])
    for f in pyr_nhwc:
        assert f.shape[0] == S and f.shape[3] == 128
    out = torch.empty(N, S, L.LEVELS * 49, device=coords.device, dtype=torch.float32)
    L.check(L.load().ctk_corrblock_sample(fm, Hs, Ws, S, N, _ptr(targets), _ptr(coords), _ptr(out), _stream()),
            "ctk_corrblock_sample")
    return out



# ------------------------------------------------------------------------------------------
# CoTracker2 iteration (cotracker.py:86-173): row kernels + the general update former
# ------------------------------------------------------------------------------------------
def sample_features4d(map_hwc: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """sample_features4d (model_utils.py:258-290): channels-last map [H,W,C] sampled at coords [N,2]=(x,y) -> [N,C]."""
    _chk_f32(map_hwc, coords)
    H, W, Cc = map_hwc.shape
    N = coords.shape[0]
    out = torch.empty(N, Cc, device=coords.device, dtype=torch.float32)
    L.check(L.load().ctk_sample_features4d(_ptr(map_hwc), H, W, Cc, _ptr(coords), N, _ptr(out), _stream()), "ctk_sample_features4d")
    return out


def bilinear_sampler(input: torch.Tensor, coords: torch.Tensor, align_corners: bool = True, padding_mode: str = "border") -> torch.Tensor:
    """Op D: bilinear_sampler (model_utils.py:191-255), the reference's signature and layouts -- input [B,C,H,W] with coords
    [B,Ho,Wo,2] = (x, y) -> [B,C,Ho,Wo], or input [B,C,T,H,W] with coords [B,Do,Ho,Wo,3] = (t, x, y) -> [B,C,Do,Ho,Wo].
    Bit-identical to the reference's F.grid_sample on the CPU (ctk_bilinear_sampler, csrc/sampler.hip)."""
    input, coords = input.contiguous(), coords.contiguous()  # (the reference's own unit test hands over permuted coordinates)
    _chk_f32(input, coords)
    if padding_mode not in ("zeros", "border"):
        raise NotImplementedError(f"padding_mode={padding_mode!r}: the HIP sampler implements 'zeros' and 'border'")
    sizes = input.shape[2:]
    assert len(sizes) in (2, 3), "input must be [B,C,H,W] or [B,C,T,H,W]"
    nd = len(sizes)
    assert coords.dim() == nd + 2 and coords.shape[-1] == nd and coords.shape[0] == input.shape[0]
    B, Cc = input.shape[:2]
    D = sizes[0] if nd == 3 else 0
    H, W = sizes[-2], sizes[-1]
    inner = tuple(coords.shape[1:-1])
    P = 1
    for d in inner:
        P *= d
    out = torch.empty((B, Cc) + inner, device=input.device, dtype=torch.float32)
    if P > 0:
        L.check(L.load().ctk_bilinear_sampler(_ptr(input), B, Cc, D, H, W, _ptr(coords), P, int(bool(align_corners)),
                                              L.PAD_BORDER if padding_mode == "border" else L.PAD_ZEROS, _ptr(out), _stream()),
                "ctk_bilinear_sampler")
    return out


def v2_assemble(coords, fcorrs, track_feat, track_mask, vis, pos, in_ld: int, out_split: bool) -> torch.Tensor:
    """Transformer input of CoTracker2 (cotracker.py:135-150 without the time embedding): [N*S, in_ld] f32 or SH."""
    _chk_f32(coords, fcorrs, track_feat, track_mask, vis, pos)
    S, N = coords.shape[0], coords.shape[1]
    assert fcorrs.shape == (N, S, 196) and track_feat.shape == (S, N, 128) and pos.shape == (N, 456)
    assert track_mask.shape == (S, N) and vis.shape == (S, N)
    x = (torch.empty(N * S, in_ld // 32, 2, 32, device=coords.device, dtype=torch.float16) if out_split
         else torch.empty(N * S, in_ld, device=coords.device, dtype=torch.float32))
    L.check(L.load().ctk_v2_assemble(S, N, _ptr(coords), _ptr(fcorrs), _ptr(track_feat), _ptr(track_mask), _ptr(vis), _ptr(pos),
                                     in_ld, _ptr(x), int(out_split), _stream()), "ctk_v2_assemble")
    return x


def v2_apply_delta(delta: torch.Tensor, coords: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """coords [S,N,2] += delta[:, :2] in place; returns GroupNorm(1,128)(delta[:, 2:130]) as [S*N,128] (row t*N+n)."""
    _chk_f32(delta, coords, gamma, beta)
    S, N = coords.shape[0], coords.shape[1]
    assert delta.shape[0] == N * S
    normed = torch.empty(S * N, 128, device=coords.device, dtype=torch.float32)
    L.check(L.load().ctk_v2_apply_delta(S, N, _ptr(delta), delta.shape[1], _ptr(coords), _ptr(gamma), _ptr(beta), float(eps),
                                        _ptr(normed), _stream()), "ctk_v2_apply_delta")
    return normed


def v2_vis_head(track_feat: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """vis_predictor (cotracker.py:172): track_feat [S,N,128] -> logits [S,N]."""
    _chk_f32(track_feat, w, b)
    S, N, _ = track_feat.shape
    out = torch.empty(S, N, device=track_feat.device, dtype=torch.float32)
    L.check(L.load().ctk_v2_vis_head(_ptr(track_feat), S * N, _ptr(w), _ptr(b), _ptr(out), _stream()), "ctk_v2_vis_head")
    return out


def update_former_ex(x: torch.Tensor, x_split: bool, S: int, N: int, fw: "L.FormerWeights", point_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """General EfficientUpdateFormer.forward (cotracker.py:483-531) with the CoTracker2 attention mask:
    x [N*S, in_ld] (f32 or SH) -> delta [N*S, out_ld] f32."""
    lib = L.load()
    nbytes = C.c_size_t(0)
    L.check(lib.ctk_update_former_workspace_bytes(S, N, C.byref(nbytes)), "ctk_update_former_workspace_bytes")
    ws = _workspace(nbytes.value, x.device)
    delta = torch.empty(N * S, fw.out_ld, device=x.device, dtype=torch.float32)
    if point_mask is not None:
        assert point_mask.dtype == torch.uint8 and point_mask.shape == (N,) and point_mask.is_cuda
    L.check(lib.ctk_update_former_ex(S, N, _ptr(x), int(x_split), C.byref(fw), _ptr(point_mask), _ptr(delta), _ptr(ws), ws.numel(),
                                     _stream()), "ctk_update_former_ex")
    return delta

class V2Window:
    """ctypes ctk_v2_window_args of one CoTracker2 window plus the tensors it points to (coords / track_feat are
    updated in place by forward_window_v2, vis_out receives the visibility logits)."""

    def __init__(self, pyr: Sequence[torch.Tensor], coords: torch.Tensor, track_feat: torch.Tensor, vis: torch.Tensor,
                 track_mask: torch.Tensor, point_mask: Optional[torch.Tensor], iters: int):
        _chk_f32(*pyr, coords, track_feat, vis, track_mask)
        S, N = coords.shape[0], coords.shape[1]
        assert coords.shape == (S, N, 2) and track_feat.shape == (S, N, 128) and vis.shape == (S, N) and track_mask.shape == (S, N)
        a = L.V2WindowArgs()
        a.S, a.N, a.iters = S, N, iters
        for l in range(L.LEVELS):
            assert pyr[l].shape[0] == S and pyr[l].shape[3] == 128
            a.H[l], a.W[l], a.fmaps[l] = pyr[l].shape[1], pyr[l].shape[2], _ptr(pyr[l])
        if point_mask is not None:
            assert point_mask.dtype == torch.uint8 and point_mask.is_cuda and point_mask.shape == (N,)
        self.vis_out = torch.empty(S, N, device=coords.device, dtype=torch.float32)
        a.coords, a.track_feat, a.vis, a.track_mask = _ptr(coords), _ptr(track_feat), _ptr(vis), _ptr(track_mask)
        a.point_mask, a.vis_out = _ptr(point_mask), _ptr(self.vis_out)
        self.args = a
        self.S, self.N = S, N
        self.keep = (list(pyr), coords, track_feat, vis, track_mask, point_mask)
        self.device = coords.device


def forward_window_v2(win: V2Window, weights) -> None:
    """CoTracker2.forward_window (cotracker.py:86-173) as one C call: `iters` iterations in place on win's coords /
    track_feat, visibility logits into win.vis_out."""
    lib = L.load()
    nbytes = C.c_size_t(0)
    L.check(lib.ctk_forward_window_v2_workspace_bytes(C.byref(win.args), C.byref(weights.struct), C.byref(nbytes)),
            "ctk_forward_window_v2_workspace_bytes")
    ws = _workspace(nbytes.value, win.device)
    L.check(lib.ctk_forward_window_v2(C.byref(win.args), C.byref(weights.struct), _ptr(ws), ws.numel(), _stream()),
            "ctk_forward_window_v2")


class V2WindowGraph:
    """hipGraph of one CoTracker2 window (ctk_v2_window_graph_create): same contract as WindowGraph -- pointers of the
    window's tensors, the weights and a private workspace are baked in; refresh contents in place, then launch()."""

    def __init__(self, win: V2Window, weights):
        lib = L.load()
        nbytes = C.c_size_t(0)
        L.check(lib.ctk_forward_window_v2_workspace_bytes(C.byref(win.args), C.byref(weights.struct), C.byref(nbytes)),
                "ctk_forward_window_v2_workspace_bytes")
        self.win, self.weights = win, weights
        self.ws = torch.empty(nbytes.value, device=win.device, dtype=torch.uint8)
        # one direct iteration first so that every kernel's code object is resident (no lazy loads inside a capture)
        state = (win.keep[1], win.keep[2])
        saved = [t_.clone() for t_ in state]
        iters, win.args.iters = win.args.iters, 1
        L.check(lib.ctk_forward_window_v2(C.byref(win.args), C.byref(weights.struct), _ptr(self.ws), self.ws.numel(), _stream()),
                "ctk_forward_window_v2")
        win.args.iters = iters
        for t_, s_ in zip(state, saved):
            t_.copy_(s_)
        torch.cuda.synchronize(win.device)
        h = C.c_void_p()
        L.check(lib.ctk_v2_window_graph_create(C.byref(win.args), C.byref(weights.struct), _ptr(self.ws), self.ws.numel(), C.byref(h)),
                "ctk_v2_window_graph_create")
        self._h = h
        n = C.c_int64(0)
        L.check(lib.ctk_window_graph_nodes(self._h, C.byref(n)), "ctk_window_graph_nodes")
        self.nodes = n.value

    def launch(self) -> None:
        L.check(L.load().ctk_window_graph_launch(self._h, _stream()), "ctk_window_graph_launch")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().ctk_window_graph_destroy(h)
            except Exception:
                pass


# ------------------------------------------------------------------------------------------
# window-level ops
# ------------------------------------------------------------------------------------------
class Window:
    """Holds the ctypes ctk_window_args plus the tensors it points to."""

    def __init__(self, fmaps: Sequence[torch.Tensor], support: Sequence[torch.Tensor], coords: torch.Tensor,
                 vis: torch.Tensor, conf: torch.Tensor, scale_xy, iters: int = 6,
                 point_mask: Optional[torch.Tensor] = None, max_corr_rows: int = 262144, use_aux_stream: bool = True,
                 space_attn: bool = True):
        _chk_f32(*fmaps, *support, coords, vis, conf)
        S, N = coords.shape[0], coords.shape[1]
        assert coords.shape == (S, N, 2) and vis.shape == (S, N) and conf.shape == (S, N)
        a = L.WindowArgs()
        a.S, a.N, a.iters = S, N, iters
        for l in range(L.LEVELS):
            assert fmaps[l].shape[0] == S and fmaps[l].shape[3] == 128
            assert support[l].shape == (N, 49, 128)
            a.H[l], a.W[l] = fmaps[l].shape[1], fmaps[l].shape[2]
            a.fmaps[l] = _ptr(fmaps[l])
            a.support[l] = _ptr(support[l])
        if point_mask is not None:
            assert point_mask.dtype == torch.uint8 and point_mask.is_cuda and point_mask.shape == (N,)
        a.point_mask = _ptr(point_mask)
        a.coords, a.vis, a.conf = _ptr(coords), _ptr(vis), _ptr(conf)
        a.scale_x, a.scale_y = float(scale_xy[0]), float(scale_xy[1])
        a.points_per_chunk = max(1, min(N, max_corr_rows // S))
        a.aux_stream = aux_stream(coords.device).cuda_stream if use_aux_stream else None
        a.flags = 0 if space_attn else L.WINDOW_NO_SPACE_ATTN  # add_space_attn=False (cotracker.py:496-502)
        self.args = a
        self.S, self.N = S, N
        self.keep = (list(fmaps), list(support), coords, vis, conf, point_mask)
        self.device = coords.device


_ws_cache = {}
_aux_streams = {}


def aux_stream(device) -> torch.cuda.Stream:
    """The per-device auxiliary stream handed to ctk_forward_window (ctk_window_args.aux_stream): the library forks
    independent launches onto it and joins it back, so callers never synchronise with it themselves."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _aux_streams:
        _aux_streams[key] = torch.cuda.Stream(device=device)
    return _aux_streams[key]



def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        _ws_cache.pop(key, None)
        buf = None
        buf = torch.empty(nbytes, device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


def forward_window(win: Window, weights) -> None:
    """`iters` update iterations in place on win's coords/vis/conf (cotracker3_online.py:171-264)."""
    lib = L.load()
    nbytes = C.c_size_t(0)
    L.check(lib.ctk_forward_window_workspace_bytes(C.byref(win.args), C.byref(nbytes)), "ctk_forward_window_workspace_bytes")
    ws = _workspace(nbytes.value, win.device)
    mw = weights.struct_for(win.S)
    L.check(lib.ctk_forward_window(C.byref(win.args), C.byref(mw), _ptr(ws), ws.numel(), _stream()), "ctk_forward_window")


class WindowGraph:
    """hipGraph of one whole window (all `iters` iterations, ~190 launches each) captured once by
    ctk_window_graph_create and replayed with ONE graph launch per call (BASELINE.json configs[3]).

    The graph bakes in the pointers of ``win``'s tensors, of the weights and of a private workspace, so the
    caller refreshes the CONTENTS of win's tensors in place (``copy_``) and calls ``launch()``."""

    def __init__(self, win: Window, weights):
        lib = L.load()
        nbytes = C.c_size_t(0)
        L.check(lib.ctk_forward_window_workspace_bytes(C.byref(win.args), C.byref(nbytes)), "ctk_forward_window_workspace_bytes")
        self.win = win
        self.weights = weights                      # keeps every weight tensor (and in_bias_t for this S) alive
        self.ws = torch.empty(nbytes.value, device=win.device, dtype=torch.uint8)  # private: its address is baked in
        mw = weights.struct_for(win.S)
        h = C.c_void_p()
        # One direct iteration first: every kernel of the window is resident (HIP loads code objects lazily, which
        # is not allowed inside a capture); the state it touched is restored afterwards.
        state = win.keep[2:5]
        saved = [t_.clone() for t_ in state]
        iters, win.args.iters = win.args.iters, 1
        L.check(lib.ctk_forward_window(C.byref(win.args), C.byref(mw), _ptr(self.ws), self.ws.numel(), _stream()),
                "ctk_forward_window")
        win.args.iters = iters
        for t_, s_ in zip(state, saved):
            t_.copy_(s_)
        torch.cuda.synchronize(win.device)          # weight packing / input copies issued so far are complete
        L.check(lib.ctk_window_graph_create(C.byref(win.args), C.byref(mw), _ptr(self.ws), self.ws.numel(), C.byref(h)),
                "ctk_window_graph_create")
        self._h = h
        n = C.c_int64(0)
        L.check(lib.ctk_window_graph_nodes(self._h, C.byref(n)), "ctk_window_graph_nodes")
        self.nodes = n.value

    def launch(self) -> None:
        L.check(L.load().ctk_window_graph_launch(self._h, _stream()), "ctk_window_graph_launch")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                L.load().ctk_window_graph_destroy(h)
            except Exception:
                pass


def corr_volume(win: Window) -> torch.Tensor:
    out = torch.empty(L.LEVELS, win.N * win.S, L.CORR_LD, device=win.device, dtype=torch.float32)
    L.check(L.load().ctk_corr_volume(C.byref(win.args), _ptr(out), _stream()), "ctk_corr_volume")
    return out


def corr_volume_sh(win: Window) -> torch.Tensor:
    """Split-half sampler: SH volumes [4, N*S, 76, 2, 32] float16 (use unsplit on a level to compare)."""
    lib = L.load()
    out = torch.empty(L.LEVELS, win.N * win.S, L.CORR_LD // 32, 2, 32, device=win.device, dtype=torch.float16)
    nbytes = C.c_size_t(0)
    L.check(lib.ctk_corr_volume_sh_workspace_bytes(C.byref(win.args), C.byref(nbytes)), "ctk_corr_volume_sh_workspace_bytes")
    ws = _workspace(nbytes.value, win.device)
    L.check(lib.ctk_corr_volume_sh(C.byref(win.args), _ptr(out), _ptr(ws), ws.numel(), _stream()), "ctk_corr_volume_sh")
    return out


def corr_embed(win: Window, weights, x: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = L.load()
    if x is None:
        x = torch.zeros(win.N * win.S, L.X_LD, device=win.device, dtype=torch.float32)
    nbytes = C.c_size_t(0)
    L.check(lib.ctk_corr_embed_workspace_bytes(C.byref(win.args), C.byref(nbytes)), "ctk_corr_embed_workspace_bytes")
    ws = _workspace(nbytes.value, win.device)
    mw = weights.struct_for(win.S)
    L.check(lib.ctk_corr_embed(C.byref(win.args), C.byref(mw), _ptr(x), _ptr(ws), ws.numel(), _stream()), "ctk_corr_embed")
    return x


def assemble_tokens(win: Window, x: torch.Tensor) -> torch.Tensor:
    L.check(L.load().ctk_assemble_tokens(C.byref(win.args), _ptr(x), 0, _stream()), "ctk_assemble_tokens")
    return x


def tap_indices(win: Window) -> torch.Tensor:
    out = torch.empty(win.S, win.N, L.LEVELS, 2, 7, device=win.device, dtype=torch.int32)
    L.check(L.load().ctk_tap_indices(C.byref(win.args), _ptr(out), _stream()), "ctk_tap_indices")
    return out


def update_former(x: torch.Tensor, S: int, N: int, weights) -> torch.Tensor:
    """x [N*S, 1120] (our column layout) -> delta [N*S,4]."""
    _chk_f32(x)
    lib = L.load()
    nbytes = C.c_size_t(0)
    L.check(lib.ctk_update_former_workspace_bytes(S, N, C.byref(nbytes)), "ctk_update_former_workspace_bytes")
    ws = _workspace(nbytes.value, x.device)
    delta = torch.empty(N * S, 4, device=x.device, dtype=torch.float32)
    mw = weights.struct_for(S)
    L.check(lib.ctk_update_former(S, N, _ptr(x), C.byref(mw), _ptr(delta), _ptr(ws), ws.numel(), _stream()), "ctk_update_former")
    return delta


# ------------------------------------------------------------------------------------------
# opt-in per-kernel timing (bench.py)
# ------------------------------------------------------------------------------------------
def profile_enable(on: bool) -> None:
    L.check(L.load().ctk_profile_enable(1 if on else 0), "ctk_profile_enable")


def profile_read():
    rows = (L.ProfileRow * 64)()
    n = C.c_int(0)
    L.check(L.load().ctk_profile_read(rows, 64, C.byref(n)), "ctk_profile_read")
    return [dict(name=rows[i].name.decode(), launches=rows[i].launches, total_ms=rows[i].total_ms, flops=rows[i].flops,
                 bytes=rows[i].bytes) for i in range(n.value)]
