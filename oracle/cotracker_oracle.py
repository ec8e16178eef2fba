"""CPU oracle (numpy, float32) for CoTracker3's iterative-update hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``co-tracker_amd/``) never imports anything from
``oracle/`` and fails loudly when the HIP library is missing.

It restates, function by function, the algorithm of the reference
(facebookresearch/co-tracker @ 2025-03-04); every function cites the
reference file:line it follows.  Arithmetic that lives in the reference's
third-party dependency (PyTorch 2.10.0 ATen: ``grid_sampler_3d``,
``layer_norm``, ``softmax``, ``gelu``, ``avg_pool2d``, ``upsample_linear1d``)
is restated from its published semantics.

Pinning: ``tests/golden/*.npz`` hold outputs of the *unmodified reference*
executed in the build container (torch 2.10.0+rocm7.0, CPU, fp32) by
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` checks every
function here against them (sampler: bit-exact; contractions: <=2e-5 abs).
The reference's own test (tests/test_bilinear_sample.py:16-47, identity
sampling) is replayed in ``tests/test_sampler_identity.py``.

Layout conventions follow the reference: fmaps [B,S,C,H,W], coords [B,S,N,2]
in level-0 feature units (pixels / stride), support features [B,49,N,C].
"""
from __future__ import annotations

import math

import numpy as np

try:  # exact erf for nn.GELU() (blocks.py:48 default act_layer)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover - scipy is present in the image
    _erf = np.vectorize(math.erf, otypes=[np.float64])

f32 = np.float32


# --------------------------------------------------------------------------
# a-1  bilinear_sampler (5-D path)       model_utils.py:191-255
#      + ATen grid_sampler_3d CPU semantics (align_corners=True, border)
# --------------------------------------------------------------------------
def _unnormalize_border(c, size):
    """model_utils.py:242-251 then ATen GridSampler.h:27-36,58-60.

    c * f32(2/max(size-1,1)); c -= 1; ((c+1)/2)*(size-1); clip to [0,size-1].
    All steps are separate float32 roundings (no FMA contraction).
    """
    s = f32(2.0 / max(size - 1, 1))
    g = (c.astype(f32) * s).astype(f32)
    g = (g - f32(1)).astype(f32)
    u = ((g + f32(1)).astype(f32) / f32(2)).astype(f32)
    u = (u * f32(size - 1)).astype(f32)
    return np.minimum(f32(size - 1), np.maximum(u, f32(0))).astype(f32)


def sampler_indices_weights(c, size):
    """floor index, weight of index, weight of index+1 for one axis."""
    u = _unnormalize_border(c, size)
    i0 = np.floor(u).astype(f32)
    w1 = (u - i0).astype(f32)
    w0 = ((i0 + f32(1)) - u).astype(f32)
    return i0.astype(np.int64), w0, w1


def bilinear_sampler_5d(inp, coords):
    """bilinear_sampler(input[B,C,D,H,W], coords[B,...,3]=(t,x,y)) -> [B,C,...].

    model_utils.py:238-255 (align_corners=True, padding_mode="border").
    Corner order z0{(x0,y0),(x1,y0),(x0,y1),(x1,y1)} then z1, weight
    (wx*wy)*wz, out-of-range corners skipped, plain mul+add accumulation
    (verified bit-identical to F.grid_sample on CPU).
    """
    inp = np.asarray(inp, dtype=f32)
    coords = np.asarray(coords, dtype=f32)
    B, C, D, H, W = inp.shape
    zi, wz0, wz1 = sampler_indices_weights(coords[..., 0], D)
    xi, wx0, wx1 = sampler_indices_weights(coords[..., 1], W)
    yi, wy0, wy1 = sampler_indices_weights(coords[..., 2], H)
    out = np.zeros((B, C) + coords.shape[1:-1], dtype=f32)
    cl = np.ascontiguousarray(np.moveaxis(inp, 1, -1))  # channel-last copy: gathers read contiguous rows
    for b in range(B):
        acc = np.zeros(coords.shape[1:-1] + (C,), dtype=f32)
        for dz, wz in ((0, wz0), (1, wz1)):
            if dz == 1 and D == 1:
                continue  # z1 = 1 > D-1: corner out of range, skipped by ATen
            for dy, wy in ((0, wy0), (1, wy1)):
                for dx, wx in ((0, wx0), (1, wx1)):
                    X = xi[b] + dx
                    Y = yi[b] + dy
                    Z = zi[b] + dz
                    ok = (X <= W - 1) & (Y <= H - 1) & (Z <= D - 1)
                    w = ((wx[b] * wy[b]).astype(f32) * wz[b]).astype(f32)
                    v = cl[b][np.minimum(Z, D - 1), np.minimum(Y, H - 1), np.minimum(X, W - 1)]
                    acc = np.where(ok[..., None], (acc + (v * w[..., None]).astype(f32)).astype(f32), acc)
        out[b] = np.moveaxis(acc, -1, 0)
    return out


def sample_features5d(inp, coords):
    """model_utils.py:293-323: input [B,T,C,H,W], coords [B,R1,R2,3] -> [B,R1,R2,C]."""
    x = np.transpose(np.asarray(inp, dtype=f32), (0, 2, 1, 3, 4))
    feats = bilinear_sampler_5d(x, np.asarray(coords, dtype=f32)[:, :, :, None, :])
    return np.ascontiguousarray(np.transpose(feats, (0, 2, 3, 1, 4))[..., 0])


# --------------------------------------------------------------------------
# a-2  get_support_points                cotracker3_online.py:94-111
# --------------------------------------------------------------------------
def get_support_points(coords, r, reshape_back=True):
    """coords [B,1,N,3]=(t,x,y) -> lattice; first 7-index = x offset, second = y."""
    coords = np.asarray(coords, dtype=f32)
    B, _, N, _ = coords.shape
    centroid = coords.reshape(B, N, 1, 1, 3)
    d = np.linspace(-r, r, 2 * r + 1, dtype=f32)
    xgrid, ygrid = np.meshgrid(d, d, indexing="ij")
    delta = np.stack([np.zeros_like(xgrid), xgrid, ygrid], axis=-1).astype(f32)
    lvl = (centroid + delta.reshape(1, 1, 2 * r + 1, 2 * r + 1, 3)).astype(f32)
    if reshape_back:
        return np.transpose(lvl.reshape(B, N, (2 * r + 1) ** 2, 3), (0, 2, 1, 3))
    return lvl


# --------------------------------------------------------------------------
# a-4  get_track_feat                    cotracker3_online.py:113-128
# --------------------------------------------------------------------------
def get_track_feat(fmaps, queried_frames, queried_coords, support_radius=3):
    """fmaps [B,T,C,H,W]; frames [B,N] (int); coords [B,N,2] -> support [B,49,N,C]."""
    fr = np.asarray(queried_frames).astype(f32)[:, None, :, None]
    sc = np.concatenate([fr, np.asarray(queried_coords, dtype=f32)[:, None]], axis=-1)
    pts = get_support_points(sc, support_radius)
    return sample_features5d(fmaps, pts)


# --------------------------------------------------------------------------
# a-3  get_correlation_feat              cotracker3_online.py:130-143
# --------------------------------------------------------------------------
def get_correlation_feat(fmaps, coords, r=3):
    """fmaps [B,T,C,H,W], coords [B*T,N,2] -> [B,T,N,7,7,C]."""
    fmaps = np.asarray(fmaps, dtype=f32)
    B, T, C, H, W = fmaps.shape
    coords = np.asarray(coords, dtype=f32)
    N = coords.shape[1]
    sc = np.concatenate([np.zeros_like(coords[..., :1]), coords], axis=-1)[:, None]
    pts = get_support_points(sc, r, reshape_back=False)  # [B*T,N,7,7,3]
    feat = bilinear_sampler_5d(fmaps.reshape(B * T, C, 1, H, W), pts)
    feat = feat.reshape(B, T, C, N, 2 * r + 1, 2 * r + 1)
    return np.transpose(feat, (0, 1, 3, 4, 5, 2))


def sampler_floor_indices(coords_lvl, H, W, r=3):
    """Integer (x0,y0) of every lattice tap, as ATen computes them.

    coords_lvl [...,2] (x,y) in that level's units -> x0 [...,7], y0 [...,7]
    (tap (i,j) uses x0[i], y0[j]).  The bit-exact contract of BASELINE.md §2.
    """
    d = np.linspace(-r, r, 2 * r + 1, dtype=f32)
    x = (np.asarray(coords_lvl, dtype=f32)[..., 0:1] + d).astype(f32)
    y = (np.asarray(coords_lvl, dtype=f32)[..., 1:2] + d).astype(f32)
    x0, _, _ = sampler_indices_weights(x, W)
    y0, _, _ = sampler_indices_weights(y, H)
    return x0, y0


# --------------------------------------------------------------------------
# a-5  49x49 correlation                 cotracker3_online.py:196-204
# --------------------------------------------------------------------------
def corr_volume(corr_feat, support):
    """corr_feat [B,T,N,7,7,C], support [B,49,N,C] -> [B,T,N,2401] (h,w,i,j row-major)."""
    B, T, N = corr_feat.shape[:3]
    C = corr_feat.shape[-1]
    a = np.asarray(corr_feat, dtype=f32).reshape(B, T, N, 49, C)
    s = np.transpose(np.asarray(support, dtype=f32), (0, 2, 1, 3))  # [B,N,49,C]
    vol = np.matmul(a, np.swapaxes(s, -1, -2)[:, None]).astype(f32)  # [B,T,N,49,49] (BLAS)
    return vol.reshape(B, T, N, 49 * 49)


# --------------------------------------------------------------------------
# a-1 (4-D path) bilinear_sampler -> ATen grid_sampler_2d   model_utils.py:191-255
# a-9  CorrBlock (CoTracker2)                              blocks.py:284-362
# --------------------------------------------------------------------------
def _fma32(a, b, c):
    """float32 fused multiply-add: the f32 x f32 product is exact in float64."""
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def bilinear_sampler_4d(inp, coords):
    """bilinear_sampler(input[B,C,H,W], coords[B,Ho,Wo,2]=(x,y)) -> [B,C,Ho,Wo]  (model_utils.py:242-255,
    align_corners=True, padding_mode="border") through ATen's vectorised CPU ``grid_sampler_2d``:
    unnormalise (in+1)*((size-1)/2) (rounds like ((in+1)/2)*(size-1)), clip, x_w=floor, w=x-x_w, e=1-w, n=y-y_n,
    s=1-n, weights nw=s*e, ne=s*w, sw=n*e, se=n*w, out-of-range corners gathered as 0, and
    ``nw_val*nw + ne_val*ne + sw_val*sw + se_val*se`` accumulated left to right with FMA contraction
    (pinned bit-exact by tests/golden/corrblock.npz)."""
    inp = np.asarray(inp, dtype=f32)
    coords = np.asarray(coords, dtype=f32)
    B, C, H, W = inp.shape
    xi, wx0, wx1 = sampler_indices_weights(coords[..., 0], W)
    yi, wy0, wy1 = sampler_indices_weights(coords[..., 1], H)
    out = np.zeros((B, C) + coords.shape[1:-1], dtype=f32)
    for b in range(B):
        acc = None
        for dy, wy in ((0, wy0), (1, wy1)):
            for dx, wx in ((0, wx0), (1, wx1)):
                X, Y = xi[b] + dx, yi[b] + dy
                ok = (X <= W - 1) & (Y <= H - 1)
                v = inp[b][:, np.minimum(Y, H - 1), np.minimum(X, W - 1)]  # [C,Ho,Wo]
                v = np.where(ok[None], v, f32(0))
                w = (wy[b] * wx[b]).astype(f32)[None]
                acc = (v * w).astype(f32) if acc is None else _fma32(v, w, acc)
        out[b] = acc
    return out


def corrblock_pyramid(fmaps, num_levels=4):
    """CorrBlock.__init__ (blocks.py:300-307): fmaps [B,S,C,H,W] + (num_levels-1) x avg_pool2d(2, stride=2)."""
    return build_pyramid(fmaps, num_levels)


def corrblock_corr(fmaps_pyramid, targets):
    """CorrBlock.corr (blocks.py:342-362): targets [B,S,N,C] -> per level [B,S,N,H_l,W_l] = matmul / sqrt(C)."""
    targets = np.asarray(targets, dtype=f32)
    B, S, N, C = targets.shape
    out = []
    for fm in fmaps_pyramid:
        fm = np.asarray(fm, dtype=f32)
        H, W = fm.shape[-2:]
        corrs = np.matmul(targets, fm.reshape(B, S, C, H * W)).astype(f32)
        out.append((corrs / np.sqrt(f32(C))).astype(f32).reshape(B, S, N, H, W))
    return out


def corrblock_sample(corrs_pyramid, coords, r=3):
    """CorrBlock.sample (blocks.py:309-340): coords [B,S,N,2] -> [B*N, S, levels*(2r+1)^2].
    delta = stack(meshgrid(dy, dx, "ij")) is ADDED to (x, y): the first lattice index moves x, the second y."""
    coords = np.asarray(coords, dtype=f32)
    B, S, N, _ = coords.shape
    d = np.linspace(-r, r, 2 * r + 1, dtype=f32)
    g0, g1 = np.meshgrid(d, d, indexing="ij")
    delta = np.stack([g0, g1], axis=-1).astype(f32)[None]  # [1,7,7,2]
    outs = []
    for i, corrs in enumerate(corrs_pyramid):
        H, W = corrs.shape[-2:]
        centroid = (coords.reshape(B * S * N, 1, 1, 2) / f32(2 ** i)).astype(f32)
        lvl = (centroid + delta).astype(f32)
        smp = bilinear_sampler_4d(np.asarray(corrs, dtype=f32).reshape(B * S * N, 1, H, W), lvl)
        outs.append(smp.reshape(B, S, N, -1))
    out = np.concatenate(outs, axis=-1)
    return np.ascontiguousarray(np.transpose(out, (0, 2, 1, 3))).reshape(B * N, S, -1).astype(f32)


# --------------------------------------------------------------------------
# a-8d  Mlp / GELU / LayerNorm / Linear  blocks.py:40-76, 411-418
# --------------------------------------------------------------------------
def linear(x, w, b=None):
    y = np.asarray(x, dtype=f32) @ np.asarray(w, dtype=f32).T
    if b is not None:
        y = y + np.asarray(b, dtype=f32)
    return y.astype(f32)


def gelu_erf(x):
    x = np.asarray(x, dtype=f32)
    return (x * f32(0.5) * (f32(1) + _erf(x * f32(0.7071067811865476)).astype(f32))).astype(f32)


def gelu_tanh(x):
    x = np.asarray(x, dtype=f32)
    k = f32(0.7978845608028654)
    inner = k * (x + f32(0.044715) * x * x * x)
    return (f32(0.5) * x * (f32(1) + np.tanh(inner))).astype(f32)


def layer_norm(x, weight=None, bias=None, eps=1e-6):
    x = np.asarray(x, dtype=f32)
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    y = ((x - mu) / np.sqrt(var + eps)).astype(f32)
    if weight is not None:
        y = (y * np.asarray(weight, dtype=f32) + np.asarray(bias, dtype=f32)).astype(f32)
    return y


def mlp(x, p, prefix, act):
    """blocks.py:70-76: fc2(act(fc1(x))); dropout p=0."""
    h = act(linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"]))
    return linear(h, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])


# --------------------------------------------------------------------------
# a-8a  Attention                        blocks.py:379-398
# --------------------------------------------------------------------------
def attention(x, context, p, prefix, heads=8, attn_bias=None):
    """x [B,N1,C], context [B,N2,C]; k = first half of to_kv, v = second half.
    attn_bias (broadcastable to [B,heads,N1,N2]) is added to the scaled logits (blocks.py:392-394)."""
    B, N1, C = x.shape
    N2 = context.shape[1]
    hd = C // heads
    q = linear(x, p[prefix + "to_q.weight"], p[prefix + "to_q.bias"])
    kv = linear(context, p[prefix + "to_kv.weight"], p[prefix + "to_kv.bias"])
    k, v = kv[..., :C], kv[..., C:]
    q = q.reshape(B, N1, heads, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, N2, heads, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, N2, heads, hd).transpose(0, 2, 1, 3)
    sim = (q @ k.transpose(0, 1, 3, 2)).astype(f32) * f32(48 ** -0.5)  # blocks.py:372
    if attn_bias is not None:
        sim = (sim + np.asarray(attn_bias, dtype=f32)).astype(f32)
    sim = sim - sim.max(axis=-1, keepdims=True)
    e = np.exp(sim).astype(f32)
    attn = (e / e.sum(axis=-1, keepdims=True)).astype(f32)
    o = (attn @ v).astype(f32).transpose(0, 2, 1, 3).reshape(B, N1, C)
    return linear(o, p[prefix + "to_out.weight"], p[prefix + "to_out.bias"])


def attn_block(x, p, prefix):
    """AttnBlock.forward, blocks.py:426-438 (mask=None)."""
    xn = layer_norm(x, eps=1e-6)
    x = (x + attention(xn, xn, p, prefix + "attn.")).astype(f32)
    x = (x + mlp(layer_norm(x, eps=1e-6), p, prefix + "mlp.", gelu_tanh)).astype(f32)
    return x


def cross_attn_block(x, context, p, prefix, mask=None):
    """CrossAttnBlock.forward, cotracker.py:559-577.  mask [B, n] bool (CoTracker2 only): when n equals the number
    of queries it masks QUERIES (every logit of a masked query gets -FLT_MAX, i.e. that query attends uniformly),
    otherwise KEYS (:560-572)."""
    ctx = layer_norm(context, p[prefix + "norm_context.weight"],
                     p[prefix + "norm_context.bias"], eps=1e-5)  # cotracker.py:540
    bias = None
    if mask is not None:
        mask = np.asarray(mask, dtype=bool)
        neg = -np.finfo(np.float32).max
        if mask.shape[1] == x.shape[1]:
            bias = ((~mask)[:, None, :, None] * f32(neg)).astype(f32)
        else:
            bias = ((~mask)[:, None, None, :] * f32(neg)).astype(f32)
    x = (x + attention(layer_norm(x, eps=1e-6), ctx, p, prefix + "cross_attn.", attn_bias=bias)).astype(f32)
    x = (x + mlp(layer_norm(x, eps=1e-6), p, prefix + "mlp.", gelu_tanh)).astype(f32)
    return x


# --------------------------------------------------------------------------
# a-8  EfficientUpdateFormer.forward     cotracker.py:483-531
# --------------------------------------------------------------------------
def update_former(x, p, prefix="updateformer.", num_virtual=64, depth=3, mask=None, add_space_attn=True):
    """x [B,N,T,input_dim] -> delta [B,N,T,out].  CoTracker3: depth 3, flow_head(2) ++ vis_conf_head(2), no mask.
    CoTracker2 (cotracker.py:46-56): depth 6, one flow_head of 130 outputs, mask [B*T, N] (attention_mask)."""
    x = np.asarray(x, dtype=f32)
    tokens = linear(x, p[prefix + "input_transform.weight"], p[prefix + "input_transform.bias"])
    B, _, T, C = tokens.shape
    virt = np.broadcast_to(np.asarray(p[prefix + "virual_tracks"], dtype=f32), (B, num_virtual, T, C))
    tokens = np.concatenate([tokens, virt], axis=1)
    N = tokens.shape[1]
    for i in range(depth):
        tt = attn_block(tokens.reshape(B * N, T, C), p, f"{prefix}time_blocks.{i}.")
        tokens = tt.reshape(B, N, T, C)
        if not add_space_attn:  # cotracker.py:496-502: only the time blocks
            continue
        st = np.ascontiguousarray(tokens.transpose(0, 2, 1, 3)).reshape(B * T, N, C)
        pt, vt = st[:, : N - num_virtual], st[:, N - num_virtual:]
        vt = cross_attn_block(vt, pt, p, f"{prefix}space_virtual2point_blocks.{i}.", mask)
        vt = attn_block(vt, p, f"{prefix}space_virtual_blocks.{i}.")
        pt = cross_attn_block(pt, vt, p, f"{prefix}space_point2virtual_blocks.{i}.", mask)
        st = np.concatenate([pt, vt], axis=1)
        tokens = st.reshape(B, T, N, C).transpose(0, 2, 1, 3)
    tokens = tokens[:, : N - num_virtual]
    flow = linear(tokens, p[prefix + "flow_head.weight"], p[prefix + "flow_head.bias"])
    if prefix + "vis_conf_head.weight" not in p:  # linear_layer_for_vis_conf=False (CoTracker2)
        return flow.astype(f32)
    vc = linear(tokens, p[prefix + "vis_conf_head.weight"], p[prefix + "vis_conf_head.bias"])
    return np.concatenate([flow, vc], axis=-1).astype(f32)


# --------------------------------------------------------------------------
# a-7  posenc / time embedding           cotracker3_online.py:19-39,145-156
# --------------------------------------------------------------------------
def posenc(x, min_deg=0, max_deg=10):
    x = np.asarray(x, dtype=f32)
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype=f32)
    xb = (x[..., None, :] * scales[:, None]).astype(f32).reshape(x.shape[:-1] + (-1,))
    four = np.sin(np.concatenate([xb, (xb + f32(0.5 * math.pi)).astype(f32)], axis=-1)).astype(f32)
    return np.concatenate([x, four], axis=-1)


def sincos_time_embed(dim, window_len):
    """embeddings.py:59-84 on linspace(0,W-1,W): [1,W,dim] float32 (float64 internally)."""
    omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
    omega = 1.0 / 10000 ** omega
    pos = np.linspace(0, window_len - 1, window_len).astype(f32).astype(np.float64)
    out = np.einsum("m,d->md", pos, omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)[None].astype(f32)


def interpolate_time_embed(time_emb, t):
    """cotracker3_online.py:145-156: F.interpolate(mode='linear', align_corners=False)."""
    time_emb = np.asarray(time_emb, dtype=f32)
    T = time_emb.shape[1]
    if t == T:
        return time_emb
    scale = f32(T) / f32(t)
    dst = np.arange(t, dtype=f32)
    src = np.maximum(scale * (dst + f32(0.5)) - f32(0.5), f32(0)).astype(f32)
    i0 = np.floor(src).astype(np.int64)
    i1 = np.minimum(i0 + 1, T - 1)
    l1 = (src - i0.astype(f32)).astype(f32)
    l0 = (f32(1) - l1).astype(f32)
    return (l0[None, :, None] * time_emb[:, i0] + l1[None, :, None] * time_emb[:, i1]).astype(f32)


def assemble_tokens(coords, vis, conf, corr_embs, time_emb, model_resolution=(384, 512), stride=4):
    """cotracker3_online.py:212-248: coords [B,S,N,2], vis/conf [B,S,N,1], corr [B,S,N,1024]
    -> x [B,N,S,1110]."""
    coords = np.asarray(coords, dtype=f32)
    B, S, N, _ = coords.shape
    zero = np.zeros((B, 1, N, 2), dtype=f32)
    fwd = np.concatenate([coords[:, :-1] - coords[:, 1:], zero], axis=1)
    bwd = np.concatenate([zero, coords[:, 1:] - coords[:, :-1]], axis=1)
    scale = (np.array([model_resolution[1], model_resolution[0]], dtype=f32) / f32(stride)).astype(f32)
    pe = posenc(np.concatenate([(fwd / scale).astype(f32), (bwd / scale).astype(f32)], axis=-1))
    x = np.concatenate([vis, conf, corr_embs, pe], axis=-1).astype(f32)
    x = x.transpose(0, 2, 1, 3).reshape(B * N, S, -1)
    x = (x + interpolate_time_embed(time_emb, S)).astype(f32)
    return x.reshape(B, N, S, -1)


# --------------------------------------------------------------------------
# a-6 + a-3 + a-5: per-iteration correlation embedding
#                                        cotracker3_online.py:190-210
# --------------------------------------------------------------------------
def corr_embed(fmaps_pyramid, coords, support_pyramid, p, r=3):
    """-> corr_embs [B,S,N,1024]; coords in level-0 units."""
    B, S, N, _ = coords.shape
    embs = []
    for i, fm in enumerate(fmaps_pyramid):
        cf = get_correlation_feat(fm, (coords.reshape(B * S, N, 2) / f32(2 ** i)).astype(f32), r)
        vol = corr_volume(cf, support_pyramid[i])
        embs.append(mlp(vol.reshape(B * S * N, -1), p, "corr_mlp.", gelu_erf))
    return np.concatenate(embs, axis=-1).reshape(B, S, N, -1)


# --------------------------------------------------------------------------
# a-10 forward_window                    cotracker3_online.py:171-264
# --------------------------------------------------------------------------
def forward_window(fmaps_pyramid, coords, support_pyramid, vis, conf, p, iters=4,
                   model_resolution=(384, 512), stride=4, trace=None, add_space_attn=True):
    """Returns final (coords [B,S,N,2] level-0 units, vis logits [B,S,N,1], conf logits)."""
    coords = np.asarray(coords, dtype=f32)
    vis = np.asarray(vis, dtype=f32)
    conf = np.asarray(conf, dtype=f32)
    for it in range(iters):
        ce = corr_embed(fmaps_pyramid, coords, support_pyramid, p)
        x = assemble_tokens(coords, vis, conf, ce, p["time_emb"], model_resolution, stride)
        delta = update_former(x, p, add_space_attn=add_space_attn)
        d = delta.transpose(0, 2, 1, 3)
        coords = (coords + d[..., :2]).astype(f32)
        vis = (vis + d[..., 2:3]).astype(f32)
        conf = (conf + d[..., 3:4]).astype(f32)
        if trace is not None:
            trace.append(dict(corr_embs=ce, x=x, delta=delta, coords=coords.copy(),
                              vis=vis.copy(), conf=conf.copy()))
    return coords, vis, conf


# --------------------------------------------------------------------------
# pyramid (setup, once per forward)      cotracker3_online.py:384-409
# --------------------------------------------------------------------------
def normalize_fmaps(fmaps):
    """fmaps [B,T,C,H,W] -> channel L2 normalised (cotracker3_online.py:384-394)."""
    fmaps = np.asarray(fmaps, dtype=f32)
    n = np.sqrt(np.maximum((fmaps * fmaps).sum(axis=2, keepdims=True, dtype=f32), f32(1e-12)))
    return (fmaps / n).astype(f32)


def avg_pool2(x):
    """F.avg_pool2d(x,2,stride=2) on [...,H,W] (floor on odd sizes)."""
    H, W = x.shape[-2] // 2 * 2, x.shape[-1] // 2 * 2
    x = x[..., :H, :W]
    s = ((x[..., 0::2, 0::2] + x[..., 0::2, 1::2]).astype(f32) + x[..., 1::2, 0::2]).astype(f32)
    s = (s + x[..., 1::2, 1::2]).astype(f32)
    return (s / f32(4)).astype(f32)


def build_pyramid(fmaps, levels=4):
    pyr = [np.asarray(fmaps, dtype=f32)]
    for _ in range(levels - 1):
        pyr.append(avg_pool2(pyr[-1]))
    return pyr


def sigmoid(x):
    x = np.asarray(x, dtype=f32)
    return (f32(1) / (f32(1) + np.exp(-x))).astype(f32)


# --------------------------------------------------------------------------
# L3  CoTrackerThreeOnline.forward (post-encoder part)
#                                        cotracker3_online.py:266-541
# --------------------------------------------------------------------------
class OnlineState:
    """cotracker3_online.py:163-169"""

    def __init__(self, levels=4):
        self.online_ind = 0
        self.track_support = [None] * levels
        self.coords = None
        self.vis = None
        self.conf = None


def model_forward_online(fmaps, queries, p, window_len=16, iters=4, stride=4,
                         model_resolution=(384, 512), is_online=False, state=None, T=None):
    """Everything after fnet + L2-normalise.

    fmaps [B,T_pad,C,H4,W4] *already padded* to the window multiple (the
    reference pads the video by repeating the last frame, :321-328, which is
    the same as repeating the last feature map because fnet is per-frame).
    queries [B,N,3]=(t,x,y) pixels at model resolution.  T = unpadded length.
    Returns coords [B,T,N,2] px, vis [B,T,N], conf [B,T,N] (post-sigmoid).
    """
    fmaps = np.asarray(fmaps, dtype=f32)
    queries = np.asarray(queries, dtype=f32)
    B, T_pad = fmaps.shape[:2]
    T = T_pad if T is None else T
    N = queries.shape[1]
    S = window_len
    step = S // 2
    qf = queries[:, :, 0].astype(np.int64)
    qc = (queries[..., 1:3] / f32(stride)).astype(f32)
    pyr = build_pyramid(fmaps)
    L = len(pyr)

    coords_pred = np.zeros((B, T, N, 2), dtype=f32)
    vis_pred = np.zeros((B, T, N), dtype=f32)
    conf_pred = np.zeros((B, T, N), dtype=f32)
    if is_online:
        if state.coords is not None:  # :349-360
            pad = min(step, T - step)
            coords_pred = np.concatenate([state.coords, np.zeros((B, pad, N, 2), f32)], axis=1)
            vis_pred = np.concatenate([state.vis, np.zeros((B, pad, N), f32)], axis=1)
            conf_pred = np.concatenate([state.conf, np.zeros((B, pad, N), f32)], axis=1)
        left = 0 if state.online_ind == 0 else state.online_ind + step
        right = state.online_ind + S
        sample_mask = ((qf >= left) & (qf < right))[:, None, :, None]  # B 1 N 1

    support = []
    for i in range(L):
        frames = qf - state.online_ind if is_online else qf
        sup = get_track_feat(pyr[i], frames, (qc / f32(2 ** i)).astype(f32), 3)
        if is_online:  # :424-440
            if state.track_support[i] is None:
                state.track_support[i] = np.zeros_like(sup)
            state.track_support[i] = (state.track_support[i] + sup * sample_mask).astype(f32)
            sup = state.track_support[i]
        support.append(sup)

    vis_init = np.zeros((B, S, N, 1), dtype=f32)
    conf_init = np.zeros((B, S, N, 1), dtype=f32)
    coords_init = np.broadcast_to(qc.reshape(B, 1, N, 2), (B, S, N, 2)).astype(f32)

    num_windows = (T - S + step - 1) // step + 1
    indices = [state.online_ind] if is_online else list(range(0, step * num_windows, step))
    for ind in indices:
        if ind > 0:  # :457-482
            overlap = S - step
            copy_over = (qf < ind + overlap)[:, None, :, None]
            cprev = coords_pred[:, ind: ind + overlap] / f32(stride)
            cprev = np.concatenate([cprev, np.repeat(cprev[:, -1:], step, axis=1)], axis=1)
            vprev = vis_pred[:, ind: ind + overlap, :, None]
            vprev = np.concatenate([vprev, np.repeat(vprev[:, -1:], step, axis=1)], axis=1)
            fprev = conf_pred[:, ind: ind + overlap, :, None]
            fprev = np.concatenate([fprev, np.repeat(fprev[:, -1:], step, axis=1)], axis=1)
            coords_init = np.where(copy_over, cprev, coords_init).astype(f32)
            vis_init = np.where(copy_over, vprev, vis_init).astype(f32)
            conf_init = np.where(copy_over, fprev, conf_init).astype(f32)
        amask = (qf < ind + S)  # [B,N]  :484
        sup_masked = [(amask[:, None, :, None] * s).astype(f32) for s in support]  # :493-496
        win_pyr = pyr if is_online else [f[:, ind: ind + S] for f in pyr]
        c, v, f = forward_window(win_pyr, coords_init, sup_masked, vis_init, conf_init, p,
                                 iters=iters, model_resolution=model_resolution, stride=stride)
        S_trim = T if is_online else min(T - ind, S)
        coords_pred[:, ind: ind + S] = (c * f32(stride))[:, :S_trim]
        vis_pred[:, ind: ind + S] = v[:, :S_trim, :, 0]
        conf_pred[:, ind: ind + S] = f[:, :S_trim, :, 0]
    if is_online:
        state.online_ind += step
        state.coords, state.vis, state.conf = coords_pred, vis_pred, conf_pred
    return coords_pred, sigmoid(vis_pred), sigmoid(conf_pred)


# --------------------------------------------------------------------------
# L3  CoTrackerThreeOffline.forward (post-encoder part)
#                                        cotracker3_offline.py:104-233
# --------------------------------------------------------------------------
def model_forward_offline(fmaps, queries, p, iters=4, stride=4, model_resolution=(384, 512), add_space_attn=True):
    fmaps = np.asarray(fmaps, dtype=f32)
    queries = np.asarray(queries, dtype=f32)
    B, T = fmaps.shape[:2]
    N = queries.shape[1]
    qf = queries[:, :, 0].astype(np.int64)
    qc = (queries[..., 1:3] / f32(stride)).astype(f32)
    pyr = build_pyramid(fmaps)
    support = [get_track_feat(pyr[i], qf, (qc / f32(2 ** i)).astype(f32), 3) for i in range(len(pyr))]
    coords = np.broadcast_to(qc.reshape(B, 1, N, 2), (B, T, N, 2)).astype(f32)
    vis = np.zeros((B, T, N, 1), dtype=f32)
    conf = np.zeros((B, T, N, 1), dtype=f32)
    c, v, f = forward_window(pyr, coords, support, vis, conf, p, iters=iters,
                             model_resolution=model_resolution, stride=stride, add_space_attn=add_space_attn)  # cotracker.py:496-502
    return (c * f32(stride)).astype(f32), sigmoid(v[..., 0]), sigmoid(f[..., 0])


# ==========================================================================
# CoTracker2 (cotracker.py:29-384) -- SURVEY section 8(f) rank 3: the model around CorrBlock
# ==========================================================================
def get_2d_embedding(xy, C=64, cat_coords=True):
    """embeddings.py:87-120: [.., 2] -> [.., 2 (coords) + C (x: sin/cos interleaved) + C (y)]."""
    xy = np.asarray(xy, dtype=f32)
    div = (np.arange(0, C, 2, dtype=f32) * f32(1000.0 / C)).astype(f32)
    out = []
    for a in (xy[..., 0:1], xy[..., 1:2]):
        arg = (a * div).astype(f32)
        pe = np.zeros(xy.shape[:-1] + (C,), dtype=f32)
        pe[..., 0::2] = np.sin(arg)
        pe[..., 1::2] = np.cos(arg)
        out.append(pe)
    pe = np.concatenate(out, axis=-1)
    return np.concatenate([xy, pe], axis=-1).astype(f32) if cat_coords else pe


def sample_features4d(inp, coords):
    """model_utils.py:258-290: input [B,C,H,W], coords [B,R,2]=(x,y) -> [B,R,C] (4-D grid_sample path)."""
    out = bilinear_sampler_4d(inp, np.asarray(coords, dtype=f32)[:, :, None, :])  # [B,C,R,1]
    return np.ascontiguousarray(out[..., 0].transpose(0, 2, 1))


def group_norm1(x, weight, bias, eps=1e-5):
    """nn.GroupNorm(1, C) on a 2-D [rows, C] input (cotracker.py:79, used at :167): one group = the whole row."""
    return layer_norm(x, weight, bias, eps=eps)


def forward_window_v2(fmaps, coords, track_feat, vis, track_mask, attention_mask, p, iters=4, stride=4, trace=None):
    """CoTracker2.forward_window (cotracker.py:86-173).  fmaps [B,S,C,H,W] (NOT normalised), coords [B,S,N,2]
    feature units, track_feat [B,S,N,C], vis [B,S,N,1], track_mask [B,S_init,N,1], attention_mask [B,S,N] bool.
    Returns (coords of the last iteration in feature units [B,S,N,2], vis logits [B,S,N])."""
    fmaps = np.asarray(fmaps, dtype=f32)
    coords = np.asarray(coords, dtype=f32)
    track_feat = np.asarray(track_feat, dtype=f32)
    B, S = fmaps.shape[:2]
    N = coords.shape[2]
    C = track_feat.shape[-1]
    tm = np.asarray(track_mask, dtype=f32)
    if tm.shape[1] < S:
        tm = np.concatenate([tm, np.zeros((B, S - tm.shape[1], N, 1), f32)], axis=1)
    tmv = np.concatenate([tm, np.asarray(vis, dtype=f32)], axis=-1).transpose(0, 2, 1, 3).reshape(B * N, S, 2)
    pyr = corrblock_pyramid(fmaps)
    pos = sample_features4d(np.repeat(np.asarray(p["pos_emb"], dtype=f32), B, axis=0), coords[:, 0])  # [B,N,E]
    pos = pos.reshape(B * N, 1, -1)
    amask = np.asarray(attention_mask, dtype=bool).reshape(B * S, N)
    for _ in range(iters):
        fcorrs = corrblock_sample(corrblock_corr(pyr, track_feat), coords)  # [(B N), S, 196]
        flows = (coords - coords[:, 0:1]).astype(f32).transpose(0, 2, 1, 3).reshape(B * N, S, 2)
        flow_emb = get_2d_embedding(flows, 64, cat_coords=True)
        tf_ = track_feat.transpose(0, 2, 1, 3).reshape(B * N, S, C)
        x = np.concatenate([flow_emb, fcorrs, tf_, tmv], axis=2).astype(f32)
        x = ((x + pos).astype(f32) + np.asarray(p["time_emb"], dtype=f32)).astype(f32)
        delta = update_former(x.reshape(B, N, S, -1), p, depth=6, mask=amask)  # [B,N,S,130]
        coords = (coords + delta[..., :2].transpose(0, 2, 1, 3)).astype(f32)
        dfe = delta[..., 2:].reshape(B * N * S, C)
        upd = gelu_erf(linear(group_norm1(dfe, p["norm.weight"], p["norm.bias"]),
                              p["track_feat_updater.0.weight"], p["track_feat_updater.0.bias"]))
        tfn = (upd + track_feat.transpose(0, 2, 1, 3).reshape(B * N * S, C)).astype(f32)
        track_feat = np.ascontiguousarray(tfn.reshape(B, N, S, C).transpose(0, 2, 1, 3))
        if trace is not None:
            trace.append(dict(x=x, delta=delta, coords=coords.copy(), track_feat=track_feat.copy()))
    vis_pred = linear(track_feat, p["vis_predictor.0.weight"], p["vis_predictor.0.bias"]).reshape(B, S, N)
    return coords, vis_pred


class OnlineStateV2:
    """cotracker.py:187-191"""

    def __init__(self):
        self.online_ind = 0
        self.track_feat = None
        self.coords = None
        self.vis = None


def model_forward_v2(fmaps, queries, p, window_len=8, iters=4, stride=4, is_online=False, state=None, T=None):
    """CoTracker2.forward after fnet (cotracker.py:193-384).  fmaps [B,T_pad,C,H4,W4] already padded to the window
    multiple (the reference pads the video with its last frame; fnet is per-frame).  queries [B,N,3]=(t,x,y) px.
    Returns coords [B,T,N,2] px, vis [B,T,N] (post-sigmoid)."""
    fmaps = np.asarray(fmaps, dtype=f32)
    queries = np.asarray(queries, dtype=f32)
    B, T_pad = fmaps.shape[:2]
    T = T_pad if T is None else T
    N = queries.shape[1]
    S = window_len
    step = S // 2
    qf = queries[:, :, 0].astype(np.int64)
    qc = (queries[..., 1:3] / f32(stride)).astype(f32)
    coords_pred = np.zeros((B, T, N, 2), dtype=f32)
    vis_pred = np.zeros((B, T, N), dtype=f32)
    if is_online and state.coords is not None:
        pad = min(step, T - step)
        coords_pred = np.concatenate([state.coords, np.zeros((B, pad, N, 2), f32)], axis=1)
        vis_pred = np.concatenate([state.vis, np.zeros((B, pad, N), f32)], axis=1)
    frames = qf - state.online_ind if is_online else qf
    # get_track_feat (cotracker.py:175-185): trilinear sample at (t, x, y), one vector per track, repeated over S
    sc = np.concatenate([frames[:, None, :, None].astype(f32), qc[:, None]], axis=-1)  # [B,1,N,3]
    tf0 = sample_features5d(fmaps, sc)  # [B,1,N,C]
    track_feat = np.repeat(tf0, S, axis=1).astype(f32)
    if is_online:
        left = 0 if state.online_ind == 0 else state.online_ind + step
        right = state.online_ind + S
        smask = ((qf >= left) & (qf < right))[:, None, :, None]
        if state.track_feat is None:
            state.track_feat = np.zeros_like(track_feat)
        state.track_feat = (state.track_feat + track_feat * smask).astype(f32)
        track_feat = state.track_feat.copy()
    num_windows = (T - S + step - 1) // step + 1
    indices = [state.online_ind] if is_online else list(range(0, step * num_windows, step))
    coords_init = np.broadcast_to(qc.reshape(B, 1, N, 2), (B, S, N, 2)).astype(f32)
    vis_init = np.full((B, S, N, 1), 10.0, dtype=f32)
    for ind in indices:
        overlap = S - step
        if ind > 0:
            copy_over = (qf < ind + overlap)[:, None, :, None]
            cprev = coords_pred[:, ind: ind + overlap] / f32(stride)
            cprev = np.concatenate([cprev, np.repeat(cprev[:, -1:], step, axis=1)], axis=1)
            vprev = vis_pred[:, ind: ind + overlap, :, None]
            vprev = np.concatenate([vprev, np.repeat(vprev[:, -1:], step, axis=1)], axis=1)
            coords_init = np.where(copy_over, cprev, coords_init).astype(f32)
            vis_init = np.where(copy_over, vprev, vis_init).astype(f32)
        amask = np.repeat((qf < ind + S)[:, None, :], S, axis=1)  # [B,S,N]
        tmask = (qf[:, None, :, None] <= np.arange(ind, ind + S)[None, :, None, None])  # [B,S,N,1]
        if ind > 0:
            tmask = tmask.copy()
            tmask[:, :overlap] = False
        win = fmaps if is_online else fmaps[:, ind: ind + S]
        c, v = forward_window_v2(win, coords_init, (amask[..., None] * track_feat).astype(f32), vis_init, tmask, amask,
                                 p, iters=iters, stride=stride)
        S_trim = T if is_online else min(T - ind, S)
        coords_pred[:, ind: ind + S] = (c * f32(stride))[:, :S_trim]
        vis_pred[:, ind: ind + S] = v[:, :S_trim]
    if is_online:
        state.online_ind += step
        state.coords, state.vis = coords_pred, vis_pred
    return coords_pred, sigmoid(vis_pred)
