"""f16 range of the split-half back end (include/ctk.h, "Numeric range of the split-half format") -- -m gpu.

The SH format stores x = hi + lo in two IEEE halves, so it carries ~21 significant bits only while
2^-3 <~ |x| < 65504; below, `lo` (then `hi`) falls into the f16 subnormals and the error becomes ABSOLUTE
(<= 2^-25 ~ 3e-8 per element); above, `hi` overflows.  These tests (a) pin that accuracy model on activations and
weights scaled by 2^-15 ... 2^13, through GEMM, LayerNorm->GEMM and attention, (b) run the whole update on weights
with trained-like wide per-layer norms, and (c) show that an activation beyond the range gives a DEFINED outcome:
a RuntimeWarning and the result of the exact-f32 MFMA back end, never NaN tracks.
"""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cotracker_oracle as O  # noqa: E402  (checker only)


def dev():
    return torch.device("cuda:0")


def maxdiff(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


@pytest.mark.parametrize("ascale", [2.0 ** -15, 2.0 ** -10, 1.0, 2.0 ** 10, 2.0 ** 13])
@pytest.mark.parametrize("wscale", [2.0 ** -12, 1.0, 2.0 ** 12])
def test_gemm_activation_and_weight_scales(ascale, wscale):
    """|A| from 3e-5 to 3.7e4 (randn, max |z| ~ 4.5), weights from 2e-4 to 4e3: relative 2^-21-class error while the
    activations are normal halves, plus the documented absolute floor of 2^-25 per activation element below."""
    from cotracker_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    M, K, N = 777, 384, 384
    a = (torch.randn(M, K, generator=g) * ascale).to(dev())
    w = (torch.randn(N, K, generator=g) * wscale / K ** 0.5).to(dev())
    assert float(a.abs().max()) < 65504
    ref = a.double() @ w.double().t()
    out = ops.gemm(ops.split_rows(a), w, packed=ops.pack_weight(w))
    rel = 3e-6 * ascale * wscale                        # ~2^-21 per product, sqrt(K) accumulation, |ref| ~ ascale*wscale
    floor = 2.0 ** -25 * float(w.abs().max()) * K ** 0.5 * 4  # subnormal halves: absolute 2^-25 per A element
    assert maxdiff(out, ref) < rel + floor
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("xscale", [2.0 ** -15, 2.0 ** -10, 2.0 ** 10, 2.0 ** 15])
def test_layernorm_then_gemm_is_scale_free(xscale):
    """The residual stream is f32 and only enters a GEMM through LayerNorm, whose output is O(1) whatever the stream's
    scale: LN -> GEMM stays at full accuracy for tokens of magnitude 3e-5 ... 1.5e5."""
    from cotracker_amd import ops
    g = torch.Generator().manual_seed(2)
    x = ((torch.randn(515, 384, generator=g) * 3 + 0.5) * xscale).to(dev())
    w = (torch.randn(384, 384, generator=g) / 20).to(dev())
    eps = 1e-6
    xd = x.double()
    xn = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + eps)
    ref = xn @ w.double().t()
    out = ops.gemm(ops.layernorm(x, eps=eps, out_split=True), w, packed=ops.pack_weight(w))
    assert maxdiff(out, ref) < 2e-5 * max(1.0, float(ref.abs().max()) / 4)


@pytest.mark.parametrize("scale", [2.0 ** -10, 2.0 ** -5, 1.0, 2.0 ** 3])
def test_attention_qkv_scales(scale):
    """q / k / v magnitudes from 1e-3 to 8 (scores up to ~ +-3000 before the softmax scale at 2^3)."""
    from cotracker_amd import ops
    g = torch.Generator().manual_seed(9)
    B, N1, N2 = 16, 64, 600
    q = (torch.randn(B, N1, 384, generator=g) * scale).to(dev())
    k = (torch.randn(B, N2, 384, generator=g) * scale).to(dev())
    v = (torch.randn(B, N2, 384, generator=g) * scale).to(dev())
    out = ops.attention(q, k, v, splits=2)
    qd, kd, vd = (x.double().reshape(B, -1, 8, 48).transpose(1, 2) for x in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * 48 ** -0.5, dim=-1) @ vd).transpose(1, 2).reshape(B, N1, 384)
    # the score error 2^-21 * |q.k| enters the output through the softmax: d(out) ~ |v| * d(score)
    score = float((qd @ kd.transpose(-1, -2)).abs().max()) * 48 ** -0.5
    assert maxdiff(out, ref) < scale * (3e-6 + 2e-6 * score) + 2.0 ** -23  # + the absolute floor of subnormal halves


def _window_case(widen, seed=3):
    """forward_window inputs + oracle outputs on weights whose per-layer norms are spread over [1/widen, widen]."""
    from cotracker_amd.model import CoTrackerThreeOnline
    from cotracker_amd.weights import fill_synthetic_
    m = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=8, model_resolution=(192, 256)).eval()
    fill_synthetic_(m, seed=seed, head_scale=4.0)
    r = np.random.RandomState(17)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith("fnet.") or ".flow_head." in name or ".vis_conf_head." in name or name.endswith("virual_tracks"):
                continue
            if name.endswith(".weight") and p.dim() == 2:  # log-uniform per-layer gain: trained nets are not xavier-flat
                p.mul_(float(np.exp(r.uniform(-np.log(widen), np.log(widen)))))
            elif name.endswith(".bias"):
                p.mul_(float(np.exp(r.uniform(0, np.log(widen) * 2))))
    m.invalidate_packed_weights()
    return m


_oracle_cache = {}


def _trained_like_oracle():
    """Inputs + numpy-oracle outputs, computed once for both precisions (the oracle takes ~10-60 s on CPU)."""
    if not _oracle_cache:
        m = _window_case(4.0)
        p = {k: v.numpy() for k, v in m.state_dict().items() if not k.startswith("fnet.")}
        r = np.random.RandomState(1)
        S, N = 8, 40
        f = r.standard_normal((1, S, 128, 48, 64)).astype(np.float32)
        pyr = O.build_pyramid(O.normalize_fmaps(f))
        qf = r.randint(0, S, size=(1, N))
        qc = (r.uniform(0, 1, size=(1, N, 2)) * np.array([63, 47])).astype(np.float32)
        sup = [O.get_track_feat(pyr[i], qf, (qc / np.float32(2 ** i)).astype(np.float32)) for i in range(4)]
        cinit = np.broadcast_to(qc.reshape(1, 1, N, 2), (1, S, N, 2)).astype(np.float32)
        z = np.zeros((1, S, N, 1), np.float32)
        out = O.forward_window(pyr, cinit, sup, z, z, p, iters=2, model_resolution=(192, 256))
        _oracle_cache.update(m=m, pyr=pyr, sup=sup, cinit=cinit, out=out, S=S, N=N)
    return _oracle_cache


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_forward_window_trained_like_weight_spread(precision):
    """Two update iterations with per-layer weight norms spread over 1/4x ... 4x (biases up to 16x) against the numpy
    oracle: the surrogate for a released checkpoint (none is available offline).  The tracks move ~3.5 px and the
    map's sensitivity to its input is ~4x (measured on the oracle), so 1e-3 px / 1e-4 logit is a precision statement."""
    from cotracker_amd import ops
    k = _trained_like_oracle()
    m, pyr, sup, cinit, (c, v, cf), S, N = k["m"], k["pyr"], k["sup"], k["cinit"], k["out"], k["S"], k["N"]
    m.precision = precision
    m = m.to(dev())
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())  # noqa: E731
    fm = [T(np.transpose(x[0], (0, 2, 3, 1))) for x in pyr]
    sp = [T(np.transpose(s[0], (1, 0, 2))) for s in sup]
    coords, vis, conf = T(cinit[0]), torch.zeros(S, N, device=dev()), torch.zeros(S, N, device=dev())
    ops.forward_window(ops.Window(fm, sp, coords, vis, conf, (64.0, 48.0), iters=2), m.packed(dev()))
    assert torch.isfinite(coords).all()
    assert float((coords.cpu() - torch.from_numpy(cinit[0])).abs().max()) > 1e-2  # the update does move the tracks
    assert maxdiff(coords, torch.from_numpy(c[0])) * 4 < 1e-3     # px
    assert maxdiff(vis, torch.from_numpy(v[0, ..., 0])) < 1e-4
    assert maxdiff(conf, torch.from_numpy(cf[0, ..., 0])) < 1e-4


def test_overflow_gives_defined_result_not_nan():
    """An MLP whose hidden activations exceed 65504 (fc1 weights x 3e5): the split-half run goes non-finite, the guard
    notices, warns, and returns what the exact-f32 back end computes."""
    from cotracker_amd.model import CoTrackerThreeOnline
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_

    def build(precision):
        m = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=8, model_resolution=(64, 96)).eval()
        fill_synthetic_(m, seed=1)
        with torch.no_grad():
            m.updateformer.time_blocks[0].mlp.fc1.weight.mul_(3e5)
            m.updateformer.time_blocks[0].mlp.fc2.weight.mul_(1e-5)
        m.invalidate_packed_weights()
        m.precision = precision
        return m.to(dev())

    video = synthetic_video(12, 64, 96, seed=5).to(dev())
    q = torch.tensor([[[0.0, 20.0, 20.0], [2.0, 60.0, 40.0], [0.0, 80.0, 10.0]]], device=dev())
    exact = build("f32")
    c32, v32, f32_, _ = exact(video, q, iters=3)
    assert torch.isfinite(c32).all() and exact.range_fallbacks == 0
    m = build("f16x3")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c, v, f, _ = m(video, q, iters=3)
    assert m.range_fallbacks == 1 and any(issubclass(x.category, RuntimeWarning) for x in w)
    assert torch.isfinite(c).all() and torch.isfinite(v).all()
    assert torch.equal(c, c32) and torch.equal(v, v32)   # the fallback IS the exact-f32 back end (deterministic HIP encoder)
    # with the guard off the caller sees the non-finite values (documented, opt-out only)
    m.range_guard = False
    c_raw, *_ = m(video, q, iters=3)
    assert not torch.isfinite(c_raw).all()
    # streaming through the window graph (CoTrackerOnlinePredictor's mode): the check is deferred by one call so the
    # chunk stream never waits for the GPU -- the NEXT call raises a defined error instead of handing on NaN state
    m.range_guard = True
    m.hip_graph = True
    m.init_video_online_processing()
    m(video[:, 0:8], q, iters=2, is_online=True)
    with pytest.raises(FloatingPointError, match="f16 range"):
        m(video[:, 4:12], q, iters=2, is_online=True)
    # ... and when no next call comes (the LAST chunk of a stream), CoTrackerOnlinePredictor.finish() examines it: same error
    m.init_video_online_processing()
    m(video[:, 0:8], q, iters=2, is_online=True)
    with pytest.raises(FloatingPointError, match="f16 range"):
        m._resolve_deferred_range_check()  # = what predictor.finish() calls
    m._resolve_deferred_range_check()      # nothing pending any more: a no-op
    # stream_range_check = "immediate": graph streaming that never raises -- the flag is waited for inside the call and the chunk
    # is re-run on the exact-f32 back end (online state restored first) before it is returned, like every non-streaming path
    m.stream_range_check = "immediate"
    exact.hip_graph = True
    m.init_video_online_processing()
    exact.init_video_online_processing()
    before = m.range_fallbacks
    for ind in range(0, 8, 4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cs, *_ = m(video[:, ind:ind + 8], q, iters=2, is_online=True)
        ce, *_ = exact(video[:, ind:ind + 8], q, iters=2, is_online=True)
        assert torch.equal(cs, ce) and m.online_ind == exact.online_ind
    assert m.range_fallbacks == before + 2 and m._pending_range is None
    m.stream_range_check = "deferred"
    exact.hip_graph = False
    m.hip_graph = False
    # streaming without the graph: immediate check, the online state is restored before the f32 re-run
    m.init_video_online_processing()
    exact.init_video_online_processing()
    for ind in range(0, 8, 4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cs, *_ = m(video[:, ind:ind + 8], q, iters=2, is_online=True)
        ce, *_ = exact(video[:, ind:ind + 8], q, iters=2, is_online=True)
        assert m.online_ind == exact.online_ind
        assert torch.equal(cs, ce)


def test_cotracker2_overflow_gives_defined_result_not_nan():
    """The f16 range guard of CoTracker2 (round 4; round 3 had none): an MLP whose hidden activations exceed 65504 makes the
    split-half run non-finite -> sliding windows: RuntimeWarning + the exact-f32 back end's result, bit for bit (deterministic
    HIP encoder); streaming through the window graph: the NEXT call (or init_video_online_processing / finish) raises."""
    from cotracker_amd.model_v2 import CoTracker2
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_

    def build(precision):
        m = CoTracker2(stride=4, window_len=8, model_resolution=(64, 96)).eval()
        fill_synthetic_(m, seed=6, head_scale=1.0)
        with torch.no_grad():
            m.updateformer.time_blocks[0].mlp.fc1.weight.mul_(3e5)
            m.updateformer.time_blocks[0].mlp.fc2.weight.mul_(1e-5)
        m.invalidate_packed_weights()
        m.precision = precision
        return m.to(dev())

    video = synthetic_video(12, 64, 96, seed=5).to(dev())
    q = torch.tensor([[[0.0, 20.0, 20.0], [2.0, 60.0, 40.0], [0.0, 80.0, 10.0]]], device=dev())
    exact = build("f32")
    c32, v32, _ = exact(video, q, iters=2)
    assert torch.isfinite(c32).all() and exact.range_fallbacks == 0
    m = build("f16x3")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c, v, _ = m(video, q, iters=2)
    assert m.range_fallbacks == 1 and any(issubclass(x.category, RuntimeWarning) for x in w)
    assert torch.equal(c, c32) and torch.equal(v, v32)  # the fallback IS the exact-f32 back end
    m.range_guard = False
    assert not torch.isfinite(m(video, q, iters=2)[0]).all()
    m.range_guard = True
    m.hip_graph = True
    m.init_video_online_processing()
    m(video[:, 0:8], q, iters=2, is_online=True)
    with pytest.raises(FloatingPointError, match="f16 range"):
        m(video[:, 4:12], q, iters=2, is_online=True)
    m.init_video_online_processing()
    m(video[:, 0:8], q, iters=2, is_online=True)
    with pytest.raises(FloatingPointError, match="f16 range"):
        m.init_video_online_processing()   # the last chunk of a stream is examined when the next stream starts ...
    m.hip_graph = False
    # streaming without the graph: immediate check, the online state is restored before the f32 re-run
    m.init_video_online_processing()
    exact.init_video_online_processing()
    for ind in range(0, 8, 4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cs, vs, _ = m(video[:, ind:ind + 8], q, iters=2, is_online=True)
        ce, ve, _ = exact(video[:, ind:ind + 8], q, iters=2, is_online=True)
        assert m.online_ind == exact.online_ind
        assert torch.equal(cs, ce) and torch.equal(vs, ve)


def test_range_guard_sweep_at_c2_scale():
    """with no released checkpoint available, sweep synthetic weights whose hidden activations cover
    ~1e1 ... 1e6 through the FULL predictor at BASELINE configs[1] scale (256x256, T=48, N=400, offline) and record where the
    split-half back end first leaves the f16 range.  Below that point it must agree with the exact-f32 back end to the
    parity bar; from that point on the guard must fire and the returned result must be the exact-f32 back end's, bit for bit
    (the HIP encoder is deterministic, so the re-run reproduces a direct f32 run exactly).  The table goes to
    gpurun_out/range_sweep_c2.json (copied to profiles/)."""
    import json
    import os
    from cotracker_amd import model as M
    from cotracker_amd.predictor import CoTrackerPredictor
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_

    video = synthetic_video(48, 256, 256, seed=1234).to(dev())

    def run(precision, scale):
        old, M.DEFAULT_PRECISION = M.DEFAULT_PRECISION, precision
        try:
            p = CoTrackerPredictor(checkpoint=None, offline=True, window_len=60)
        finally:
            M.DEFAULT_PRECISION = old
        fill_synthetic_(p.model, seed=0)
        with torch.no_grad():  # hidden = gelu(fc1(x)) grows with `scale`; fc2 undoes it, so the function stays comparable
            for blk in p.model.updateformer.time_blocks:
                blk.mlp.fc1.weight.mul_(scale)
                blk.mlp.fc1.bias.mul_(scale)
                blk.mlp.fc2.weight.mul_(1.0 / scale)
        p.model.invalidate_packed_weights()
        p = p.to(dev())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            tracks, vis = p(video, grid_size=20)
        vl, cl = p.model.last_logits
        return tracks.clone(), vl.clone(), cl.clone(), p.model.range_fallbacks

    rows, first = [], None
    for scale in (1.0, 1e2, 1e3, 3e3, 1e4, 3e4, 1e5, 1e6):
        t32, v32, c32, fb32 = run("f32", scale)
        t16, v16, c16, fb16 = run("f16x3", scale)
        assert fb32 == 0 and torch.isfinite(t32).all()
        row = {"fc1_scale": scale, "fallback": bool(fb16), "tracks_px": maxdiff(t16, t32), "vis_logit": maxdiff(v16, v32),
               "conf_logit": maxdiff(c16, c32), "bit_identical_to_f32": bool(torch.equal(t16, t32) and torch.equal(v16, v32))}
        rows.append(row)
        if fb16:
            first = first or scale
            assert row["bit_identical_to_f32"], row   # the fallback IS the exact-f32 back end
        else:
            assert first is None, "the guard fired at a smaller scale but not here"
            assert row["tracks_px"] < 2e-3 and row["vis_logit"] < 2e-4 and row["conf_logit"] < 2e-4, row
    assert first is not None, "the sweep never left the f16 range: extend it"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "range_sweep_c2.json"), "w") as f:
        json.dump({"workload": "BASELINE configs[1]: 256x256 T=48 N=400 offline, time_blocks[*].mlp.fc1 x scale, fc2 / scale",
                   "first_fallback_at_scale": first, "rows": rows}, f, indent=1)
    print(json.dumps(rows))
