"""The persistent ping-pong split-half GEMM kernels (co-tracker_amd/csrc/gemm_pp.hip) through the C-ABI (-m gpu).

Every Linear flavour of the update path, at sizes that take the persistent path (>= one 256-row tile per CU):
  * against fp64 (the nn.Linear contract) and against gemm_f16x3.hip's kernels (`ctk_gemm_pp_mode(0)`), which run the same
    MFMA sequence per output element: bit-identical unless a residual is added (round 5: the residual tile rides on the
    first eight K-tiles of a tile and is added into the accumulators between MFMA phases; gemm_f16x3.hip adds it last);
  * with the TAIL SPLIT (mode bit 5, the default since round 5): the row blocks of a nearly empty last round go to the
    64 x 64-tile kernel -- rows are independent, so the bits must not depend on where the cut is;
  * CONCURRENCY (round 6): two host threads launching on two streams at once -- the release library keeps no device-side or
    host-side state a launch could trample on;
  * TIMING ROBUSTNESS: in the DEV build (`make dev`, libctk_hip_dev.so) mode bit 3 makes every wave sleep pseudo-random times
    around every barrier.  The LDS-DMA ring / barrier protocol must not depend on timing, so the result has to stay
    bit-identical.  (This is the test that found the round-3 race: a wave leaving its epilogue early issued LDS-DMA into a ring
    slot another wave of its group was still using as store-transpose scratch -- invisible alone, garbage when a second process
    shared the GPU.)  The release library has no such switch: the jitter runs happen in a subprocess that loads the dev build.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH = 0, 1, 2
# name: (M, K, N, act, residual, SH output, bias rows, bias)
CASES = {
    "mlp.fc1": (256 * 47 + 100, 384, 1536, ACT_GELU_TANH, False, True, False, True),    # 256x256 tiles, ragged M
    "to_kv": (256 * 90, 384, 768, ACT_NONE, False, False, False, True),                 # 256x256
    "corr_mlp.fc2": (256 * 260, 384, 256, ACT_NONE, False, True, False, True),          # 256x256, one column block
    "to_q": (256 * 130 + 31, 384, 384, ACT_NONE, False, False, False, True),            # 256x192, ragged M
    "to_out": (256 * 131, 384, 384, ACT_NONE, True, False, False, True),                # 256x192, residual on the first 8 of 12 K-tiles
    "mlp.fc2": (256 * 129, 1536, 384, ACT_NONE, True, False, False, True),              # 256x192, long K
    "input_transform": (256 * 140, 1120, 384, ACT_NONE, False, False, True, False),     # 256x192, odd K-tile count (35), bias rows
    "corr_mlp.fc1": (256 * 128, 2432, 384, ACT_GELU_ERF, False, True, False, True),     # 256x192, erf GELU, SH output
}


def _run(case, mode, data):
    from cotracker_amd import _lib, ops
    M, K, N, act, res, split, brows, bias = CASES[case]
    a_sh, w, wp, b, br, r = data
    _lib.load().ctk_gemm_pp_mode(mode)
    try:
        if res:
            out = r.clone()
            ops.gemm(a_sh, w, bias=b, act=act, resid=out, out=out, packed=wp)  # x += Linear(.)
        else:
            out = ops.gemm(a_sh, w, bias=b, act=act, bias_rows=br, packed=wp, out_split=split)
        torch.cuda.synchronize()
    finally:
        _lib.load().ctk_gemm_pp_mode(33)  # the library's default: persistent kernels + tail split
    return out


def _make(case):
    from cotracker_amd import ops
    M, K, N, act, res, split, brows, bias = CASES[case]
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(K + N + act)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    br = torch.randn(16, N, generator=g).to(dev) if brows else None
    r = (3 * torch.randn(M, N, generator=g)).to(dev) if res else None
    data = (ops.split_rows(a), w, ops.pack_weight(w), b, br, r)
    return a, w, b, br, r, data


@pytest.mark.parametrize("case", sorted(CASES))
def test_gemm_pp_matches_fp64_and_old_kernels(case):
    from cotracker_amd import ops
    M, K, N, act, res, split, brows, bias = CASES[case]
    dev = torch.device("cuda:0")
    a, w, b, br, r, data = _make(case)
    ref = a.double() @ w.double().t()
    if bias:
        ref += b.double()
    if brows:
        ref += br.double()[torch.arange(M, device=dev) % 16]
    if act == ACT_GELU_ERF:
        ref = torch.nn.functional.gelu(ref)
    elif act == ACT_GELU_TANH:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if res:
        ref += r.double()

    def f32(o):
        return ops.unsplit(o) if split else o

    old = _run(case, 0, data)
    new = _run(case, 1, data)
    tol = 4e-5 * max(1.0, float(ref.abs().max()) / 4)
    assert float((f32(new).double() - ref).abs().max()) < tol
    assert float((f32(old).double() - ref).abs().max()) < tol
    if not res:  # same MFMA sequence per output element
        assert torch.equal(old, new)
    for _ in range(3):
        assert torch.equal(_run(case, 1, data), new), "persistent kernel is not deterministic"
    # tail split: the last row blocks run as 64 x 64 tiles with the same compile-time epilogue -- same bits;
    # with a residual the persistent kernel adds it into the accumulators and the 64 x 64 kernel last: a rounding apart
    cut = _run(case, 33, data)
    assert torch.equal(cut, _run(case, 33, data))
    if res:
        assert float((cut.double() - ref).abs().max()) < tol
        assert float((cut - new).abs().max()) <= 4e-6 * max(1.0, float(ref.abs().max()))
    else:
        assert torch.equal(cut, new), "tail split changed the result"


def test_release_library_rejects_experiment_bits():
    """include/ctk.h v9: the release build accepts bits 0 and 5 of CTK_OPT_GEMM_PP only (no 'no stores', no jitter, no stream-K)."""
    import ctypes as C

    from cotracker_amd import _lib
    lib = _lib.load()
    if os.path.basename(_lib.LIB_PATH) != "libctk_hip.so":
        pytest.skip("running against a dev build")
    cur = C.c_int(0)
    for bad in (2, 8, 9, 16, 17, 64, 128, 256):
        assert lib.ctk_set_option(_lib.OPT_GEMM_PP, bad) == -2
        lib.ctk_gemm_pp_mode(bad)  # the pre-v9 name ignores an invalid mode
        _lib.check(lib.ctk_get_option(_lib.OPT_GEMM_PP, C.byref(cur)), "ctk_get_option")
        assert cur.value == 33
    assert not hasattr(lib, "ctk_debug_pp_clock") and not hasattr(lib, "ctk_debug_pp_trace")


def test_two_host_threads_two_streams_concurrently():
    """SURVEY 8(b) "re-entrant": two host threads enqueue different Linears on two streams at the same time, many times; every
    result equals the one the same call produces alone.  (Until round 5 every persistent launch wrote a device-side global,
    g_pp_clock, and the sampler / attention launchers called getenv() per launch.)"""
    import threading

    from cotracker_amd import ops
    cases = ["to_q", "to_kv"]
    made = {c: _make(c) for c in cases}
    alone = {c: _run(c, 33, made[c][5]) for c in cases}
    errs = []

    def worker(case):
        try:
            M, K, N, act, res, split, brows, bias = CASES[case]
            a_sh, w, wp, b, br, r = made[case][5]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(12):
                    out = ops.gemm(a_sh, w, bias=b, act=act, bias_rows=br, packed=wp, out_split=split)
                    if not torch.equal(out, alone[case]):  # (synchronises this stream only)
                        errs.append(f"{case}: result differs under concurrency")
                        return
        except Exception as e:  # noqa: BLE001
            errs.append(f"{case}: {type(e).__name__}: {e}")

    th = [threading.Thread(target=worker, args=(c,)) for c in cases]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs


DEV_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "co-tracker_amd", "libctk_hip_dev.so")


def _jitter_main():
    """Runs in a subprocess with CTK_LIB_PATH = the dev build: every case with and without timing jitter (mode bit 3), with and
    without the tail split; prints one line per case."""
    from cotracker_amd import _lib
    assert os.path.basename(_lib.LIB_PATH) == "libctk_hip_dev.so"
    for case in sorted(CASES):
        data = _make(case)[5]
        plain, split = _run(case, 1, data), _run(case, 33, data)
        for _ in range(3):
            assert torch.equal(_run(case, 9, data), plain), f"{case}: result depends on wave timing: LDS-DMA / barrier protocol race"
        assert torch.equal(_run(case, 41, data), split), f"{case}: tail split + timing jitter changed the result"
        print("jitter ok", case, flush=True)


def test_gemm_pp_survives_timing_jitter_in_the_dev_build():
    if not os.path.exists(DEV_LIB):
        pytest.skip("libctk_hip_dev.so not built (make -C co-tracker_amd/csrc dev)")
    env = dict(os.environ, CTK_LIB_PATH=DEV_LIB)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_gemm_pp as t; t._jitter_main()" % (root, os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("jitter ok") == len(CASES), out.stdout
