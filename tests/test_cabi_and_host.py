"""CPU-side checks: the C-ABI library loads and exports every symbol include/ctk.h declares (no compute
without a GPU), argument validation returns error codes, and the host mirror keeps the reference's
surface (state_dict keys, packed-weight layout maths)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from cotracker_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from cotracker_amd import _lib
    header = open(os.path.join(ROOT, "include", "ctk.h")).read()
    declared = set(re.findall(r"\b(ctk_[a-z0-9_]+)\s*\(", header))
    declared -= {"ctk_block_weights", "ctk_model_weights"}
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), f"binding/header mismatch: {declared ^ set(_lib.SYMBOLS)}"
    for name in declared:
        assert hasattr(lib, name)
    assert lib.ctk_abi_version() == _lib.ABI_VERSION == 9


def test_release_library_has_no_debug_symbols_and_only_documented_knobs(lib):
    """Round 6: the library the package loads is a RELEASE build -- no ctk_debug_* entry point, no experiment
    knob; the only CTK_* names inside it are the seven option variables include/ctk.h documents (read once at load)."""
    import subprocess

    from cotracker_amd import _lib
    if os.path.basename(_lib.LIB_PATH) != "libctk_hip.so":
        pytest.skip("CTK_LIB_PATH points at another build")
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("ctk_")}
    assert exported == set(_lib.SYMBOLS), exported ^ set(_lib.SYMBOLS)
    assert not [x for x in exported if "debug" in x]
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(re.findall(rb"CTK_[A-Z0-9_]{3,}", blob))
    documented = {b"CTK_GEMM_PP", b"CTK_GEMM_TAIL_PCT", b"CTK_CORR", b"CTK_CORR_MAP", b"CTK_ATTN", b"CTK_ATTN_TIME", b"CTK_OVERLAP"}
    assert names <= documented, names - documented
    header = open(os.path.join(ROOT, "include", "ctk.h")).read()
    for n in documented:
        assert n.decode() in header


def test_options_are_validated(lib):
    from cotracker_amd import _lib as L
    v = C.c_int(-1)
    defaults = {L.OPT_GEMM_PP: 33, L.OPT_GEMM_TAIL_PCT: 25, L.OPT_CORR_VERSION: 3, L.OPT_CORR_MAP: 3, L.OPT_ATTENTION_VALU: 0,
                L.OPT_ATTENTION_TIME_PERSISTENT: 1, L.OPT_OVERLAP: 0}
    for k, d in defaults.items():
        assert lib.ctk_get_option(k, C.byref(v)) == 0 and v.value == d, (k, v.value)
    assert lib.ctk_get_option(0, None) == -1 and lib.ctk_get_option(7, C.byref(v)) == -2 and lib.ctk_get_option(-1, C.byref(v)) == -2
    for k, bad in ((L.OPT_CORR_VERSION, 2), (L.OPT_CORR_VERSION, 0), (L.OPT_CORR_VERSION, 4), (L.OPT_CORR_MAP, 5), (L.OPT_ATTENTION_VALU, 2),
                   (L.OPT_OVERLAP, 4), (L.OPT_OVERLAP, 8), (L.OPT_GEMM_TAIL_PCT, 101), (L.OPT_GEMM_PP, -1), (7, 0)):
        assert lib.ctk_set_option(k, bad) == -2, (k, bad)
    with L.option(L.OPT_CORR_VERSION, 1):
        assert lib.ctk_get_option(L.OPT_CORR_VERSION, C.byref(v)) == 0 and v.value == 1
    assert lib.ctk_get_option(L.OPT_CORR_VERSION, C.byref(v)) == 0 and v.value == 3


def test_argument_validation_without_gpu(lib):
    from cotracker_amd import _lib as L
    assert lib.ctk_error_string(0) == b"ok"
    n = C.c_size_t(0)
    assert lib.ctk_update_former_workspace_bytes(16, 6400, C.byref(n)) == 0
    R = (6400 + 64) * 16
    assert n.value >= R * 4 * (384 * 3 + 1152 + 1536)
    assert lib.ctk_update_former_workspace_bytes(0, 10, C.byref(n)) == -2  # CTK_E_SHAPE
    g = L.GemmArgs()
    assert lib.ctk_gemm(C.byref(g), None) == -1  # CTK_E_NULL
    g.A, g.W, g.C = 16, 16, 16
    g.M, g.N, g.K, g.lda, g.ldw, g.ldc = 8, 60, 32, 32, 32, 60
    assert lib.ctk_gemm(C.byref(g), None) == -2  # N % 64 != 0
    g.N, g.A = 64, 20
    assert lib.ctk_gemm(C.byref(g), None) == -3  # misaligned A
    a = L.WindowArgs()
    a.S, a.N, a.iters = 16, 100, 6
    assert lib.ctk_forward_window_workspace_bytes(C.byref(a), C.byref(n)) == 0 and n.value > 0
    assert lib.ctk_forward_window(C.byref(a), None, None, 0, None) == -1
    h = C.c_void_p()
    assert lib.ctk_window_graph_create(C.byref(a), None, None, 0, C.byref(h)) == -1 and not h.value  # validated before capture
    assert lib.ctk_window_graph_launch(None, None) == -1
    assert lib.ctk_window_graph_destroy(None) == 0


def test_v2_driver_argument_validation_without_gpu(lib):
    """ctk_forward_window_v2 / ctk_v2_window_graph_create reject bad arguments before touching the device."""
    from cotracker_amd import _lib as L
    a, w, n = L.V2WindowArgs(), L.V2Weights(), C.c_size_t(0)
    assert lib.ctk_forward_window_v2_workspace_bytes(None, None, C.byref(n)) == -1
    a.S, a.N, a.iters = 8, 10, 4
    assert lib.ctk_forward_window_v2_workspace_bytes(C.byref(a), C.byref(w), C.byref(n)) == -2   # former dims unset
    w.former.in_dim, w.former.in_ld, w.former.out_dim, w.former.out_ld = 456, 480, 130, 192
    assert lib.ctk_forward_window_v2_workspace_bytes(C.byref(a), C.byref(w), C.byref(n)) == -1   # NULL tensors
    assert lib.ctk_forward_window_v2(C.byref(a), C.byref(w), None, 0, None) == -1
    h = C.c_void_p()
    assert lib.ctk_v2_window_graph_create(C.byref(a), C.byref(w), None, 0, C.byref(h)) == -1 and not h.value


def test_struct_sizes_match_header(tmp_path):
    """ctypes mirrors vs the C compiler's view of include/ctk.h: sizeof of every struct that crosses the boundary."""
    import subprocess
    from cotracker_amd import _lib as L
    pairs = {"ctk_block_weights": L.BlockWeights, "ctk_model_weights": L.ModelWeights, "ctk_window_args": L.WindowArgs,
             "ctk_gemm_args": L.GemmArgs, "ctk_attn_args": L.AttnArgs, "ctk_former_weights": L.FormerWeights,
             "ctk_v2_window_args": L.V2WindowArgs, "ctk_v2_weights": L.V2Weights, "ctk_profile_row": L.ProfileRow}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "ctk.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in pairs) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, cls in pairs.items():
        assert C.sizeof(cls) == int(out[n]), (n, C.sizeof(cls), out[n])
    assert C.sizeof(L.BlockWeights) == 17 * 8


def test_integration_md_binding_stub_matches_the_header():
    """the ctypes mirror INTEGRATION.md tells a maintainer to write must be the struct the library reads --
    round 3's stub stopped at aux_stream (168 bytes) while ctk_forward_window reads `flags` at offset 168 of 176.  The stub is
    executed as written and compared with the typed binding, field by field; unknown flag bits are refused by the library."""
    import re
    from cotracker_amd import _lib as L
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# cotracker/models/core/cotracker/_ctk_binding\.py.*?)```", md, re.S).group(1)
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    stub = ns["ctk_window_args"]
    assert C.sizeof(stub) == C.sizeof(L.WindowArgs) == 176
    assert [(n, getattr(stub, n).offset, getattr(stub, n).size) for n, _ in stub._fields_] == \
           [(n, getattr(L.WindowArgs, n).offset, getattr(L.WindowArgs, n).size) for n, _ in L.WindowArgs._fields_]
    assert f"ctk_abi_version() == {L.ABI_VERSION}" in block
    lib = L.load()
    a = L.WindowArgs()
    a.S, a.N, a.iters = 8, 4, 1
    nb = C.c_size_t()
    assert lib.ctk_forward_window_workspace_bytes(C.byref(a), C.byref(nb)) == 0
    a.flags = 2  # an unknown bit (what garbage past a short struct looks like)
    assert lib.ctk_forward_window_workspace_bytes(C.byref(a), C.byref(nb)) == -2  # CTK_E_SHAPE
    a.flags = L.WINDOW_NO_SPACE_ATTN
    assert lib.ctk_forward_window_workspace_bytes(C.byref(a), C.byref(nb)) == 0


def test_no_product_import_of_oracle():
    pkg = os.path.join(ROOT, "co-tracker_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("oracle/", "ORACLE_DIR_MENTION"), f


def test_model_fails_loudly_on_cpu_tensors():
    from cotracker_amd.build_cotracker import build_cotracker
    m = build_cotracker(None, offline=False, window_len=8).eval()
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 8, 3, 64, 64), torch.zeros(1, 2, 3))
    v2 = build_cotracker(None, v2=True, window_len=8).eval()  # CoTracker2: same contract
    assert len(v2.state_dict()) == 321 and v2.window_len == 8 and v2.model_resolution == (384, 512)
    with pytest.raises(RuntimeError, match="GPU only"):
        v2(torch.zeros(1, 8, 3, 64, 64), torch.zeros(1, 2, 3))


def test_time_embedding_fold_matches_reference_order():
    """W_ours @ x_ours + (W e_t + b) == W_ref @ (x_ref + e_t) + b for the column permutation we use."""
    from cotracker_amd.model import CoTrackerThreeOnline, PackedWeights
    from cotracker_amd.weights import fill_synthetic_
    m = CoTrackerThreeOnline(window_len=8).eval()
    fill_synthetic_(m, seed=9)
    pw = PackedWeights.__new__(PackedWeights)  # exercise the maths without a device
    sd = m.state_dict()
    w_ref, b = sd["updateformer.input_transform.weight"].double(), sd["updateformer.input_transform.bias"].double()
    te = sd["time_emb"][0].double()
    x_ref = torch.randn(8, 1110, dtype=torch.double)
    ref = (x_ref + te) @ w_ref.t() + b
    in_w = torch.zeros(384, 1120, dtype=torch.double)
    in_w[:, 0:1024] = w_ref[:, 2:1026]
    in_w[:, 1024:1026] = w_ref[:, 0:2]
    in_w[:, 1026:1110] = w_ref[:, 1026:1110]
    x = torch.zeros(8, 1120, dtype=torch.double)
    x[:, 0:1024], x[:, 1024:1026], x[:, 1026:1110] = x_ref[:, 2:1026], x_ref[:, 0:2], x_ref[:, 1026:1110]
    ours = x @ in_w.t() + (te @ w_ref.t() + b)
    assert float((ours - ref).abs().max()) < 1e-10
    del pw


def test_synthetic_inputs_are_deterministic():
    from cotracker_amd.synthetic import synthetic_video
    a, b = synthetic_video(3, 32, 48, seed=5), synthetic_video(3, 32, 48, seed=5)
    assert a.shape == (1, 3, 3, 32, 48) and torch.equal(a, b) and float(a.min()) >= 0 and float(a.max()) <= 255


def test_tail_aliases_cannot_prove_overlap_for_inference_tensors():
    """Streaming feature cache (model.online_feature_cache): the host-side overlap proof reads the autograd version counter.
    Tensors allocated under torch.inference_mode() have none (`_version` raises): the proof must answer "not proven" -- the
    chunk is then re-encoded -- instead of raising inside the predictor (advisor finding, round 4)."""
    import torch

    from cotracker_amd.model import tail_aliases

    v = torch.zeros(1, 16, 3, 8, 8)
    a = v[:, 0:8]
    a._ctk_version = a._version
    assert tail_aliases(a, v[:, 4:12], 1, 4)
    v.add_(1.0)  # a write through a tensor sharing the version counter: stale
    assert not tail_aliases(a, v[:, 4:12], 1, 4)
    with torch.inference_mode():
        w = torch.zeros(1, 16, 3, 8, 8)
        assert tail_aliases(w[:, 0:8], w[:, 4:12], 1, 4) is False
        assert tail_aliases(w[:, 0:8], w[:, 5:13], 1, 4) is False


def test_bilinear_sampler_rejects_more_points_than_a_grid_can_cover():
    """ctk_bilinear_sampler launches one thread per sample, 256 per workgroup: gridDim.x * blockDim.x must stay below 2^32, so
    P > (2^24 - 1) * 256 is CTK_E_SHAPE before anything is launched (no GPU needed: the check precedes every HIP call)."""
    import ctypes as C
    from cotracker_amd import _lib as L
    lib = L.load()
    buf = (C.c_float * 4)()
    ptr = C.cast(buf, C.c_void_p)
    fn = lib.ctk_bilinear_sampler
    too_many = ((1 << 24) - 1) * 256 + 1
    rc = fn(ptr, 1, 1, 0, 2, 2, ptr, C.c_int64(too_many), 1, 1, ptr, None)
    assert rc == -2  # CTK_E_SHAPE (include/ctk.h)
    assert fn(ptr, 1, 1, 0, 2, 2, ptr, C.c_int64(0), 1, 1, ptr, None) == -2  # P <= 0
