"""ctypes binding of the C-ABI in include/ctk.h (libctk_hip.so, built by csrc/Makefile).

There is no CPU fallback: if the shared library is missing or cannot be loaded, importing any
op raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CTK_LIB_PATH: dev knob for A/B runs of two builds in one GPU session (tools/gpu_session.sh); never set in product use
LIB_PATH = os.environ.get("CTK_LIB_PATH") or os.path.join(_HERE, "libctk_hip.so")

LEVELS = 4
DEPTH = 3
CORR_LD = 2432
CORR_K = 2401
X_LD = 1120
X_DIM = 1110
HID = 384
MLP = 1536
VIRT = 64
ACT_NONE, ACT_GELU_ERF, ACT_GELU_TANH = 0, 1, 2
ABI_VERSION = 9
PAD_ZEROS, PAD_BORDER = 0, 1  # ctk_bilinear_sampler padding_mode
# ctk_set_option keys (include/ctk.h)
(OPT_GEMM_PP, OPT_GEMM_TAIL_PCT, OPT_CORR_VERSION, OPT_CORR_MAP, OPT_ATTENTION_VALU, OPT_ATTENTION_TIME_PERSISTENT,
 OPT_OVERLAP) = range(7)

_fp = C.c_void_p  # device pointers travel as integers


class BlockWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("wq", "bq", "wkv", "bkv", "wo", "bo", "w1", "b1", "w2", "b2", "ctx_gamma", "ctx_beta",
                                   "wq_p", "wkv_p", "wo_p", "w1_p", "w2_p")]


class ModelWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("corr_fc1_w", "corr_fc1_b", "corr_fc2_w", "corr_fc2_b", "in_w", "in_bias_t",
                                   "virtual_tokens", "head_w", "head_b", "corr_fc1_p", "corr_fc2_p", "in_p")] + [
        ("time_blocks", BlockWeights * DEPTH),
        ("virtual2point", BlockWeights * DEPTH),
        ("virtual_self", BlockWeights * DEPTH),
        ("point2virtual", BlockWeights * DEPTH),
    ]


WINDOW_NO_SPACE_ATTN = 1  # ctk_window_args.flags


class WindowArgs(C.Structure):
    _fields_ = [
        ("S", C.c_int32), ("N", C.c_int32), ("iters", C.c_int32),
        ("H", C.c_int32 * LEVELS), ("W", C.c_int32 * LEVELS),
        ("fmaps", _fp * LEVELS), ("support", _fp * LEVELS),
        ("point_mask", _fp),
        ("coords", _fp), ("vis", _fp), ("conf", _fp),
        ("scale_x", C.c_float), ("scale_y", C.c_float),
        ("points_per_chunk", C.c_int32),
        ("aux_stream", _fp),
        ("flags", C.c_int32),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", _fp), ("lda", C.c_int64), ("M", C.c_int32),
        ("W", _fp), ("ldw", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
        ("Wp", _fp),
        ("C", _fp), ("ldc", C.c_int64),
        ("bias", _fp),
        ("bias_rows", _fp), ("bias_period", C.c_int32),
        ("resid", _fp), ("ldr", C.c_int64),
        ("act", C.c_int32),
        ("batch", C.c_int32), ("a_bs", C.c_int64), ("c_bs", C.c_int64),
        ("k_valid", C.c_int32), ("a_split", C.c_int32), ("c_split", C.c_int32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", _fp), ("q_ld", C.c_int64), ("q_bs", C.c_int64), ("q_is", C.c_int64),
        ("k", _fp), ("v", _fp), ("kv_ld", C.c_int64), ("kv_bs", C.c_int64), ("kv_is", C.c_int64),
        ("out", _fp), ("o_ld", C.c_int64), ("o_bs", C.c_int64), ("o_is", C.c_int64),
        ("nbatch", C.c_int32), ("n1", C.c_int32), ("n2", C.c_int32),
        ("splits", C.c_int32), ("partial", _fp), ("o_split", C.c_int32),
        ("key_mask", _fp), ("query_mask", _fp),
    ]


class FormerWeights(C.Structure):
    """ctk_former_weights: the general update former (CoTracker2)."""
    _fields_ = [
        ("depth", C.c_int32), ("in_dim", C.c_int32), ("in_ld", C.c_int32), ("out_dim", C.c_int32), ("out_ld", C.c_int32),
        ("in_w", _fp), ("in_p", _fp), ("in_b", _fp), ("in_bias_t", _fp), ("virtual_tokens", _fp),
        ("head_w", _fp), ("head_p", _fp), ("head_b", _fp),
        ("time_blocks", C.POINTER(BlockWeights)), ("virtual2point", C.POINTER(BlockWeights)),
        ("virtual_self", C.POINTER(BlockWeights)), ("point2virtual", C.POINTER(BlockWeights)),
    ]


class V2WindowArgs(C.Structure):
    """ctk_v2_window_args: one CoTracker2 window (cotracker.py:86-173)."""
    _fields_ = [
        ("S", C.c_int32), ("N", C.c_int32), ("iters", C.c_int32),
        ("H", C.c_int32 * LEVELS), ("W", C.c_int32 * LEVELS),
        ("fmaps", _fp * LEVELS),
        ("coords", _fp), ("track_feat", _fp), ("vis", _fp), ("track_mask", _fp), ("point_mask", _fp), ("vis_out", _fp),
    ]


class V2Weights(C.Structure):
    """ctk_v2_weights."""
    _fields_ = [
        ("former", FormerWeights),
        ("pos_hwc", _fp), ("pos_h", C.c_int32), ("pos_w", C.c_int32),
        ("norm_w", _fp), ("norm_b", _fp),
        ("upd_w", _fp), ("upd_p", _fp), ("upd_b", _fp),
        ("vis_w", _fp), ("vis_b", _fp),
    ]


class ProfileRow(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/ctk.h declares: (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    "ctk_abi_version": (C.c_int, []),
    "ctk_error_string": (C.c_char_p, [C.c_int]),
    "ctk_forward_window_workspace_bytes": (C.c_int, [_P(WindowArgs), _P(C.c_size_t)]),
    "ctk_forward_window": (C.c_int, [_P(WindowArgs), _P(ModelWeights), _fp, C.c_size_t, _fp]),
    "ctk_window_graph_create": (C.c_int, [_P(WindowArgs), _P(ModelWeights), _fp, C.c_size_t, _P(C.c_void_p)]),
    "ctk_window_graph_launch": (C.c_int, [C.c_void_p, _fp]),
    "ctk_window_graph_nodes": (C.c_int, [C.c_void_p, _P(C.c_int64)]),
    "ctk_window_graph_destroy": (C.c_int, [C.c_void_p]),
    "ctk_corr_embed_workspace_bytes": (C.c_int, [_P(WindowArgs), _P(C.c_size_t)]),
    "ctk_corr_embed": (C.c_int, [_P(WindowArgs), _P(ModelWeights), _fp, _fp, C.c_size_t, _fp]),
    "ctk_corr_volume": (C.c_int, [_P(WindowArgs), _fp, _fp]),
    "ctk_corr_volume_sh_workspace_bytes": (C.c_int, [_P(WindowArgs), _P(C.c_size_t)]),
    "ctk_corr_volume_sh": (C.c_int, [_P(WindowArgs), _fp, _fp, C.c_size_t, _fp]),
    "ctk_assemble_tokens": (C.c_int, [_P(WindowArgs), _fp, C.c_int32, _fp]),
    "ctk_update_former_workspace_bytes": (C.c_int, [C.c_int32, C.c_int32, _P(C.c_size_t)]),
    "ctk_update_former": (C.c_int, [C.c_int32, C.c_int32, _fp, _P(ModelWeights), _fp, _fp, C.c_size_t, _fp]),
    "ctk_update_former_ex": (C.c_int, [C.c_int32, C.c_int32, _fp, C.c_int32, _P(FormerWeights), _fp, _fp, _fp, C.c_size_t, _fp]),
    "ctk_forward_window_v2_workspace_bytes": (C.c_int, [_P(V2WindowArgs), _P(V2Weights), _P(C.c_size_t)]),
    "ctk_forward_window_v2": (C.c_int, [_P(V2WindowArgs), _P(V2Weights), _fp, C.c_size_t, _fp]),
    "ctk_v2_window_graph_create": (C.c_int, [_P(V2WindowArgs), _P(V2Weights), _fp, C.c_size_t, _P(C.c_void_p)]),
    "ctk_v2_assemble": (C.c_int, [C.c_int32, C.c_int32, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int32, _fp, C.c_int32, _fp]),
    "ctk_v2_apply_delta": (C.c_int, [C.c_int32, C.c_int32, _fp, C.c_int32, _fp, _fp, _fp, C.c_float, _fp, _fp]),
    "ctk_v2_vis_head": (C.c_int, [_fp, C.c_int64, _fp, _fp, _fp, _fp]),
    "ctk_sample_features4d": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int32, _fp, _fp]),
    "ctk_bilinear_sampler": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int64, C.c_int32,
                                       C.c_int32, _fp, _fp]),
    "ctk_tap_indices": (C.c_int, [_P(WindowArgs), _fp, _fp]),
    "ctk_sample_patches": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int32, C.c_int32, _fp, _fp]),
    "ctk_sample_support": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, C.c_int32, _fp, _fp]),
    "ctk_corrblock_sample": (C.c_int, [_P(_fp), _P(C.c_int32), _P(C.c_int32), C.c_int32, C.c_int32, _fp, _fp, _fp, _fp]),
    "ctk_normalize_to_nhwc": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp]),
    "ctk_avg_pool2_nhwc": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp]),
    "ctk_conv2d_sh": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, _fp, _fp, _fp]),
    "ctk_enc_stem_im2col": (C.c_int, [_fp, C.c_int32, C.c_int32, C.c_int32, _fp, _fp]),
    "ctk_enc_inorm_workspace_bytes": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, _P(C.c_size_t)]),
    "ctk_enc_inorm_stats": (C.c_int, [_fp, C.c_int32, C.c_int64, C.c_int32, C.c_float, _fp, _fp, _fp]),
    "ctk_enc_inorm_apply": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int32, C.c_int64, C.c_int32, _fp, _fp, _fp]),
    "ctk_enc_fuse": (C.c_int, [_P(_fp), _P(C.c_int32), _P(C.c_int32), _P(C.c_int32), C.c_int32, C.c_int32, C.c_int32, _fp, _fp]),
    "ctk_enc_l2norm": (C.c_int, [_fp, C.c_int64, _fp, _fp]),
    "ctk_gemm": (C.c_int, [_P(GemmArgs), _fp]),
    "ctk_pack_weight_bytes": (C.c_int, [C.c_int32, C.c_int32, _P(C.c_size_t)]),
    "ctk_pack_weight": (C.c_int, [_fp, C.c_int64, C.c_int32, C.c_int32, _fp, _fp]),
    "ctk_layernorm": (C.c_int, [_fp, _fp, C.c_int64, _fp, _fp, C.c_float, C.c_int32, _fp]),
    "ctk_split_rows": (C.c_int, [_fp, C.c_int64, C.c_int64, C.c_int32, _fp, _fp]),
    "ctk_attention": (C.c_int, [_P(AttnArgs), _fp]),
    "ctk_profile_enable": (C.c_int, [C.c_int]),
    "ctk_gemm_pp_mode": (None, [C.c_int]),
    "ctk_set_option": (C.c_int, [C.c_int, C.c_int]),
    "ctk_get_option": (C.c_int, [C.c_int, _P(C.c_int)]),
    "ctk_profile_read": (C.c_int, [_P(ProfileRow), C.c_int, _P(C.c_int)]),
    "ctk_probe_mfma": (C.c_int, [C.c_int, C.c_int, _fp, _P(C.c_double), _fp]),
}

_lib = None


def load():
    """Load libctk_hip.so (once) and type every exported symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built.  Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C co-tracker_amd/csrc`).  There is no CPU fallback.")
    # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7).  It must be mapped BEFORE our
    # library so that both share ONE HIP runtime; loaded the other way round the process ends up
    # with two runtimes and ours sees no device (hipErrorNoDevice).
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.ctk_abi_version() != ABI_VERSION:
        raise RuntimeError("libctk_hip.so ABI version mismatch")
    _lib = lib
    return lib


class option:
    """`with option(OPT_CORR_VERSION, 1): ...` -- set a back-end option (include/ctk.h: ctk_set_option) for a scope and restore the
    previous value afterwards.  Process-wide (the table is one per loaded library): meant for tests and A/B tools."""

    def __init__(self, key: int, value: int):
        self.key, self.value, self.old = key, value, None

    def __enter__(self):
        lib = load()
        old = C.c_int(0)
        check(lib.ctk_get_option(self.key, C.byref(old)), "ctk_get_option")
        self.old = old.value
        check(lib.ctk_set_option(self.key, self.value), f"ctk_set_option({self.key}, {self.value})")
        return self

    def __exit__(self, *exc):
        load().ctk_set_option(self.key, self.old)
        return False


def check(rc: int, what: str):
    if rc != 0:
        msg = load().ctk_error_string(rc)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc} (code {rc})")
