"""Drop-in boundary proof against the imported reference (CPU, build container only: skipped where /root/reference is
absent, e.g. on the GPU box).  For every model the reference's ``build_cotracker`` can return, ours must
  * expose the identical ``state_dict`` key set with identical shapes and dtypes,
  * accept the reference's own state_dict with ``load_state_dict(strict=True)`` (and vice versa),
  * have identical ``__init__`` / ``forward`` signatures (names, order, defaults),
and the predictors / hub entry points must have the reference's signatures.
"""
import importlib
import inspect
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cotracker")), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REF)
    try:
        mods = dict(
            build=importlib.import_module("cotracker.models.build_cotracker"),
            predictor=importlib.import_module("cotracker.predictor"),
            online=importlib.import_module("cotracker.models.core.cotracker.cotracker3_online"),
            offline=importlib.import_module("cotracker.models.core.cotracker.cotracker3_offline"),
            v2=importlib.import_module("cotracker.models.core.cotracker.cotracker"),
        )
        spec = importlib.util.spec_from_file_location("ref_hubconf", os.path.join(REF, "hubconf.py"))
        hub = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(hub)
        mods["hub"] = hub
        yield mods
    finally:
        sys.path.remove(REF)


def sig(fn):
    return [(p.name, p.kind, p.default) for p in inspect.signature(fn).parameters.values()]


def ctor_sig(cls):
    """Constructor signature a caller sees: CoTrackerThreeOnline.__init__(self, **args) just forwards to its base
    (cotracker3_online.py:160-161), so walk the MRO to the first __init__ that names its parameters."""
    for c in cls.__mro__:
        init = c.__dict__.get("__init__")
        if init is None:
            continue
        s = sig(init)
        if any(k == inspect.Parameter.POSITIONAL_OR_KEYWORD and n != "self" for n, k, _ in s):
            return s
    return sig(cls.__init__)


CASES = [  # (reference module key, class name, ours, kwargs)
    ("online", "CoTrackerThreeOnline", "cotracker_amd.model", dict(window_len=16)),
    ("offline", "CoTrackerThreeOffline", "cotracker_amd.model", dict(window_len=60)),
    ("v2", "CoTracker2", "cotracker_amd.model_v2", dict(window_len=8)),
    # round 4: the two constructor variants that only change the PARAMETER SET (same kernels) -- cotracker3_online.py:43-53
    ("online", "CoTrackerThreeOnline", "cotracker_amd.model", dict(window_len=16, linear_layer_for_vis_conf=False)),
    ("offline", "CoTrackerThreeOffline", "cotracker_amd.model", dict(window_len=60, add_space_attn=False)),
    ("online", "CoTrackerThreeOnline", "cotracker_amd.model", dict(window_len=16, add_space_attn=False, linear_layer_for_vis_conf=False)),
]


@pytest.mark.parametrize("key,cls,ours_mod,kw", CASES)
def test_state_dict_and_signatures_match_reference(ref, key, cls, ours_mod, kw):
    R = getattr(ref[key], cls)
    O = getattr(importlib.import_module(ours_mod), cls)
    common = dict(stride=4, model_resolution=(384, 512), **kw)
    r, o = R(**common), O(**common)
    rs, os_ = r.state_dict(), o.state_dict()
    assert list(rs.keys()) == list(os_.keys()) or set(rs.keys()) == set(os_.keys())
    assert set(rs.keys()) == set(os_.keys())
    for k in rs:
        assert rs[k].shape == os_[k].shape and rs[k].dtype == os_[k].dtype, k
    if not any(k in kw for k in ("add_space_attn", "linear_layer_for_vis_conf")):
        assert len(rs) == (321 if key == "v2" else 188)
    if kw.get("add_space_attn") is False:
        assert not any(".space_" in k for k in os_)
    if kw.get("linear_layer_for_vis_conf") is False:
        assert os_["updateformer.flow_head.weight"].shape == (4, 384) and not any("vis_conf_head" in k for k in os_)
    # deterministic buffers carry the same values (sin/cos tables built independently)
    for k in ("time_emb", "pos_emb"):
        if k in rs:
            assert torch.allclose(rs[k], os_[k], atol=1e-6), k
    # checkpoints travel both ways
    assert o.load_state_dict(rs, strict=True) is not None
    r.load_state_dict(os_, strict=True)
    # signatures: constructor and forward
    assert ctor_sig(R) == ctor_sig(O)
    assert sig(R.forward) == sig(O.forward)
    for attr in ("model_resolution", "window_len", "stride"):
        assert getattr(r, attr) == getattr(o, attr)
    if key != "offline":
        assert sig(R.init_video_online_processing) == sig(O.init_video_online_processing)


def test_predictor_and_builder_signatures(ref):
    from cotracker_amd import predictor as P
    from cotracker_amd import build_cotracker as B
    for name in ("CoTrackerPredictor", "CoTrackerOnlinePredictor"):
        R, O = getattr(ref["predictor"], name), getattr(P, name)
        assert sig(R.__init__) == sig(O.__init__), name
        assert sig(R.forward) == sig(O.forward), name
    assert sig(ref["build"].build_cotracker) == sig(B.build_cotracker)
    assert sig(ref["predictor"].CoTrackerPredictor._compute_sparse_tracks) == sig(P.CoTrackerPredictor._compute_sparse_tracks)
    assert sig(ref["predictor"].CoTrackerPredictor._compute_dense_tracks) == sig(P.CoTrackerPredictor._compute_dense_tracks)


def test_hub_entry_points(ref):
    spec = importlib.util.spec_from_file_location("our_hubconf", os.path.join(os.path.dirname(os.path.dirname(__file__)), "hubconf.py"))
    ours = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ours)
    names = [n for n, f in vars(ref["hub"]).items() if inspect.isfunction(f) and n.startswith("cotracker") and f.__module__ == "ref_hubconf"]
    assert len(names) == 6
    for n in names:
        assert hasattr(ours, n), n
        assert sig(getattr(ref["hub"], n)) == sig(getattr(ours, n)), n


def test_reference_predictor_accepts_our_model(ref):
    """The seam of SURVEY §8b: a reference CoTrackerPredictor whose .model is swapped for ours (no GPU needed to swap)."""
    from cotracker_amd.model import CoTrackerThreeOffline
    p = ref["predictor"].CoTrackerPredictor(checkpoint=None, offline=True, window_len=60)
    ours = CoTrackerThreeOffline(window_len=60, stride=4, model_resolution=(384, 512))
    ours.load_state_dict(p.model.state_dict(), strict=True)
    p.model = ours.eval()
    assert p.interp_shape == ours.model_resolution
    with pytest.raises(RuntimeError, match="MI355X GPU only"):  # product path fails loudly off-GPU: no CPU fallback
        p(torch.zeros(1, 4, 3, 64, 64), grid_size=2)
