"""TAP-Vid metrics (cotracker_amd.evaluation.compute_tapvid_metrics) -- CPU.  Known-answer cases worked by hand, and,
where the reference checkout is present (build container), agreement with the reference's own function
(cotracker/evaluation/core/eval_utils.py:12-138) on random inputs incl. both query modes and multi-video batches."""
import importlib.util
import os

import numpy as np
import pytest

from cotracker_amd.evaluation import compute_tapvid_metrics

REF = "/root/reference/cotracker/evaluation/core/eval_utils.py"


def test_known_answers_first_mode():
    # one video, two points, four frames.  Point 0 queried at t=0, point 1 at t=2 ("first": only later frames count)
    q = np.array([[[0, 0, 0], [2, 0, 0]]], dtype=np.float64)
    gt = np.zeros((1, 2, 4, 2))
    gt_occ = np.zeros((1, 2, 4), bool)
    gt_occ[0, 0, 3] = True                      # point 0 occluded in the last frame
    pred = gt.copy()
    pred[0, 0, 1] = [0.5, 0.0]                  # 0.5 px off: within every threshold
    pred[0, 0, 2] = [3.0, 0.0]                  # 3 px off: within 4, 8, 16 only
    pred[0, 1, 3] = [20.0, 0.0]                 # 20 px off: outside every threshold
    pred_occ = np.zeros((1, 2, 4), bool)
    pred_occ[0, 0, 3] = True                    # correct occlusion call
    m = compute_tapvid_metrics(q, gt_occ, gt, pred_occ, pred, "first")
    # evaluated: point 0 frames 1,2,3; point 1 frame 3  -> 4 points, all occlusion calls right
    assert m["occlusion_accuracy"][0] == 1.0
    # visible evaluated GT points: p0 f1, p0 f2, p1 f3 = 3
    np.testing.assert_allclose(m["pts_within_1"], [1 / 3])
    np.testing.assert_allclose(m["pts_within_2"], [1 / 3])
    np.testing.assert_allclose(m["pts_within_4"], [2 / 3])
    np.testing.assert_allclose(m["pts_within_16"], [2 / 3])
    # jaccard_1: TP = 1 (p0 f1); FP = predicted visible but not a hit = p0 f2, p1 f3 = 2 -> 1 / (3 + 2)
    np.testing.assert_allclose(m["jaccard_1"], [1 / 5])
    np.testing.assert_allclose(m["jaccard_4"], [2 / (3 + 1)])
    np.testing.assert_allclose(m["average_pts_within_thresh"], [(1 / 3 + 1 / 3 + 2 / 3 * 3) / 5])


def test_strided_mode_counts_every_frame_but_the_query():
    q = np.array([[[1, 0, 0]]], dtype=np.float64)
    gt = np.zeros((1, 1, 3, 2))
    occ = np.zeros((1, 1, 3), bool)
    pred = gt.copy()
    pred[0, 0, 1] = [100.0, 100.0]              # the query frame itself is never evaluated
    m = compute_tapvid_metrics(q, occ, gt, occ.copy(), pred, "strided")
    assert m["pts_within_1"][0] == 1.0 and m["average_jaccard"][0] == 1.0
    with pytest.raises(ValueError):
        compute_tapvid_metrics(q, occ, gt, occ, pred, "last")


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present")
@pytest.mark.parametrize("mode", ["first", "strided"])
def test_agrees_with_the_reference_function(mode):
    spec = importlib.util.spec_from_file_location("ref_eval_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    r = np.random.RandomState(0)
    for b, n, t in [(1, 7, 12), (3, 20, 9)]:
        q = np.stack([r.randint(0, t, size=(b, n)).astype(np.float64), r.uniform(0, 255, (b, n)), r.uniform(0, 255, (b, n))], -1)
        gt = r.uniform(0, 255, (b, n, t, 2))
        pred = gt + r.standard_normal((b, n, t, 2)) * r.choice([0.3, 1.5, 5.0, 12.0, 30.0], size=(b, n, t, 1))
        gt_occ = r.uniform(size=(b, n, t)) < 0.3
        pred_occ = gt_occ ^ (r.uniform(size=(b, n, t)) < 0.2)
        a = compute_tapvid_metrics(q, gt_occ, gt, pred_occ, pred, mode)
        e = ref.compute_tapvid_metrics(q, gt_occ, gt, pred_occ, pred, mode)
        assert set(a) == set(e)
        for k in e:
            np.testing.assert_allclose(a[k], e[k], rtol=0, atol=1e-12, err_msg=k)


def test_grid_helper_matches_reference_with_center():
    mu = "/root/reference/cotracker/models/core/model_utils.py"
    if not os.path.exists(mu):
        pytest.skip("reference checkout not present")
    import sys
    sys.path.insert(0, "/root/reference")
    try:
        from cotracker.models.core.model_utils import get_points_on_a_grid as ref_grid
    finally:
        sys.path.remove("/root/reference")
    import torch
    from cotracker_amd.predictor import get_points_on_a_grid
    for size, extent, center in [(8, (50, 50), [123.4, 301.7]), (5, (384, 512), None), (80, (384, 512), None), (1, (10, 20), None)]:
        assert torch.equal(get_points_on_a_grid(size, extent, center), ref_grid(size, extent, center))
