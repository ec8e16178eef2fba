"""The edge-case inputs of tests/test_gpu_parity.py::test_model_edge_cases_vs_torch_port, checked HERE (CPU, build container: needs
/root/reference) between the checker that test uses -- oracle/torch_port.py -- and the imported, unmodified reference: a single
point, a video shorter than one window, odd pyramid sizes, queries on / outside the border, queries entering in later windows with
a ragged last window, the offline model on an odd frame count.  So the GPU test's oracle is pinned on the reference for exactly
the inputs it is used on."""
import copy
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cotracker")), reason="reference checkout not present")

CASES = {  # name: (H, W, T, offline, queries)
    "one_point": (64, 96, 20, False, [[0.0, 41.3, 22.7]]),
    "short_video": (64, 96, 5, False, [[0.0, 10.0, 12.0], [2.0, 70.5, 40.25], [4.0, 33.0, 60.0]]),
    "odd_pyramid": (72, 104, 20, False, [[0.0, 10.0, 12.0], [5.0, 70.5, 40.25], [3.0, 103.0, 71.0], [1.0, 51.5, 35.5]]),
    "border_queries": (64, 96, 20, False, [[0.0, 0.0, 0.0], [0.0, 95.0, 63.0], [1.0, -3.0, 10.0], [2.0, 99.5, 70.0], [0.0, 47.5, 0.0], [3.0, 0.0, 31.5]]),
    "late_queries_sliding": (64, 96, 23, False, [[0.0, 10.0, 12.0], [5.0, 20.0, 50.0], [13.0, 80.0, 9.0], [21.0, 70.5, 40.25]]),
    "offline_odd": (64, 96, 11, True, [[0.0, 10.0, 12.0], [10.0, 70.5, 40.25], [4.0, 3.0, 60.0]]),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_torch_port_equals_reference_on_edge_cases(case):
    sys.path.insert(0, REF)
    try:
        from cotracker.models.core.cotracker.cotracker3_online import CoTrackerThreeOnline as RefOn
        from cotracker.models.core.cotracker.cotracker3_offline import CoTrackerThreeOffline as RefOff
    finally:
        sys.path.remove(REF)
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_
    from oracle import torch_port as TP
    H, W, T, offline, q = CASES[case]
    q = torch.tensor([q])
    torch.manual_seed(0)
    ref = (RefOff if offline else RefOn)(stride=4, corr_radius=3, window_len=8, model_resolution=(H, W)).eval()
    fill_synthetic_(ref, seed=7)
    video = synthetic_video(T, H, W, seed=11)
    with torch.no_grad():
        rc, rv, rf, _ = ref(video, q, iters=4)
    p = {k: v.clone() for k, v in ref.state_dict().items() if not k.startswith("fnet.")}
    pc, pv, pf = TP.model_forward(copy.deepcopy(ref.fnet), p, video, q, iters=4, window_len=8, offline=offline)
    assert torch.isfinite(rc).all()
    # same ATen kernels in the same order; what differs is blocking (the port encodes and correlates in other chunk sizes), i.e.
    # reduction order: measured 6e-5 ... 8e-5 px here, the size of the reference's own thread-count spread -- an order of
    # magnitude inside the 1e-3 px / 1e-4 logit bar the GPU test holds the HIP path to against this port
    dc = float((pc - rc).abs().max())
    dv = float((torch.sigmoid(pv) - rv).abs().max())
    df = float((torch.sigmoid(pf) - rf).abs().max())
    print(case, "port vs reference: px", dc, "vis", dv, "conf", df)
    assert dc <= 2e-4 and dv <= 2e-5 and df <= 2e-5, (dc, dv, df)
