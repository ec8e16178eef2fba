"""bench.py contract pieces that need no GPU: workload table vs BASELINE.json, the roofline object, the CPU baseline
(numpy oracle on a bounded sample) and the PMC-traffic attachment rule."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_default_workload_is_the_baseline_config():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    H, W, T, G, offline, wl, desc = bench.WORKLOADS["c3_sliding"]
    assert (H, W, T, G * G) == (512, 512, 120, 6400) and "configs[2]" in desc
    assert "512" in base["configs"][2] and "6400" in base["configs"][2]
    old = sys.argv
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = old
    assert (a.gpus, a.workload, a.precision) == (1, "c3_sliding", "f16x3") and a.steps >= 1 and a.warmup >= 1


def test_roofline_entry_split_half_and_traffic():
    row = {"name": "gemm_sh_128x128", "launches": 10, "total_ms": 4.0, "flops": 1.0e12, "bytes": 8.0e9}
    traffic = {"gemm_sh_128x128": {"hbm_bytes_per_launch": 9.0e8, "fetch_bytes_per_launch": 6e8, "write_bytes_per_launch": 3e8,
                                   "dispatches": 10, "source": "test"}}
    r = bench.roofline_entry(row, traffic)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 250.0) < 1e-6 and abs(r["peak"] - 833.3) < 0.1     # 1e12 flop / 4 ms; 2500 / 3
    assert abs(r["frac"] - 250.0 / r["peak"]) < 1e-3 and r["mfma_issued"] == 750.0
    assert r["avg_launch_us"] == 400.0 and r["traffic"] == 9.0e8
    r32 = bench.roofline_entry(dict(row, name="gemm_f32_128x128"), {})
    assert r32["peak"] == 157.3 and r32["traffic"] is None
    hbm = bench.roofline_entry(dict(row, name="corr_volume_sh"), {}, force_hbm=True)
    assert hbm["bound"] == "hbm" and hbm["unit"] == "GB/s" and abs(hbm["achieved"] - 2000.0) < 1e-6 and hbm["peak"] == 8000.0


def test_cpu_baseline_is_the_oracle_on_a_bounded_sample():
    cb = bench.cpu_baseline(8, 224 / 120, n_points=4, reps=1)  # 8 frames x 4 points keeps this test at a few seconds
    assert cb["kind"] == "port" and cb["unit"] == "tracked-point-frames/s" and cb["cores"] >= 1 and cb["value"] > 0
    assert "numpy oracle" in cb["sample"]


def test_committed_pmc_traffic_is_tagged_with_its_workload():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert d["_workload"] == "c3_sliding"
    assert d["gemm_sh_128x128"]["hbm_bytes_per_launch"] > 0 and "FETCH_SIZE x2" in d["gemm_sh_128x128"]["source"]
