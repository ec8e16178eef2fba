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


def test_roofline_objects():
    row = {"name": "gemm_sh_128_k384_n384", "launches": 10, "total_ms": 4.0, "flops": 1.0e12, "bytes": 8.0e9}
    traffic = {row["name"]: {"hbm_bytes_per_launch": 9.0e8, "fetch_bytes_per_launch": 6e8, "write_bytes_per_launch": 3e8,
                             "dispatches": 10, "source": "test"}}
    r = bench.roofline_mfma(row["name"], [row], traffic)
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["achieved"] - 250.0) < 1e-6 and abs(r["peak"] - 833.3) < 0.1     # 1e12 flop / 4 ms; 2500 / 3
    assert abs(r["frac"] - 250.0 / r["peak"]) < 1e-3 and r["mfma_issued"] == 750.0
    assert r["avg_launch_us"] == 400.0 and r["traffic"] == 9.0e8
    r32 = bench.roofline_mfma("gemm_f32_128x128", [dict(row, name="gemm_f32_128x128")], {})
    assert r32["peak"] == 157.3 and r32["traffic"] is None
    both = bench.roofline_mfma("gemm (all Linear launches, call-weighted)", [row, dict(row, total_ms=12.0)], traffic)
    assert abs(both["achieved"] - 2.0e12 / 16e-3 / 1e12) < 1e-6 and both["launches"] == 20
    assert both["traffic"] == 9.0e8  # every row of the class has a PMC entry: call-weighted bytes per launch
    other = dict(row, name="gemm_sh_pp192_k384_n384")
    assert bench.roofline_mfma("gemm (all Linear launches, call-weighted)", [row, other], traffic)["traffic"] is None
    # the sustained-clock ceiling: issued rate / what the probe held
    rs = bench.roofline_mfma(row["name"], [row], traffic, {"f16_tflops": 1500.0, "clock_ghz": 1.43})
    assert abs(rs["frac_at_sustained_clock"] - 750.0 / 1500.0) < 1e-4 and rs["sustained_mfma_clock_ghz"] == 1.43
    # the sampler is priced against max(measured HBM time, MFMA time), never against its no-reuse algorithmic bytes
    srow = {"name": "corr_volume_sh", "launches": 2, "total_ms": 5.6, "flops": 2 * 2.52e11, "bytes": 2 * 1.8e10}
    st = {"corr_volume_sh": {"hbm_bytes_per_launch": 5.94e9, "fetch_bytes_per_launch": 2.22e9, "write_bytes_per_launch": 3.72e9}}
    s_ = bench.roofline_sampler(srow, st)
    assert s_["bound"] == "hbm" and abs(s_["frac"] - (5.94e9 / 8e12) / 2.8e-3) < 1e-3 and abs(s_["hbm_frac"] - 0.2652) < 1e-3
    assert s_["algorithmic_GBs_no_reuse"] > 6000 and s_["mfma_frac"] < 0.12
    assert bench.roofline_sampler(srow, {})["bound"] == "mfma"  # no PMC pass: HBM side unknown, say so
    assert "unmeasured" in bench.roofline_sampler(srow, {})["note"]


def test_split_half_pricing_follows_the_back_end_not_a_name_prefix():
    """Round-3 defect: conv_pp128_* (3 f16 MFMAs per product like every split-half kernel) was priced against the exact-f32
    peak because `split` was keyed on a list of name prefixes -> frac 0.93 instead of 0.18."""
    conv = {"name": "conv_pp128_3x3_s1_c64_n64", "launches": 32, "total_ms": 11.2, "flops": 32 * 51.3e9, "bytes": 1e9}
    r = bench.roofline_mfma(conv["name"], [conv], {})
    assert abs(r["peak"] - 833.3) < 0.1 and r["frac"] < 0.2 and r["mfma_issued"] == round(3 * r["achieved"], 1)
    for name in ("gemm_sh_pp192_k384_n384", "gemm_f16x3_128x128", "corr_volume_sh", "attention_time", "conv_pp128_1x1_s1_c160_n64"):
        assert not bench.is_exact_f32(name), name
    for name in ("gemm_f32_128x128", "gemm_f32_64x64", "corr_volume", "corrblock_sample"):
        assert bench.is_exact_f32(name) and bench.mfma_peak(name) == 157.3, name


def test_roofline_names_the_time_dominant_class():
    """One sampler row (206 ms) must not out-rank 839 ms of GEMM launches spread over per-shape rows."""
    rows = [{"name": "corr_volume_sh", "launches": 84, "total_ms": 206.0, "flops": 84 * 2.52e11, "bytes": 84 * 1.8e10}]
    for k, ms in ((384, 130.0), (1536, 400.0), (2432, 150.0), (768, 160.0)):
        rows.append({"name": f"gemm_sh_pp192_k{k}_n384", "launches": 500, "total_ms": ms, "flops": 500 * 5e10, "bytes": 1e9})
    rows.append({"name": "layernorm", "launches": 2520, "total_ms": 75.0, "flops": 1e9, "bytes": 1e12})
    r = bench.rooflines(rows, {})
    assert r["roofline"]["dominant_class"] == "gemm" and r["roofline"]["kernel"].startswith("gemm (all Linear")
    assert r["roofline"]["total_ms"] == 840.0 and len(r["roofline_gemm"]["rows"]) == 4
    s_ = r["roofline_sampler"]
    assert s_["kernel"] == "corr_volume_sh" and "frac_no_reuse" not in s_  # no PMC entry: MFMA side only
    tr = {"corr_volume_sh": {"hbm_bytes_per_launch": 6.5e9}}
    s2 = bench.rooflines(rows, tr)["roofline_sampler"]
    assert abs(s2["frac_no_reuse"] - 1.8e10 / (206e-3 / 84) / 8e12) < 1e-3 and s2["frac_measured"] == s2["frac"] < s2["frac_no_reuse"]
    # a sampler-dominated step (few GEMM rows) names the sampler
    assert bench.rooflines(rows[:2], {})["roofline"]["dominant_class"] == "sampler"


def test_cpu_baseline_states_the_host_cpu():
    c = bench.cpu_model()
    assert c["logical_cpus"] == os.cpu_count() and (c["model"] is None or isinstance(c["model"], str))


def test_cpu_baseline_is_the_torch_port_on_a_bounded_sample(monkeypatch):
    """The baseline leg runs oracle/torch_port.py in a subprocess (ATen CPU kernels, encoder included); here on the
    smoke-sized workload so the CPU suite stays short."""
    monkeypatch.setitem(bench.WORKLOADS, "tiny_cpu", (64, 64, 8, 3, True, 60, "test"))
    import oracle.torch_port as TP
    r = TP.bench("offline", 6, 3, 64, 2)
    assert r["tracked_point_frames_per_s"] > 0 and r["points"] == 9 and r["threads"] == 2
    base = bench.cpu_baseline("c2_offline") if os.environ.get("CTK_SLOW_TESTS") else None
    if base is not None:
        assert base["kind"] == "port-torch" and base["value"] > 0


def test_workload_table_covers_every_baseline_config():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "70225" in base["configs"][4] and bench.WORKLOADS["c5_shard"][3] == 265
    from cotracker_amd.sharding import chunk_bounds
    assert chunk_bounds(265 * 265, 8, 0) == (0, 8779)
    for name in ("c2_offline", "c3_sliding", "c4_online", "c5_shard"):
        assert name in bench.WORKLOADS


def test_committed_pmc_traffic_is_tagged_with_its_workload():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert d["_workload"] == "c3_sliding"
    rows = [v for k, v in d.items() if k.startswith("gemm_sh")]
    assert rows and rows[0]["hbm_bytes_per_launch"] > 0 and "FETCH_SIZE x2" in rows[0]["source"]
