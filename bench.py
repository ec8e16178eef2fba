#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X CoTracker3 hot path.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one ``CoTrackerPredictor.forward`` (resize -> CNN encoder -> pyramid -> all windows x 6
update iterations -> post-processing) on a synthetic video that is already resident in HBM.
Default workload = BASELINE.json configs[2]: 512x512, T=120, N=6400 (grid_size=80), the online
weights in sliding-window mode (``offline=False, window_len=16``: 14 windows of 16 frames).
metric = tracked-point-frames/s = N*T / seconds per step, summed over ranks (weak scaling: every
rank tracks its own 6400-point chunk; results are all-gathered inside the timed region).

Prints ONE JSON line (rank 0) with the contract keys plus
  roofline     -- the TIME-DOMINANT CLASS of kernels of the step (all Linear GEMM launches call-weighted, or the correlation
                  sampler): algorithmic flops / HIP-event duration vs the split-half MFMA ceiling (2500/3 TF/s), with
                  frac_at_sustained_clock = issued f16 MFMA rate / the rate ctk_probe_mfma(kind 2) holds in THIS run
                  (sustained_mfma: TF/s and the clock it implies); traffic = call-weighted HBM bytes per launch from the committed
                  rocprofv3 --pmc passes (profiles/pmc_traffic.json, stamped with the library hash; not re-measured here)
  roofline_gemm / roofline_sampler -- both classes under their own keys; the sampler with frac_measured (max(measured HBM
                  bytes / 8 TB/s, flops / MFMA ceiling) / launch time) and frac_no_reuse (SURVEY 8d's algorithmic bytes)
  extra_lines  -- the same protocol on the exact-f32 back end (value_f32), BASELINE configs[1] (C2) and configs[3] (C4, feature
                  cache off = reference behaviour, and on)
  cpu_baseline -- oracle/torch_port.py (the reference's ATen CPU kernels in the reference's order, encoder included)
                  timed on this host's cores (CPU model stated) on a bounded sample of the same workload
  parity       -- max-abs error of THIS run's tracks / logits against the unmodified reference's CPU outputs at
                  BASELINE scale (tests/golden/scale_*.npz), next to the reference's own thread-count noise
  kernels      -- per-kernel launch counts / avg duration / achieved rate from the same HIP events
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16), no sparsity
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (H, W, T, grid, offline, window_len, description)
    "c3_sliding": (512, 512, 120, 80, False, 16, "512x512 T=120 N=6400 cotracker3 online-weights sliding window S=16 (BASELINE.json configs[2])"),
    "c3_offline": (512, 512, 120, 80, True, 60, "512x512 T=120 N=6400 cotracker3_offline single window S=120"),
    "c2_offline": (256, 256, 48, 20, True, 60, "256x256 T=48 N=400 cotracker3_offline (BASELINE.json configs[1])"),
    # BASELINE.json configs[0] stand-in: assets/apple.mp4 is 1296x720, 50 frames, but there is no video decoder offline ->
    # a synthetic video of exactly that geometry through the hub recipe (cotracker3_offline, grid_size=10)
    "c1_standin": (720, 1296, 50, 10, True, 60, "720x1296 T=50 N=100 cotracker3_offline, grid_size=10 (BASELINE.json configs[0] stand-in: apple.mp4's geometry, synthetic pixels)"),
    # the offline single-window mode at the size the reference itself was run at (62 GB host): parity of the timed step
    "c3_offline_g40": (512, 512, 120, 40, True, 60, "512x512 T=120 N=1600 cotracker3_offline single window S=120"),
    "c4_online": (512, 512, 0, 32, False, 16, "512x512 cotracker3_online streaming, 16-frame chunks advancing 8, N=1024, "
                  "window replayed as ONE hipGraph per chunk (BASELINE.json configs[3])"),
    # BASELINE.json configs[4]: the 265x265 quasi-dense grid (N=70 225) in 8 contiguous chunks (sharding.chunk_bounds),
    # one chunk of <= 8 779 points per GPU; rank r tracks chunk r, so 1 GPU times chunk 0 (the per-GPU number) and
    # 8 GPUs the whole job.  grid = 265 is the JOINT grid; points_per_gpu is reported separately.
    "c5_shard": (512, 512, 120, 265, False, 16, "512x512 T=120 quasi-dense N=70225 (265x265 grid) in 8 contiguous point chunks of "
                 "<=8779, one per GPU, cotracker3 online-weights sliding window S=16 (BASELINE.json configs[4])"),
    "tiny": (128, 160, 24, 8, False, 8, "smoke-sized"),
    # CoTracker2 (hub entry point cotracker2: window 8, sliding) on the configs[2] video -- not a BASELINE config, the
    # measured line of SURVEY 8f-3
    "v2_sliding": (512, 512, 120, 80, "v2", 8, "512x512 T=120 N=6400 cotracker2 (window 8, sliding): CorrBlock sampler + "
                   "6+6-layer update former with attention masks on the same kernels"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3_sliding", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extra-lines", action="store_true", help="skip the extra timed rows (exact-f32 back end, C2, C4) of the default run")
    ap.add_argument("--no-graph", action="store_true", help="c4_online: direct launches instead of the captured hipGraph")
    ap.add_argument("--feature-cache", action="store_true", help="c4_online: opt in to model.online_feature_cache (re-use the previous "
                    "chunk's features for the overlapping frames; the reference re-encodes them)")
    ap.add_argument("--pmc-traffic", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                    help="per-kernel HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_traffic.py)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only to exercise the "
                         "multi-rank path on a box with fewer GPUs than ranks, together with --single-device)")
    ap.add_argument("--single-device", action="store_true",
                    help="dev/test: every rank uses cuda:0 (rank-sharded code path on a 1-GPU box; numbers are meaningless)")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32"],
                    help="Linear back end: split-half MFMA (3 f16 MFMAs per product, fp32-class accuracy) or exact-f32 MFMA")
    return ap.parse_args()


def cpu_model():
    """The host CPU as /proc/cpuinfo names it (model string, sockets, physical cores, logical CPUs)."""
    info = {"model": None, "sockets": None, "physical_cores": None, "logical_cpus": os.cpu_count()}
    try:
        phys, cores = set(), set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and info["model"] is None:
                info["model"] = v
            elif k == "physical id":
                pid = v
                phys.add(v)
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                cores.add((pid, cid))
                pid = cid = None
        info["sockets"] = len(phys) or None
        info["physical_cores"] = len(cores) or None
    except OSError:
        pass
    return info


def cpu_baseline(workload):
    """oracle/torch_port.py (kind "port-torch": the ATen CPU ops the reference calls, in its order, encoder included)
    in a subprocess with glibc malloc tuned (see torch_port.MALLOC_ENV), on a bounded sample of the workload:
    same video size / window length / weights, fewer frames and points so that it finishes in ~30-60 s."""
    import subprocess
    from oracle.torch_port import MALLOC_ENV
    H, W, T, G, offline, wl, _ = WORKLOADS[workload]
    if offline is True:
        kind, frames, grid = "offline", min(T, 48), 20
    else:  # sliding windows (c3 / c4 / c5): 3 windows of 16 frames, 1600 points (a quarter of the headline's per-window rows)
        kind, frames, grid = "sliding", 32, 40
    env = dict(os.environ, **MALLOC_ENV)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    # intra-op threads: ATen's CPU kernels stop scaling well before a 128-core host is full on tensors of this size
    # (measured on the MI355X node, T=32: 128 threads 227 pf/s); 32 threads is the stated configuration
    threads = min(32, os.cpu_count() or 1)
    cmd = [sys.executable, "-m", "oracle.torch_port", "--bench", kind, "--frames", str(frames), "--grid", str(grid),
           "--size", str(H), "--threads", str(threads)]
    host = cpu_model()
    try:
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:  # the baseline is a reported number, never a reason to lose the bench line
        return {"value": None, "unit": "tracked-point-frames/s", "cores": os.cpu_count(), "kind": "port-torch", "host_cpu": host,
                "sample": f"failed: {type(e).__name__}: {e}"}
    res = {"value": r["tracked_point_frames_per_s"], "unit": "tracked-point-frames/s", "cores": r["threads"],
           "kind": "port-torch", "host_cpu": host, "host_logical_cpus": os.cpu_count(), "seconds": r["seconds"],
           # the sample is the bench workload's video size / window length / weights at FEWER frames and points (a ~30-60 s CPU
           # budget); it is NOT the headline configuration and the value is NOT rescaled to it ("rescaled_to_config": false)
           "sample_is_config": False, "rescaled_to_config": False, "sample_frames": r["frames"], "sample_points": r["points"],
           "sample": f"oracle/torch_port.py predictor path incl. encoder, {kind}, {r['video'][0]}x{r['video'][1]} video, "
                     f"T={r['frames']}, N={r['points']} (grid {grid}), 6 iterations, {r['threads']} threads of "
                     f"{host['model']} ({host['sockets']} sockets, {host['physical_cores']} cores, {host['logical_cpus']} logical), glibc malloc "
                     f"tuned ({r['malloc_tuned']}): {r['seconds']} s; value = N*T/s of that sample (not rescaled)"}
    # what the UNMODIFIED reference did in the build container on the BASELINE-scale goldens (recorded by
    # tests/golden/make_golden_scale.py): same metric, different host
    try:
        import numpy as np
        ref = {}
        for name in ("c1", "c2", "c4", "c3_g40", "c3_g80", "c3_off", "c5_chunk0"):
            f = os.path.join(ROOT, "tests", "golden", f"scale_{name}.npz")
            if os.path.exists(f):
                g = np.load(f)
                n, t = int(g["n_points_total"]) if "n_points_total" in g else g["coords"].shape[1], g["coords"].shape[0]
                # two runs per golden (different intra-op thread counts); report the faster: the slower one may have
                # shared the container with other jobs
                runs = [(float(g["seconds"]), int(g["threads"])), (float(g["noise_seconds"]), int(g["noise_threads"]))]
                sec, thr = min(runs)
                ref[name] = {"tracked_point_frames_per_s": round(n * t / sec, 1), "threads": thr, "host_cpus": int(g["host_cpus"]),
                             "seconds": round(sec, 1), "points": n, "frames": t}
        res["reference_in_build_container"] = ref
        # First-class: the UNMODIFIED reference on EXACTLY this workload (same video, weights, N, T -- the run that produced the
        # golden the timed step is checked against).  It is the number SURVEY 8d's protocol asks for, but measured on the build
        # container's few vCPUs, not on this node's host: so both are reported and neither is silently "the" baseline --
        # `value` above = the port on THIS node's cores on a bounded sample (comparable host, smaller job),
        # `reference_exact_config` = the reference itself on the exact job (other host).  vs_baseline stays null: BASELINE.md
        # publishes no number for this metric; if one were formed it would use `reference_exact_config`.
        gname = {"c3_sliding": "c3_g80", "c2_offline": "c2", "c1_standin": "c1", "c3_offline_g40": "c3_off", "c4_online": "c4",
                 "c5_shard": "c5_chunk0"}.get(workload)
        if gname in ref:
            res["reference_exact_config"] = dict(ref[gname], kind="reference", golden=f"tests/golden/scale_{gname}.npz",
                                                 host="build container (no GPU), intra-op threads as stated",
                                                 note="unmodified facebookresearch/co-tracker CoTrackerPredictor / model forward, CPU fp32")
        res["which_is_the_baseline"] = ("value = oracle/torch_port.py on this node's host cores, bounded sample (kind port-torch); "
                                        "reference_exact_config = the unmodified reference on the exact configuration, build-container "
                                        "CPUs; vs_baseline would be formed against reference_exact_config, and is null because "
                                        "BASELINE.md publishes no throughput")
    except Exception:
        pass
    return res


def gemm_clock_probe(dev):
    """The shader clock the persistent Linear kernels actually run at (round 5).  Workgroup 0 / wave 0 of every persistent GEMM
    launch stamps (s_memtime, s_memrealtime) at its start and end (gemm_pp.hip: g_pp_clock; s_memrealtime is the constant 100 MHz
    counter): cycles / time of the LAST of 8 back-to-back launches of a C3-window Linear = the clock under the power cap, and
    mfma_issue_cycles / cycles = how much of the workgroup's life its SIMDs spent issuing MFMAs (12 per phase x 32 cycles, two
    wave groups per SIMD).  The clock, not stalls, is what separates these kernels from the nominal roofline: they are power-capped
    (profiles/r06_feed_lab.txt: the same loop runs at 2.4 GHz on zero operands and at 1.8 GHz on random ones).  DEV builds only."""
    import ctypes as C
    from cotracker_amd import _lib as L
    from cotracker_amd import ops
    lib = L.load()
    if not hasattr(lib, "ctk_debug_pp_clock"):
        # the release library carries no instruments (round 6): the clock readings need `make dev` + CTK_LIB_PATH=libctk_hip_dev.so
        return {"unavailable": "release build of libctk_hip.so has no ctk_debug_pp_clock; round-6 readings of the same kind "
                               "(zero vs random operands, clock per variant): profiles/r06_feed_lab.txt"}
    fn = lib.ctk_debug_pp_clock
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    out = {}
    g = torch.Generator(device="cpu").manual_seed(5)
    for name, M, K, N, act, res, split in (("mlp.fc1", 103424, 384, 1536, 2, False, True), ("mlp.fc2", 103424, 1536, 384, 0, True, False),
                                           ("to_q", 103424, 384, 384, 0, False, False), ("to_kv", 102400, 384, 768, 0, False, False)):
        a = ops.split_rows(torch.randn(4096, K, generator=g).to(dev).repeat(M // 4096 + 1, 1)[:M].contiguous())
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        wp = ops.pack_weight(w)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, device=dev) if res else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(9):
            if i == 1:
                e0.record()
            if res:
                ops.gemm(a, w, bias=b, act=act, resid=r, out=r, packed=wp)
            else:
                ops.gemm(a, w, bias=b, act=act, packed=wp, out_split=split)
        e1.record()
        e1.synchronize()
        c = (C.c_ulonglong * 4)()
        L.check(fn(c), "ctk_debug_pp_clock")
        cycles, real = c[2] - c[0], c[3] - c[1]
        t256 = N % 256 == 0
        tiles = -(-M // 256) * (N // (256 if t256 else 192))
        if not t256 or True:
            rem = tiles % 256
            if 0 < rem <= 64 and tiles > 256:
                tiles -= rem  # the tail split leaves whole rounds to the persistent kernel
        per_wg = -(-tiles // 256)
        mfma_cycles = per_wg * (K // 32) * (4 if t256 else 3) * 2 * 12 * 32
        out[name] = {"us_per_launch": round(e0.elapsed_time(e1) * 1e3 / 8, 1), "wg0_cycles": int(cycles), "wg0_us": round(real / 100.0, 1),
                     "clock_ghz": round(cycles / real * 0.1, 3) if real else None,
                     "mfma_issue_share_of_cycles": round(mfma_cycles / cycles, 3) if cycles else None}
        del a, w, wp, b, r
    torch.cuda.empty_cache()
    return out


def _traffic_fields(out, tr):
    if tr:
        out["traffic"] = tr.get("hbm_bytes_per_launch")
        out["traffic_detail"] = {k: tr[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "dispatches", "source") if k in tr}
        out["traffic_note"] = ("HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this workload "
                               "(tools/pmc_traffic.py -> profiles/pmc_traffic.json, a committed file; not re-measured in this run)")
    else:
        out["traffic"] = None
    return out


SPLIT_NOTE = ("achieved = algorithmic (f32-equivalent) flops / HIP-event time; every product is 3 v_mfma_f32_32x32x16_f16 "
              "(hi*hi + hi*lo + lo*hi, f32 accumulate), so peak = dense f16 MFMA 2500 TF/s / 3; frac is also mfma_issued / "
              "mfma_peak.  The exact-f32 MFMA peak is 157.3 TF/s")


# Recorder rows whose contraction runs on the EXACT-f32 MFMA (v_mfma_f32_32x32x2_f32: gemm.hip, corr.hip's corr_volume) or on
# the VALU (corrblock_sample).  EVERY other row with MFMA flops -- the split-half GEMMs (gemm_sh*, gemm_f16x3*), the
# implicit-GEMM convolutions of the encoder (conv_pp128_*), the split-half sampler, the attention kernels -- issues 3
# v_mfma_f32_32x32x16_f16 per product and is priced against 2500/3 TF/s.  (Round 3 keyed this on a list of split-half name
# prefixes, which priced conv_pp128_* against 157 TF/s: frac 0.93 instead of 0.18.)
EXACT_F32_ROWS = ("gemm_f32", "corrblock_sample")


def is_exact_f32(name):
    return name.startswith(EXACT_F32_ROWS) or name == "corr_volume"


def mfma_peak(name):
    return FP32_MFMA_PEAK_TFLOPS if is_exact_f32(name) else F16_MFMA_PEAK_TFLOPS / 3.0


def roofline_mfma(name, rows, traffic, sustained=None):
    """MFMA roofline of one recorder row, or of several rows together (call-weighted: summed flops / summed time).
    Split-half kernels form every f32-class product from 3 f16 MFMAs -> ceiling 2500/3 = 833 TF/s algorithmic.
    `sustained` (bench.sustained_mfma): the f16 MFMA rate this chip holds under its power cap -> frac_at_sustained_clock."""
    sec = sum(r["total_ms"] for r in rows) * 1e-3
    n = max(sum(r["launches"] for r in rows), 1)
    flops = sum(r["flops"] for r in rows)
    ach = flops / sec / 1e12
    exact = [is_exact_f32(r["name"]) for r in rows]
    assert all(exact) or not any(exact), "a call-weighted row set must not mix exact-f32 and split-half kernels"
    split = not exact[0]
    peak = F16_MFMA_PEAK_TFLOPS / 3.0 if split else FP32_MFMA_PEAK_TFLOPS
    out = {"kernel": name, "launches": n, "avg_launch_us": round(1e6 * sec / n, 1), "bound": "mfma", "achieved": round(ach, 2),
           "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "flops_per_launch": flops / n}
    if split:
        out.update({"mfma_issued": round(3 * ach, 1), "mfma_peak": F16_MFMA_PEAK_TFLOPS, "f32_mfma_peak": FP32_MFMA_PEAK_TFLOPS,
                    "note": SPLIT_NOTE})
        if sustained and sustained.get("f16_tflops"):
            out["frac_at_sustained_clock"] = round(3 * ach / sustained["f16_tflops"], 4)
            out["sustained_mfma_clock_ghz"] = sustained["clock_ghz"]
    else:
        out["note"] = "exact-f32 MFMA (v_mfma_f32_32x32x2_f32) peak"
    if len(rows) == 1:
        return _traffic_fields(out, traffic.get(name))
    def tr_of(n):  # the PMC file is keyed by KERNEL (tools/pmc_traffic.py): every 64x64-tile shape shares the "gemm_sh_64x64" entry
        return traffic.get(n) or (traffic.get("gemm_sh_64x64") if n.startswith("gemm_sh_64_") else None)

    trs = [tr_of(r["name"]) for r in rows]
    if all(t and t.get("hbm_bytes_per_launch") for t in trs):  # call-weighted HBM bytes per launch of the class
        tot = sum(t["hbm_bytes_per_launch"] * r["launches"] for t, r in zip(trs, rows))
        agg = {"hbm_bytes_per_launch": tot / n, "source": trs[0].get("source"),
               "fetch_bytes_per_launch": sum(t.get("fetch_bytes_per_launch", 0.0) * r["launches"] for t, r in zip(trs, rows)) / n,
               "write_bytes_per_launch": sum(t.get("write_bytes_per_launch", 0.0) * r["launches"] for t, r in zip(trs, rows)) / n,
               "dispatches": sum(t.get("dispatches", 0) for t in trs)}
        return _traffic_fields(out, agg)
    return _traffic_fields(out, None)


def sustained_mfma(dev, seconds=1.0):
    """What the f16 MFMA pipe of THIS chip sustains under its power cap: ctk_probe_mfma kind 2 (register-only loop of
    v_mfma_f32_32x32x16_f16 on pseudo-random operands, 2 workgroups x 4 waves per CU) run for ~`seconds`, timed with HIP events
    on the launch stream.  clock = rate / (1024 SIMDs x 1024 flop per clock and SIMD); the nominal 2500 TF/s is 2.4 GHz."""
    import ctypes as C
    from cotracker_amd import _lib as L
    lib = L.load()
    scratch = torch.zeros(16, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run(iters):
        fl = C.c_double(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.ctk_probe_mfma(2, iters, scratch.data_ptr(), C.byref(fl), stream), "ctk_probe_mfma")
        e1.record()
        e1.synchronize()
        return fl.value, e0.elapsed_time(e1) * 1e-3

    fl, t = run(20000)  # calibrate (~15 ms), then one long launch so that the clock has settled at the cap
    iters = int(min(max(20000 * seconds / max(t, 1e-4), 20000), 4e6))
    fl, t = run(iters)
    fl2, t2 = run(max(iters // 4, 20000))  # the rate right after a second of full load (clock already down)
    tf = fl / t / 1e12
    return {"f16_tflops": round(tf, 1), "clock_ghz": round(tf * 1e12 / (1024.0 * 1024.0) / 1e9, 3), "seconds": round(t, 3),
            "f16_tflops_after": round(fl2 / t2 / 1e12, 1), "nominal_f16_tflops": F16_MFMA_PEAK_TFLOPS, "nominal_clock_ghz": 2.4,
            "note": "ctk_probe_mfma kind 2: register-only v_mfma_f32_32x32x16_f16 loop on pseudo-random operands, all 256 CUs, "
                    "measured in this run right after the timed steps; frac_at_sustained_clock = issued f16 MFMA rate / this"}


def roofline_sampler(row, traffic):
    """The correlation sampler against ITS roofline: the larger of (measured HBM bytes / 8 TB/s) and (contraction flops /
    split-half MFMA ceiling).  frac = that bound / the measured launch time.  SURVEY 8d's no-reuse algorithmic bytes
    (every point re-reads its own 8x8x128 footprint) are kept as a secondary field: neighbouring grid points share
    footprint pixels in L2 / Infinity Cache, so that figure is NOT what reaches HBM."""
    name = row["name"]
    sec = row["total_ms"] * 1e-3
    n = max(row["launches"], 1)
    t_launch = sec / n
    tr = traffic.get(name)
    peak_tf = F16_MFMA_PEAK_TFLOPS / 3.0
    t_mfma = row["flops"] / n / (peak_tf * 1e12)
    out = {"kernel": name, "launches": row["launches"], "avg_launch_us": round(1e6 * t_launch, 1),
           "flops_per_launch": row["flops"] / n, "mfma_TFLOPs": round(row["flops"] / sec / 1e12, 2),
           "mfma_frac": round(row["flops"] / sec / 1e12 / peak_tf, 4),
           "algorithmic_bytes_per_launch_no_reuse": row["bytes"] / n,
           "algorithmic_GBs_no_reuse": round(row["bytes"] / sec / 1e9, 1)}
    if tr and tr.get("hbm_bytes_per_launch"):
        hbm = tr["hbm_bytes_per_launch"]
        t_hbm = hbm / (HBM_PEAK_GBS * 1e9)
        bound = "hbm" if t_hbm >= t_mfma else "mfma"
        t_bound = max(t_hbm, t_mfma)
        out.update({"bound": bound, "achieved": round(hbm / t_launch / 1e9, 1) if bound == "hbm" else out["mfma_TFLOPs"],
                    "peak": HBM_PEAK_GBS if bound == "hbm" else round(peak_tf, 1), "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                    "frac": round(t_bound / t_launch, 4), "hbm_GBs_measured": round(hbm / t_launch / 1e9, 1),
                    "hbm_frac": round(hbm / t_launch / 1e9 / HBM_PEAK_GBS, 4), "roofline_bound_us": round(1e6 * t_bound, 1),
                    "frac_measured": round(t_bound / t_launch, 4),
                    "frac_no_reuse": round(row["bytes"] / sec / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "frac = frac_measured = max(measured HBM bytes / 8 TB/s, flops / 833 TF/s) / measured launch time; "
                            "frac_no_reuse = SURVEY 8(d)'s algorithmic bytes (every (t,n,level) unit re-reads its own footprint) / 8 TB/s / "
                            "launch time -- NOT credit: neighbouring points share footprint pixels in L2 / Infinity Cache"})
    else:  # no PMC pass for this kernel on this workload: only the MFMA side is known
        out.update({"bound": "mfma", "achieved": out["mfma_TFLOPs"], "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                    "frac": out["mfma_frac"], "note": "no PMC traffic for this kernel/workload: MFMA side only; HBM traffic unmeasured"})
    return _traffic_fields(out, tr)


GEMM_ROW_PREFIXES = ("gemm_sh", "gemm_f16x3", "gemm_f32", "mlp_sh")


def rooflines(rows, traffic, sustained=None):
    """The roofline objects of one profiled step.  `roofline` is the TIME-DOMINANT CLASS of kernels, not the largest single
    recorder row (rows are per GEMM shape, so the one sampler row used to out-rank 839 ms of GEMM launches): the Linear
    GEMMs call-weighted, or the correlation sampler, whichever owns more of the step.  Both are always present under their
    own keys (`roofline_gemm`, `roofline_sampler`)."""
    out = {}
    gemm_rows = [r for r in rows if r["name"].startswith(GEMM_ROW_PREFIXES) and r["total_ms"] > 0]
    samp_rows = [r for r in rows if r["name"].startswith("corr_") and r["total_ms"] > 0]
    if gemm_rows:
        exact = is_exact_f32(gemm_rows[0]["name"])
        same = [r for r in gemm_rows if is_exact_f32(r["name"]) == exact]  # (one back end per run; be safe)
        g = roofline_mfma("gemm (all Linear launches, call-weighted)", same, traffic, sustained)
        g["total_ms"] = round(sum(r["total_ms"] for r in same), 2)
        g["rows"] = [{"name": r["name"], "launches": r["launches"], "avg_us": round(1e3 * r["total_ms"] / max(r["launches"], 1), 1),
                      "frac": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12 / mfma_peak(r["name"]), 4)} for r in same]
        out["roofline_gemm"] = g
    if samp_rows:
        sr = roofline_sampler(samp_rows[0], traffic)
        sr["total_ms"] = round(samp_rows[0]["total_ms"], 2)
        out["roofline_sampler"] = sr
    t_gemm = sum(r["total_ms"] for r in gemm_rows)
    t_samp = samp_rows[0]["total_ms"] if samp_rows else 0.0
    others = [r for r in rows if r not in gemm_rows and r not in samp_rows and r["total_ms"] > 0]
    top_other = max(others, key=lambda r: r["total_ms"]) if others else None
    if gemm_rows and t_gemm >= t_samp and (top_other is None or t_gemm >= top_other["total_ms"]):
        out["roofline"] = dict(out["roofline_gemm"], dominant_class="gemm", class_share_of_kernel_time=round(
            t_gemm / max(sum(r["total_ms"] for r in rows), 1e-9), 4))
    elif samp_rows and (top_other is None or t_samp >= top_other["total_ms"]):
        out["roofline"] = dict(out["roofline_sampler"], dominant_class="sampler", class_share_of_kernel_time=round(
            t_samp / max(sum(r["total_ms"] for r in rows), 1e-9), 4))
    elif top_other is not None:  # e.g. the encoder's convolutions on a tiny-N workload
        out["roofline"] = dict(roofline_mfma(top_other["name"], [top_other], traffic, sustained), dominant_class="other")
    return out


def golden_parity(name, coords, vis_logit, conf_logit, coords_key="coords"):
    """Max-abs error against the unmodified reference's CPU run (tests/golden/scale_<name>.npz): coords_key "coords" =
    model.forward()[0] in model-resolution px, "tracks" = the predictor's output in raw-video px (after the query-frame
    overwrite); logits are the pre-sigmoid visibility / confidence."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", f"scale_{name}.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    sel = g["point_index"] if "point_index" in g else None  # scale_c3_g80 stores every 4th of the jointly tracked points

    def err(a, b):
        a = a.detach().double().cpu().numpy()
        return float(np.abs((a[:, sel] if sel is not None else a) - b.astype(np.float64)).max())

    return {"against": f"unmodified reference, CPU fp32, {str(g['meta'])} (tests/golden/scale_{name}.npz)",
            "coords_px": err(coords, g[coords_key if coords_key in g else "coords"]), "vis_logit": err(vis_logit, g["vis_logit"]),
            "conf_logit": err(conf_logit, g["conf_logit"]),
            "reference_own_noise": {"threads": [int(g["threads"]), int(g["noise_threads"])],
                                    "coords_px": float(g["noise_coords_max"]), "vis_logit": float(g["noise_vis_logit_max"]),
                                    "conf_logit": float(g["noise_conf_logit_max"])}}


def c2_parity(dev, precision):
    """BASELINE configs[1] end to end (predictor, encoder included) against the reference golden: 68 ms on the GPU."""
    from cotracker_amd import model as M
    from cotracker_amd.predictor import CoTrackerPredictor
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_
    old, M.DEFAULT_PRECISION = M.DEFAULT_PRECISION, precision
    try:
        p = CoTrackerPredictor(checkpoint=None, offline=True, window_len=60)
    finally:
        M.DEFAULT_PRECISION = old
    fill_synthetic_(p.model, seed=0)
    p = p.to(dev)
    cap = {}
    fwd = p.model.forward

    def tap(*a, **k):
        out = fwd(*a, **k)
        cap["coords"] = out[0].clone()  # model-level tracks, before the predictor overwrites the query-frame rows in place
        return out

    p.model.forward = tap
    p(synthetic_video(48, 256, 256, seed=1234).to(dev), grid_size=20)
    vl, cl = p.model.last_logits
    return golden_parity("c2", cap["coords"][0], vl[0], cl[0])


def quick_line(name, dev, precision="f16x3", steps=2, warmup=1, feature_cache=False):
    """One more workload timed in the same process after the headline (single GPU, no profile, no CPU leg): the same
    barrier-free protocol -- `warmup` untimed steps, synchronize, `steps` timed steps, synchronize.  Used for the extra rows
    of the driver-run JSON (`extra_lines`): the exact-f32 back end on the headline workload, BASELINE configs[1] (C2) and
    configs[3] (C4, default = the reference's behaviour of re-encoding every chunk; the feature cache is a labelled opt-in)."""
    from cotracker_amd import model as M
    from cotracker_amd.predictor import CoTrackerOnlinePredictor, CoTrackerPredictor
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_
    H, W, T, G, offline, wl, desc = WORKLOADS[name]
    old, M.DEFAULT_PRECISION = M.DEFAULT_PRECISION, precision
    try:
        if name == "c4_online":
            pred = CoTrackerOnlinePredictor(checkpoint=None, window_len=wl)
            pred.model.hip_graph = True
            pred.model.online_feature_cache = bool(feature_cache)
            T = pred.step * (steps + warmup + 2)
        else:
            pred = CoTrackerPredictor(checkpoint=None, offline=offline, window_len=wl)
    finally:
        M.DEFAULT_PRECISION = old
    fill_synthetic_(pred.model, seed=0)
    pred = pred.to(dev)
    video = synthetic_video(T, H, W, seed=1234).to(dev)
    if name == "c4_online":
        pred(video_chunk=video[:, :2 * pred.step], is_first_step=True, grid_size=G)
        cur = [0]

        def step():
            i = cur[0]
            cur[0] += pred.step
            return pred(video_chunk=video[:, i:i + 2 * pred.step])
        frames = pred.step
    else:
        def step():
            return pred(video, grid_size=G)
        frames = T
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / steps
    assert torch.isfinite(out[0]).all()
    finish = getattr(pred, "finish", None)
    if finish is not None:
        finish()  # resolve the deferred f16-range check of the last graph replay
    line = {"workload": name, "precision": precision, "steps": steps, "warmup": warmup, "ms_per_step": round(sec * 1e3, 2),
            "value": round(G * G * frames / sec, 1), "unit": "tracked-point-frames/s", "range_fallbacks": int(getattr(pred.model, "range_fallbacks", 0))}
    if name == "c4_online":
        line["online_feature_cache"] = bool(feature_cache)
        line["hip_graph"] = True
    del pred, video
    torch.cuda.empty_cache()
    return line


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
        # plain `python bench.py --gpus N`: become the launcher (one rank per GPU over RCCL, rendezvous on 127.0.0.1)
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    if args.single_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        # one rank per GPU is the contract; two ranks on one device would report an "N-GPU" number measured on fewer GPUs
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} device(s) visible: pass --single-device (with --dist-backend gloo) "
                         "to exercise the multi-rank path on one GPU on purpose")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank_devices = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        # which physical device every rank sits on (reported in the JSON line; a shared device without --single-device is an error)
        prop = torch.cuda.get_device_properties(dev)
        ident = str(getattr(prop, "uuid", "")) or str(getattr(prop, "pci_bus_id", local_rank))
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "device": ident}
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, mine)
        if not args.single_device and len({d["device"] + ":" + str(d["device_index"]) for d in rank_devices}) < world:
            raise SystemExit(f"two ranks share a device: {rank_devices}")

    from cotracker_amd import model as ctk_model
    from cotracker_amd import ops
    from cotracker_amd.predictor import CoTrackerPredictor, get_points_on_a_grid
    ctk_model.DEFAULT_PRECISION = args.precision
    from cotracker_amd.sharding import all_gather_tracks, chunk_bounds, track_sharded
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_

    H, W, T, G, offline, wl, desc = WORKLOADS[args.workload]
    N = G * G
    streaming = args.workload == "c4_online"
    c5 = args.workload == "c5_shard"
    if streaming:
        from cotracker_amd.predictor import CoTrackerOnlinePredictor
        pred = CoTrackerOnlinePredictor(checkpoint=None, window_len=wl)
        pred.model.hip_graph = not args.no_graph
        # the workload's chunks overlap by window_len - step frames (predictor.py:288-290); re-using their features is --feature-cache
        pred.model.online_feature_cache = bool(args.feature_cache)  # default off = the reference's behaviour (every chunk re-encoded)
        T = pred.step * (args.steps + args.warmup + 3)  # one chunk call per step, plus the profiled call
    elif offline == "v2":
        pred = CoTrackerPredictor(checkpoint=None, v2=True, window_len=wl)
    else:
        pred = CoTrackerPredictor(checkpoint=None, offline=offline, window_len=wl)
    fill_synthetic_(pred.model, seed=0)
    pred = pred.to(dev)
    video = synthetic_video(T, H, W, seed=1234).to(dev)  # resident in HBM before timing starts
    ih, iw = pred.interp_shape
    to_raw = torch.tensor([(W - 1) / (iw - 1), (H - 1) / (ih - 1)], device=dev)  # model-resolution px -> raw-video px
    sharding = f"points x{world}"
    n_per_rank = N

    if streaming:
        # CoTrackerOnlinePredictor protocol (predictor.py:228-300): first call registers the grid queries, then every
        # call consumes the last 2*step frames; one bench step = one such call = `step` new frames for N points.
        pred(video_chunk=video[:, :2 * pred.step], is_first_step=True, grid_size=G)
        cursor = [0]
        if world > 1:
            raise SystemExit("c4_online is a single-GPU latency workload")

        def step():
            i = cursor[0]
            cursor[0] += pred.step
            return pred(video_chunk=video[:, i:i + 2 * pred.step])
    elif c5:
        # the joint 265x265 grid, split into 8 contiguous chunks in row-major grid order; rank r tracks
        # chunk r as ONE model call (the virtual tracks couple the points of a call: a chunk is a call, predictor.py:80-96)
        if world > 8:
            raise SystemExit("c5_shard defines 8 chunks")
        # built on the CPU with the arithmetic of tests/golden/make_golden_scale.py (c5_chunk0), then uploaded: the parity of
        # the timed step needs bit-identical queries
        pts = get_points_on_a_grid(G, (ih, iw), device=torch.device("cpu")) * to_raw.cpu()
        q_all = torch.cat([torch.zeros_like(pts[:, :, :1]), pts], dim=2)
        lo, hi = chunk_bounds(N, 8, rank)
        q = q_all[:, lo:hi].contiguous().to(dev)
        n_per_rank = hi - lo
        sharding = f"chunk {rank} of 8 per rank ({world} of 8 chunks tracked: {'the whole job' if world == 8 else 'per-GPU rate of the 8-GPU job'})"
        n_gather = sum(chunk_bounds(N, 8, r)[1] - chunk_bounds(N, 8, r)[0] for r in range(world))

        model_fwd = pred.model.forward
        tap = {}

        def tapped(*a, **k):
            o = model_fwd(*a, **k)
            tap["coords"] = o[0].clone()
            return o

        if rank == 0:  # rank 0 tracks chunk 0 whatever the world size: its timed step is checked against scale_c5_chunk0
            pred.model.forward = tapped

        def local_step():
            return pred(video, queries=q)

        def step():
            tr, vi = local_step()
            return all_gather_tracks(tr, vi, n_gather) if world > 1 else (tr, vi)
    elif world == 1:
        model_fwd = pred.model.forward
        tap = {}

        def tapped(*a, **k):  # model-level tracks before the predictor overwrites the query-frame rows in place (parity check)
            o = model_fwd(*a, **k)
            tap["coords"] = o[0].clone()
            return o

        pred.model.forward = tapped

        def step():
            return pred(video, grid_size=G)
    else:
        # weak scaling of the BASELINE configs[2] job: the query list is `world` G x G grids (each shifted by a sub-pixel
        # offset: a denser joint grid of world*N points), split into contiguous chunks by sharding.chunk_bounds; every
        # rank tracks its chunk as one call and ONE all-gather (RCCL) returns the full tracks to every rank.
        base = get_points_on_a_grid(G, (ih, iw), device=dev)
        pts = torch.cat([base + torch.tensor([0.37, 0.23], device=dev) * r for r in range(world)], dim=1) * to_raw
        q_all = torch.cat([torch.zeros_like(pts[:, :, :1]), pts], dim=2)
        sharding = f"{world * N} queries in {world} contiguous chunks of {N} (sharding.track_sharded), one all-gather"

        lo_r, hi_r = chunk_bounds(world * N, world, rank)

        def local_step():  # this rank's chunk alone (what the profiled extra step runs: no collective outside the timed steps)
            return pred(video, queries=q_all[:, lo_r:hi_r])

        def step():
            return track_sharded(pred, video, q_all)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # Every timed step of a non-streaming workload sees the same video and queries, and the library claims bit-reproducible
    # results: keep REFERENCES to every step's outputs (fresh tensors per call: no copy, no host wait inside the timed region)
    # and compare them bit for bit after the closing synchronisation (`steps_bit_identical` in the line).  One C3 step is
    # 84 sampler launches x 25 600 workgroups + ~8 600 other launches, so this is a free production-shape soak of every kernel.
    kept = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        if not streaming:
            kept.append((out[0], out[1], getattr(pred.model, "last_logits", None)))
    sync()
    dt = time.perf_counter() - t0
    steps_bit_identical = None
    if kept:
        def same(a, b):
            if a is None or b is None:
                return a is b
            if isinstance(a, (tuple, list)):
                return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
            return a.shape == b.shape and bool(torch.equal(a, b))
        bad = [i for i in range(1, len(kept)) if not same(kept[0], kept[i])]
        steps_bit_identical = {"identical": not bad, "steps_compared": len(kept), "differing_steps": bad[:16],
                               "what": "pred_tracks, pred_visibility and the model-level visibility / confidence logits of every "
                                       "timed step against the first timed step, torch.equal"}
        if bad:
            print(f"bench.py: timed steps {bad[:16]} differ bitwise from the first timed step", file=sys.stderr)
    kept = None
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    sec_per_step = float(tmax.item()) / args.steps
    assert torch.isfinite(out[0]).all()
    assert pred.model.range_fallbacks == 0 if hasattr(pred.model, "range_fallbacks") else True
    # the one collective of the path, timed alone (every rank takes part; the profiled extra step below is rank 0 only)
    all_gather_ms = None
    if world > 1:
        tr_l, vi_l = local_step()
        n_tot = n_gather if c5 else world * N
        sync()
        t_ag = time.perf_counter()
        for _ in range(5):
            all_gather_tracks(tr_l, vi_l, n_tot)
        sync()
        all_gather_ms = (time.perf_counter() - t_ag) / 5 * 1e3
    frames_per_step = pred.step if streaming else T
    if c5:
        total_points = sum(chunk_bounds(N, 8, r)[1] - chunk_bounds(N, 8, r)[0] for r in range(world))
    else:
        total_points = world * N
    value = total_points * frames_per_step / sec_per_step

    result = {
        "metric": "tracked-point-frames/sec (N*T/s)", "value": round(value, 1), "unit": "tracked-point-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (Linear layers as split-half f16 MFMA x3, f32 accumulate)" if args.precision == "f16x3" else "f32",
        "data": "synthetic",
        "rccl_ranks": dist.get_world_size() if world > 1 else 1, "dist_backend": args.dist_backend if world > 1 else None,
        "rank_devices": rank_devices, "single_device": bool(args.single_device) if world > 1 else None,
        "all_gather_ms": None if all_gather_ms is None else round(all_gather_ms, 3),
        # the path's ONE collective must stay noise beside a step (no collective on the data path): self-checking the
        # day a multi-GPU node runs this line
        "all_gather_share_of_step": None if all_gather_ms is None else round(all_gather_ms / (sec_per_step * 1e3), 5),
        "all_gather_ok": None if all_gather_ms is None else bool(all_gather_ms < 0.01 * sec_per_step * 1e3),
        "steps_bit_identical": steps_bit_identical,
        "config": {"workload": desc, "name": args.workload, "points_per_gpu": n_per_rank, "frames": frames_per_step, "video": [H, W],
                   "iters": 6, "window_len": wl, "offline": bool(offline) and offline != "v2", "sharding": sharding,
                   "precision": args.precision, "weights": "seeded synthetic (no checkpoints offline)"},
    }
    if streaming:
        result["config"]["hip_graph"] = bool(pred.model.hip_graph)
        result["config"]["online_feature_cache"] = bool(pred.model.online_feature_cache)
        result["config"]["graph_nodes"] = next(iter(pred.model._graphs.values())).nodes if pred.model._graphs else 0

    # parity of what was just timed: the last timed step's model-level outputs against the reference's CPU run of the
    # same workload (world == 1, c3_sliding / c2_offline have goldens at exactly these sizes), plus C2 end to end
    if rank == 0:
        parity = {}
        golden_name = {"c3_sliding": "c3_g80", "c2_offline": "c2", "c1_standin": "c1", "c3_offline_g40": "c3_off",
                       "c5_shard": "c5_chunk0"}.get(args.workload)
        if (world == 1 or c5) and golden_name and getattr(pred.model, "last_logits", None) is not None:
            vl, cl = pred.model.last_logits
            gp = golden_parity(golden_name, tap["coords"][0], vl[0], cl[0], coords_key="coords")
            if gp:
                gp["note"] = ("outputs of the LAST TIMED step: model-level tracks (model-resolution px, before the predictor's "
                              "query-frame overwrite) and pre-sigmoid logits")
                parity["timed_step"] = gp
        if args.workload != "c2_offline":
            try:
                parity["c2"] = c2_parity(dev, args.precision)
            except Exception as e:
                parity["c2"] = {"error": f"{type(e).__name__}: {e}"}
        result["parity"] = parity

    if rank == 0 and not args.no_profile:
        # one extra step with the library's HIP-event recorder on (events on the launch stream)
        if streaming:
            pred.model.hip_graph = False  # events cannot be recorded inside a captured graph: profile the direct launches
        ops.profile_enable(True)
        t1 = time.perf_counter()
        (local_step if world > 1 else step)()  # rank 0 only: must not enter a collective the other ranks do not join
        torch.cuda.synchronize()
        prof_step_s = time.perf_counter() - t1
        rows = ops.profile_read()
        ops.profile_enable(False)
        rows.sort(key=lambda r: -r["total_ms"])
        kern = []
        for r in rows:
            avg_us = 1e3 * r["total_ms"] / max(r["launches"], 1)
            kern.append({"name": r["name"], "launches": r["launches"], "total_ms": round(r["total_ms"], 2),
                         "avg_us": round(avg_us, 1),
                         "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2) if r["total_ms"] > 0 else 0.0,
                         "algorithmic_GBs": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1) if r["total_ms"] > 0 else 0.0})
        result["kernels"] = kern
        result["profiled_step_ms"] = round(prof_step_s * 1e3, 1)
        result["hip_kernels_ms"] = round(sum(r["total_ms"] for r in rows), 1)
        result["profile_note"] = ("kernels / roofline come from ONE extra step run after the timed ones with a HIP-event pair around "
                                  "every launch (events serialise the stream and add ~5 us per launch, so hip_kernels_ms + encoder "
                                  "can exceed ms_per_step); rocprofv3 --kernel-trace --stats of the same command is in profiles/")
        traffic = {}
        if os.path.exists(args.pmc_traffic):
            traffic = json.load(open(args.pmc_traffic))
            import hashlib
            so = os.path.join(ROOT, "co-tracker_amd", "libctk_hip.so")
            so_hash = hashlib.sha256(open(so, "rb").read()).hexdigest()
            stamp = traffic.pop("_lib_sha256", None)
            src_stamp = traffic.pop("_src_sha256", None)
            import __graft_entry__ as ge
            src_hash = ge.source_hash()
            # fresh = collected on THIS build of the kernels: same kernel sources (box-independent) or the same library bytes
            fresh = (src_stamp is not None and src_stamp == src_hash) or stamp == so_hash
            result["traffic_source"] = {"file": os.path.relpath(args.pmc_traffic, ROOT), "src_sha256": src_stamp, "this_src_sha256": src_hash,
                                        "lib_sha256": stamp, "this_lib_sha256": so_hash, "fresh": fresh}
            if traffic.pop("_workload", "c3_sliding") != args.workload or not fresh:
                traffic = {}  # the PMC passes were collected on another workload or another build of the kernels: bytes do not transfer
        sustained = None
        try:
            sustained = sustained_mfma(dev)
            result["sustained_mfma"] = sustained
        except Exception as e:  # a reported calibration, never a reason to lose the bench line
            result["sustained_mfma"] = {"error": f"{type(e).__name__}: {e}"}
        result.update(rooflines(rows, traffic, sustained))
        if args.workload == "c3_sliding" and args.precision == "f16x3":
            try:
                result["gemm_clock"] = gemm_clock_probe(dev)
            except Exception as e:  # a reported diagnosis, never a reason to lose the bench line
                result["gemm_clock"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and args.workload == "c3_sliding" and args.precision == "f16x3" and not args.no_extra_lines:
        # driver-timed versions of the numbers that used to exist only as builder-run files in profiles/ (<15 s together)
        extra = {}
        for key, kw in (("c3_sliding_f32", dict(name="c3_sliding", precision="f32", steps=2, warmup=1)),
                        ("c2_offline", dict(name="c2_offline", steps=5, warmup=2)),
                        ("c4_online", dict(name="c4_online", steps=12, warmup=3)),
                        ("c4_online_feature_cache", dict(name="c4_online", steps=12, warmup=3, feature_cache=True))):
            try:
                extra[key] = quick_line(dev=dev, **kw)
            except Exception as e:
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
        result["extra_lines"] = extra
        result["value_f32"] = extra["c3_sliding_f32"].get("value")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()
        if args.dist_backend == "nccl" and not args.single_device and not result["all_gather_ok"]:
            raise SystemExit(f"the all-gather took {all_gather_ms:.2f} ms = {100 * result['all_gather_share_of_step']:.2f} % of a step "
                             "(bar: < 1 %): the path's one collective is no longer noise -- see SURVEY 8e / DESIGN 6")


if __name__ == "__main__":
    main()
