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
  roofline     -- dominant kernel, algorithmic flops / HIP-event duration vs the fp32-MFMA peak
  cpu_baseline -- the numpy oracle ("port") timed on this host's cores on a bounded sample
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
    "c4_online": (512, 512, 0, 32, False, 16, "512x512 cotracker3_online streaming, 16-frame chunks advancing 8, N=1024, "
                  "window replayed as ONE hipGraph per chunk (BASELINE.json configs[3])"),
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
    ap.add_argument("--no-graph", action="store_true", help="c4_online: direct launches instead of the captured hipGraph")
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


def cpu_baseline(window_len, overlap_factor, iters=6, n_points=96, reps=2):
    """Numpy oracle (oracle/, kind="port") on a bounded sample of the same workload: one S=16 window,
    N=96 points on a 96x128 pyramid, 1 update iteration.  Encoder excluded (update path only)."""
    import numpy as np
    from oracle import cotracker_oracle as O
    from cotracker_amd.model import CoTrackerThreeOnline
    from cotracker_amd.weights import fill_synthetic_

    m = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=window_len).eval()
    fill_synthetic_(m, seed=0)
    p = {k: v.numpy() for k, v in m.state_dict().items() if not k.startswith("fnet.")}
    r = np.random.RandomState(0)
    S, N = window_len, n_points
    f = r.standard_normal((1, S, 128, 96, 128)).astype(np.float32)
    pyr = O.build_pyramid(O.normalize_fmaps(f))
    qc = (r.uniform(0, 1, size=(1, N, 2)) * np.array([127, 95])).astype(np.float32)
    sup = [O.get_track_feat(pyr[i], np.zeros((1, N), np.int64), (qc / np.float32(2 ** i)).astype(np.float32))
           for i in range(4)]
    c = np.broadcast_to(qc.reshape(1, 1, N, 2), (1, S, N, 2)).astype(np.float32)
    z = np.zeros((1, S, N, 1), np.float32)
    t0 = time.time()
    for _ in range(reps):
        O.forward_window(pyr, c, sup, z, z, p, iters=1)
    dt = (time.time() - t0) / reps
    units_per_s = S * N / dt                      # (frame, point, iteration) units per second
    pf_per_s = units_per_s / (iters * overlap_factor)
    try:
        import threadpoolctl
        cores = max([i.get("num_threads", 1) for i in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    return {"value": round(pf_per_s, 2), "unit": "tracked-point-frames/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle, update path only (no encoder): 1 window S={S}, N={N}, 1 iteration "
                      f"({dt:.1f} s) on a 96x128 4-level pyramid; scaled by {iters} iters x {overlap_factor:.3f} window overlap",
            "host_logical_cpus": os.cpu_count()}


def roofline_entry(row, traffic, force_hbm=False):
    """Roofline object of one kernel row of the HIP-event recorder.  `achieved` = ALGORITHMIC work of the launches /
    their summed HIP-event duration.  Split-half kernels (gemm_sh_* / gemm_f16x3_* / corr_volume_sh) form every
    f32-class product from 3 f16 MFMAs, so their MFMA ceiling in algorithmic flops is 2500/3 = 833 TF/s."""
    name = row["name"]
    sec = row["total_ms"] * 1e-3
    n = max(row["launches"], 1)
    tr = traffic.get(name)
    out = {"kernel": name, "launches": row["launches"], "avg_launch_us": round(1e6 * sec / n, 1)}
    if force_hbm:
        gbs = row["bytes"] / sec / 1e9
        out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": row["bytes"] / n,
                    "note": "algorithmic sampler bytes (SURVEY 8d: 8x8x128 footprint per (t,n,level) + support/S + volume out, "
                            "no inter-point reuse assumed) / HIP-event time; neighbouring grid points share footprint pixels "
                            "in L2, so measured HBM traffic is lower than the algorithmic figure"})
    else:
        ach = row["flops"] / sec / 1e12
        split = name.startswith(("gemm_sh", "gemm_f16x3", "corr_volume_sh"))
        peak = F16_MFMA_PEAK_TFLOPS / 3.0 if split else FP32_MFMA_PEAK_TFLOPS
        out.update({"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "flops_per_launch": row["flops"] / n})
        if split:
            out.update({"mfma_issued": round(3 * ach, 1), "mfma_peak": F16_MFMA_PEAK_TFLOPS, "f32_mfma_peak": FP32_MFMA_PEAK_TFLOPS,
                        "note": "achieved = algorithmic (f32-equivalent) flops / HIP-event time; every product is 3 "
                                "v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, f32 accumulate), so peak = dense f16 MFMA "
                                "2500 TF/s / 3; frac is also mfma_issued / mfma_peak.  The exact-f32 MFMA peak is 157.3 TF/s"})
        else:
            out["note"] = "exact-f32 MFMA (v_mfma_f32_32x32x2_f32) peak"
    if tr:
        out["traffic"] = tr.get("hbm_bytes_per_launch")
        out["traffic_detail"] = {k: tr[k] for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "dispatches", "source") if k in tr}
    else:
        out["traffic"] = None
    return out


def parity_probe(dev):
    """Tiny window vs the oracle (same check as __graft_entry__.smoke): max-abs errors."""
    import numpy as np
    from oracle import cotracker_oracle as O
    from cotracker_amd import ops
    from cotracker_amd.model import CoTrackerThreeOnline
    from cotracker_amd.weights import fill_synthetic_

    m = CoTrackerThreeOnline(stride=4, corr_radius=3, window_len=8, model_resolution=(96, 128)).eval()
    fill_synthetic_(m, seed=3)
    p = {k: v.numpy() for k, v in m.state_dict().items() if not k.startswith("fnet.")}
    m = m.to(dev)
    r = np.random.RandomState(0)
    S, N = 8, 12
    f = r.standard_normal((1, S, 128, 24, 32)).astype(np.float32)
    pyr = O.build_pyramid(O.normalize_fmaps(f))
    qf = r.randint(0, S, size=(1, N))
    qc = (r.uniform(0, 1, size=(1, N, 2)) * np.array([31, 23])).astype(np.float32)
    sup = [O.get_track_feat(pyr[i], qf, (qc / np.float32(2 ** i)).astype(np.float32)) for i in range(4)]
    cinit = np.broadcast_to(qc.reshape(1, 1, N, 2), (1, S, N, 2)).astype(np.float32)
    z = np.zeros((1, S, N, 1), np.float32)
    c, v, cf = O.forward_window(pyr, cinit, sup, z, z, p, iters=6, model_resolution=(96, 128))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    fm = [T(np.transpose(x[0], (0, 2, 3, 1))) for x in pyr]
    sp = [T(np.transpose(s[0], (1, 0, 2))) for s in sup]
    coords, vis, conf = T(cinit[0]), torch.zeros(S, N, device=dev), torch.zeros(S, N, device=dev)
    ops.forward_window(ops.Window(fm, sp, coords, vis, conf, (32.0, 24.0), iters=6), m.packed(dev))
    torch.cuda.synchronize()
    return {"coords_px": float((coords.cpu() - torch.from_numpy(c[0])).abs().max()) * 4.0,
            "vis_logit": float((vis.cpu() - torch.from_numpy(v[0, ..., 0])).abs().max()),
            "conf_logit": float((conf.cpu() - torch.from_numpy(cf[0, ..., 0])).abs().max()),
            "against": "numpy oracle, S=8 N=12 6 iterations"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from cotracker_amd import model as ctk_model
    from cotracker_amd import ops
    from cotracker_amd.predictor import CoTrackerPredictor, get_points_on_a_grid
    ctk_model.DEFAULT_PRECISION = args.precision
    from cotracker_amd.sharding import all_gather_tracks
    from cotracker_amd.synthetic import synthetic_video
    from cotracker_amd.weights import fill_synthetic_

    H, W, T, G, offline, wl, desc = WORKLOADS[args.workload]
    N = G * G
    streaming = args.workload == "c4_online"
    if streaming:
        from cotracker_amd.predictor import CoTrackerOnlinePredictor
        pred = CoTrackerOnlinePredictor(checkpoint=None, window_len=wl)
        pred.model.hip_graph = not args.no_graph
        T = pred.step * (args.steps + args.warmup + 3)  # one chunk call per step, plus the profiled call
    elif offline == "v2":
        pred = CoTrackerPredictor(checkpoint=None, v2=True, window_len=wl)
    else:
        pred = CoTrackerPredictor(checkpoint=None, offline=offline, window_len=wl)
    fill_synthetic_(pred.model, seed=0)
    pred = pred.to(dev)
    video = synthetic_video(T, H, W, seed=1234).to(dev)  # resident in HBM before timing starts

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
    elif world == 1:
        def step():
            return pred(video, grid_size=G)
    else:
        # weak scaling: rank r tracks its own G*G grid, shifted by a sub-pixel offset (a denser joint grid)
        ih, iw = pred.interp_shape
        pts = get_points_on_a_grid(G, (ih, iw), device=dev)
        pts = pts + torch.tensor([0.37, 0.23], device=dev) * rank
        pts = pts * torch.tensor([(W - 1) / (iw - 1), (H - 1) / (ih - 1)], device=dev)  # raw-video pixels
        q = torch.cat([torch.zeros_like(pts[:, :, :1]), pts], dim=2)

        def step():
            tr, vi = pred(video, queries=q)
            return all_gather_tracks(tr, vi, N * world)  # final tracks of every rank on every rank (RCCL)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    sec_per_step = float(tmax.item()) / args.steps
    assert torch.isfinite(out[0]).all()
    frames_per_step = pred.step if streaming else T
    value = world * N * frames_per_step / sec_per_step

    result = {
        "metric": "tracked-point-frames/sec (N*T/s)", "value": round(value, 1), "unit": "tracked-point-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (Linear layers as split-half f16 MFMA x3, f32 accumulate)" if args.precision == "f16x3" else "f32",
        "data": "synthetic",
        "config": {"workload": desc, "name": args.workload, "points_per_gpu": N, "frames": frames_per_step, "video": [H, W],
                   "iters": 6, "window_len": wl, "offline": bool(offline) and offline != "v2", "sharding": f"points x{world}", "precision": args.precision,
                   "weights": "seeded synthetic (no checkpoints offline)"},
    }
    if streaming:
        result["config"]["hip_graph"] = bool(pred.model.hip_graph)
        result["config"]["graph_nodes"] = next(iter(pred.model._graphs.values())).nodes if pred.model._graphs else 0

    if rank == 0 and not args.no_profile:
        # one extra step with the library's HIP-event recorder on (events on the launch stream)
        if streaming:
            pred.model.hip_graph = False  # events cannot be recorded inside a captured graph: profile the direct launches
        ops.profile_enable(True)
        t1 = time.perf_counter()
        step() if streaming else pred(video, grid_size=G)
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
        traffic = {}
        if os.path.exists(args.pmc_traffic):
            traffic = json.load(open(args.pmc_traffic))
            if traffic.pop("_workload", "c3_sliding") != args.workload:
                traffic = {}  # the PMC passes were collected on another workload: per-launch bytes do not transfer
        if rows:
            result["roofline"] = roofline_entry(rows[0], traffic)
            for r in rows:
                if r["name"].startswith("corr_volume"):
                    result["roofline_sampler"] = roofline_entry(r, traffic, force_hbm=True)
    if rank == 0:
        result["parity"] = parity_probe(dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        overlap = 1.0
        if not offline or offline == "v2":
            S, step_ = wl, wl // 2
            nwin = (T - S + step_ - 1) // step_ + 1
            overlap = nwin * S / T
        result["cpu_baseline"] = cpu_baseline(16 if not offline else 16, overlap)  # CoTracker3 update path sample
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
