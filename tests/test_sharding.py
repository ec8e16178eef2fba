"""Multi-process point sharding on CPU (gloo, world_size 2): chunking, the fixed-size all-gather with an
uneven last chunk, and agreement with the single-process result for a per-point tracker stub."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub_predictor(video, queries=None, **kw):
    """Per-point 'tracker': tracks[b,t,n] = query_xy + t * mean(video[b,t]); vis = x > 50."""
    B, T = video.shape[:2]
    drift = video.reshape(B, T, -1).mean(-1)
    tracks = queries[:, None, :, 1:3] + drift[:, :, None, None] * torch.arange(T)[None, :, None, None]
    vis = (queries[:, None, :, 1] > 50).expand(B, T, -1)
    return tracks, vis


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cotracker_amd.sharding import track_sharded
    g = torch.Generator().manual_seed(0)
    video = torch.rand(1, 5, 3, 8, 8, generator=g)
    q = torch.rand(1, n, 3, generator=g) * 100
    tr, vi = track_sharded(_stub_predictor, video, q)
    ref_t, ref_v = _stub_predictor(video, queries=q)
    ok = torch.allclose(tr, ref_t) and torch.equal(vi, ref_v) and tr.shape == (1, 5, n, 2)
    # the optional confidence column rides in the same collective (online predictor: visibility * confidence upstream)
    from cotracker_amd.sharding import all_gather_tracks, chunk_bounds
    lo, hi = chunk_bounds(n, world, rank)
    conf_ref = ref_t[..., 0] * 0.01
    tr3, vi3, cf3 = all_gather_tracks(ref_t[:, :, lo:hi], ref_v[:, :, lo:hi], n, conf=conf_ref[:, :, lo:hi])
    ok = ok and torch.equal(tr3, ref_t) and torch.equal(vi3, ref_v) and torch.equal(cf3, conf_ref) and vi3.dtype == torch.bool
    # ONE buffer: tracks / visibility / confidence are views of the same gathered allocation (no cat, no second copy)
    ok = ok and tr3.untyped_storage().data_ptr() == cf3.untyped_storage().data_ptr()
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_chunk_bounds():
    from cotracker_amd.sharding import chunk_bounds
    for n, w in [(70225, 8), (7, 2), (3, 8), (6400, 4)]:
        spans = [chunk_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert chunk_bounds(70225, 8, 0) == (0, 8779) and chunk_bounds(70225, 8, 7) == (61453, 70225)


def test_track_sharded_gloo_world2():
    for n in (7, 10):  # uneven and even split
        port = _free_port()
        with mp.Manager() as mgr:
            out = mgr.dict()
            mp.spawn(_worker, args=(2, port, n, out), nprocs=2, join=True)
            assert out[0] and out[1]


def test_track_sharded_gloo_world4_with_short_and_empty_ranks():
    """ceil(N / world) points per rank: N = 10 on 4 ranks is 3 + 3 + 3 + 1 (a short last chunk, zero-padded in the fixed-size
    all-gather), N = 3 on 4 ranks leaves rank 3 without a point (it still joins the collective with an empty chunk)."""
    for n in (10, 3):
        port = _free_port()
        with mp.Manager() as mgr:
            out = mgr.dict()
            mp.spawn(_worker, args=(4, port, n, out), nprocs=4, join=True)
            assert all(out[r] for r in range(4)), dict(out)


# ---- the same path on the GPU: two gloo ranks sharing cuda:0, a real tracker under track_sharded -----------------
import pytest  # noqa: E402


def _gpu_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cotracker_amd.predictor import CoTrackerPredictor
        from cotracker_amd.sharding import track_sharded, chunk_bounds
        from cotracker_amd.synthetic import synthetic_video
        from cotracker_amd.weights import fill_synthetic_
        dev = torch.device("cuda:0")
        p = CoTrackerPredictor(checkpoint=None, offline=False, window_len=8)
        fill_synthetic_(p.model, seed=0)
        p = p.to(dev)
        video = synthetic_video(12, 96, 160, seed=3).to(dev)
        g = torch.Generator().manual_seed(1)
        n = 37  # uneven: 19 + 18
        q = torch.cat([torch.randint(0, 6, (1, n, 1), generator=g).float(), torch.rand(1, n, 1, generator=g) * 159,
                       torch.rand(1, n, 1, generator=g) * 95], dim=2).to(dev)
        tr, vi = track_sharded(p, video, q)
        # the oracle of SURVEY 8e: the same predictor on the same chunks, one after the other
        seq_t, seq_v = [], []
        for r in range(world):
            lo, hi = chunk_bounds(n, world, r)
            t_, v_ = p(video, queries=q[:, lo:hi])
            seq_t.append(t_)
            seq_v.append(v_)
        seq_t, seq_v = torch.cat(seq_t, dim=2), torch.cat(seq_v, dim=2)
        err = float((tr - seq_t).abs().max())
        flips = int((vi != seq_v).sum())
        # dense mode (predictor.py:70-98) with its chunks dealt out over the ranks == the sequential dense run
        p.dense_group = True
        dt, dv = p(video[:, :8])
        p.dense_group = None
        st, sv = p(video[:, :8])
        derr = float((dt - st).abs().max())
        dflips = int((dv != sv).sum())
        out[rank] = (tuple(tr.shape), err, flips, tuple(dt.shape), derr, dflips)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_track_sharded_two_ranks_on_gpu_equals_sequential_chunks():
    """Two gloo ranks on cuda:0 (the 1-GPU box's stand-in for two GPUs): sharded tracking == the same predictor run on
    the same contiguous chunks in sequence -- bit for bit (deterministic HIP encoder and update path; round 3 allowed 1e-4 px
    for MIOpen's run-to-run noise)."""
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gpu_worker, args=(2, port, out), nprocs=2, join=True)
        for r in (0, 1):
            shape, err, flips, dshape, derr, dflips = out[r]
            assert shape == (1, 12, 37, 2)
            assert err == 0.0 and flips == 0, out[r]
            assert dshape == (1, 8, 4 * 80 * 48, 2)
            assert derr == 0.0 and dflips == 0, out[r]


# ---- RCCL, two real devices (round 5): collected everywhere, runs only where a second GPU exists ------------------------------
@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_track_sharded_nccl_world2():
    """BASELINE configs[4] on two real devices: `bench.py --gpus 2 --workload c5_shard` launches one rank per GPU under
    torch.distributed.run with the "nccl" backend (= RCCL), rank r tracks chunk r of the 265 x 265 grid, ONE
    all_gather_into_tensor returns the tracks; rank 0's timed chunk is checked against the reference's CPU run of chunk 0
    (tests/golden: scale_c5_chunk0) inside the bench line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c5_shard", "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extra-lines", "--no-profile"], capture_output=True, text=True, env=env, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dist_backend"] == "nccl"
    devs = {(d["device"], d["device_index"]) for d in line["rank_devices"]}
    assert len(devs) == 2, line["rank_devices"]
    par = line["parity"]["timed_step"]
    assert par["coords_px"] <= 1e-3 and par["vis_logit"] <= 1e-4 and par["conf_logit"] <= 1e-4, par  # north_star's bars
    assert line["all_gather_ms"] is not None
