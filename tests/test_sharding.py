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
