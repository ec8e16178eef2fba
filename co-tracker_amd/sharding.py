"""Multi-GPU sharding of query points: one process per GPU, contiguous chunks of the
query list tracked independently (exactly what the reference's dense mode does in sequence,
predictor.py:80-96), then ONE all-gather of the final (tracks, visibility) over RCCL/xGMI.

Every stage of the update is per-(frame, point) except the space attention, where the 64 virtual
tracks attend over all points of a call -- so a sharded run equals the reference run *on the same
chunks* (not the joint run; they differ by ~1e-3 px even at random init).
There is no collective on the data path itself; the encoder is replicated on every rank.
"""
from typing import Tuple

import torch
import torch.distributed as dist


def chunk_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous chunk [lo, hi) of rank `rank`: ceil(n/world) points each, last ranks may be short/empty."""
    per = (n + world - 1) // world
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def shard_queries(queries: torch.Tensor, world: int, rank: int) -> torch.Tensor:
    """queries [B,N,3] -> this rank's contiguous chunk [B,n_r,3]."""
    lo, hi = chunk_bounds(queries.shape[1], world, rank)
    return queries[:, lo:hi]


# Which collective carries the gather is decided WITHOUT communicating (review of round 5: a probe collective cached per process,
# not per group, can leave ranks with different cache states issuing different collectives): the "nccl" backend (= RCCL on ROCm)
# has ncclAllGather behind all_gather_into_tensor; every other backend (gloo in the CPU / single-device tests) takes
# dist.all_gather on per-rank VIEWS of the same preallocated buffer -- the same bytes in the same place, still no list of fresh
# tensors and no torch.cat.  The choice is a pure function of the group's backend name, identical on every rank of the group, and
# an error of the collective itself (RCCL failure, timeout, shape mismatch between ranks) propagates to the caller.
def _flat_all_gather_ok(device: torch.device, group=None) -> bool:
    return str(dist.get_backend(group)) == "nccl" and device.type == "cuda"


def _gather_flat(full: torch.Tensor, mine: torch.Tensor, world: int, group=None) -> None:
    """full [world * k, ...] <- every rank's mine [k, ...] in rank order, ONE collective; errors propagate."""
    if _flat_all_gather_ok(mine.device, group):
        dist.all_gather_into_tensor(full, mine, group=group)
    else:  # per-rank VIEWS of the same preallocated buffer: still no list of fresh tensors, no cat
        k = mine.shape[0]
        dist.all_gather([full[r * k:(r + 1) * k] for r in range(world)], mine, group=group)


def all_gather_tracks(tracks: torch.Tensor, vis: torch.Tensor, n_total: int, group=None, conf: torch.Tensor = None):
    """tracks [B,T,n_r,2], vis [B,T,n_r] (+ optionally conf [B,T,n_r]) of this rank -> ([B,T,N,2], [B,T,N](, [B,T,N])) on
    every rank.

    ONE fixed-size collective: every rank packs its chunk POINT-MAJOR into [per, B, T, C] (per = ceil(N/world), C = 3 or 4
    floats: x, y, visibility(, confidence)), `all_gather_into_tensor` (ncclAllGather under the "nccl" backend = RCCL; one hop
    per peer on the fully connected xGMI topology) fills a preallocated [world*per, B, T, C] buffer, and the results are
    permuted VIEWS of that buffer -- no list of per-rank tensors, no torch.cat, no second full-size copy (round 3 had both).
    The returned tensors are therefore NOT contiguous (`track_sharded`, the public entry point, returns contiguous ones).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (tracks, vis) if conf is None else (tracks, vis, conf)
    per = (n_total + world - 1) // world
    B, T, n_r = tracks.shape[:3]
    C = 3 if conf is None else 4
    packed = torch.zeros(per, B, T, C, device=tracks.device, dtype=torch.float32)
    packed[:n_r, :, :, :2] = tracks.permute(2, 0, 1, 3)
    packed[:n_r, :, :, 2] = vis.to(torch.float32).permute(2, 0, 1)
    if conf is not None:
        packed[:n_r, :, :, 3] = conf.to(torch.float32).permute(2, 0, 1)
    full = torch.empty(world * per, B, T, C, device=tracks.device, dtype=torch.float32)
    _gather_flat(full, packed, world, group)
    full = full[:n_total].permute(1, 2, 0, 3)  # [B,T,N,C] view
    vis_full = full[..., 2]
    if vis.dtype == torch.bool:
        vis_full = vis_full > 0.5
    out = (full[..., :2], vis_full)
    return out if conf is None else out + (full[..., 3],)


def track_sharded(predictor, video: torch.Tensor, queries: torch.Tensor, group=None, **kwargs):
    """Track `queries` [B,N,3] with the points sharded over the ranks of `group`.

    `predictor` is any callable with CoTrackerPredictor.forward's signature.  Returns the full
    ([B,T,N,2], [B,T,N]) on every rank, CONTIGUOUS like the predictor's own outputs (one 12-byte-per-point-frame copy of the
    gathered buffer: 12.6 MB per rank at BASELINE configs[4]).
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = shard_queries(queries, world, rank)
    if mine.shape[1] > 0:
        tracks, vis = predictor(video, queries=mine, **kwargs)
    else:  # more ranks than points
        B, T = video.shape[:2]
        tracks = torch.zeros(B, T, 0, 2, device=video.device)
        vis = torch.zeros(B, T, 0, device=video.device, dtype=torch.bool)
    tracks, vis = all_gather_tracks(tracks, vis, queries.shape[1], group)
    return tracks.contiguous(), vis.contiguous()


def dense_sharded(predictor, video: torch.Tensor, grid_query_frame: int = 0, grid_size: int = 80,
                  backward_tracking: bool = False, group=None):
    """Dense mode (predictor.py:70-98) with its grid_step^2 independent point chunks dealt out over the ranks of
    `group`: the reference tracks the chunks one after another on one device and concatenates them; here rank r
    tracks chunks r, r+world, ... with the same per-chunk call (`predictor._dense_chunk`), and ONE fixed-size
    flat all-gather per forward (the same `all_gather_into_tensor` buffer scheme as `all_gather_tracks`: no list of
    per-rank tensors, no torch.stack) puts every chunk on every rank.  Chunks all have the same number of points, so no
    padding is needed except for ranks that run out of chunks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_chunks, n_pts = predictor._dense_layout(video, grid_size)
    B, T = video.shape[:2]
    per = (n_chunks + world - 1) // world
    mine = torch.zeros(per, B, T, n_pts, 3, device=video.device, dtype=torch.float32)  # chunk-major: the gather's leading axis
    for j in range(per):
        c = rank + j * world   # round-robin: every rank has work until the chunks run out
        if c >= n_chunks:
            break
        tr, vi = predictor._dense_chunk(video, c, grid_query_frame, grid_size, backward_tracking)
        mine[j, :, :, :, :2] = tr
        mine[j, :, :, :, 2] = vi.to(torch.float32)
    if world == 1:
        full = mine  # [per, B, T, n_pts, 3], chunk j == chunk id
    else:
        buf = torch.empty(world * per, B, T, n_pts, 3, device=video.device, dtype=torch.float32)
        _gather_flat(buf, mine, world, group)
        # buf[r * per + j] = chunk j * world + r: a VIEW in chunk-id order (the one copy is the .contiguous() below)
        full = buf.view(world, per, B, T, n_pts, 3).permute(1, 0, 2, 3, 4, 5).reshape(per * world, B, T, n_pts, 3)
    full = full[:n_chunks].permute(1, 2, 0, 3, 4).reshape(B, T, n_chunks * n_pts, 3)
    return full[..., :2].contiguous(), full[..., 2] > 0.5
