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


def all_gather_tracks(tracks: torch.Tensor, vis: torch.Tensor, n_total: int, group=None, conf: torch.Tensor = None):
    """tracks [B,T,n_r,2], vis [B,T,n_r] (+ optionally conf [B,T,n_r]) of this rank -> ([B,T,N,2], [B,T,N](, [B,T,N])) on
    every rank.

    ONE fixed-size collective: every rank packs its chunk POINT-MAJOR into [per, B, T, C] (per = ceil(N/world), C = 3 or 4
    floats: x, y, visibility(, confidence)), `all_gather_into_tensor` (ncclAllGather under the "nccl" backend = RCCL; one hop
    per peer on the fully connected xGMI topology) fills a preallocated [world*per, B, T, C] buffer, and the results are
    permuted VIEWS of that buffer -- no list of per-rank tensors, no torch.cat, no second full-size copy (round 3 had both).
    The returned tensors are therefore not contiguous; call .contiguous() if a consumer needs that.
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
    try:
        dist.all_gather_into_tensor(full, packed, group=group)
    except (RuntimeError, NotImplementedError):  # a backend without the flat all-gather: per-rank VIEWS of the same buffer
        dist.all_gather([full[r * per:(r + 1) * per] for r in range(world)], packed, group=group)
    full = full[:n_total].permute(1, 2, 0, 3)  # [B,T,N,C] view
    vis_full = full[..., 2]
    if vis.dtype == torch.bool:
        vis_full = vis_full > 0.5
    out = (full[..., :2], vis_full)
    return out if conf is None else out + (full[..., 3],)


def track_sharded(predictor, video: torch.Tensor, queries: torch.Tensor, group=None, **kwargs):
    """Track `queries` [B,N,3] with the points sharded over the ranks of `group`.

    `predictor` is any callable with CoTrackerPredictor.forward's signature.  Returns the full
    ([B,T,N,2], [B,T,N]) on every rank.
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
    return all_gather_tracks(tracks, vis, queries.shape[1], group)


def dense_sharded(predictor, video: torch.Tensor, grid_query_frame: int = 0, grid_size: int = 80,
                  backward_tracking: bool = False, group=None):
    """Dense mode (predictor.py:70-98) with its grid_step^2 independent point chunks dealt out over the ranks of
    `group`: the reference tracks the chunks one after another on one device and concatenates them; here rank r
    tracks chunks r, r+world, ... with the same per-chunk call (`predictor._dense_chunk`), and ONE fixed-size
    all_gather per forward puts every chunk on every rank in the reference's chunk order.  Chunks all have the same
    number of points, so no padding is needed except for ranks that run out of chunks."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_chunks, n_pts = predictor._dense_layout(video, grid_size)
    B, T = video.shape[:2]
    per = (n_chunks + world - 1) // world
    mine = torch.zeros(B, T, per, n_pts, 3, device=video.device, dtype=torch.float32)
    for j in range(per):
        c = rank + j * world   # round-robin: every rank has work until the chunks run out
        if c >= n_chunks:
            break
        tr, vi = predictor._dense_chunk(video, c, grid_query_frame, grid_size, backward_tracking)
        mine[:, :, j, :, :2] = tr
        mine[:, :, j, :, 2] = vi.to(torch.float32)
    if world == 1:
        full = mine
    else:
        out = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(out, mine, group=group)
        full = torch.stack(out, dim=3).reshape(B, T, per * world, n_pts, 3)  # index j*world + r == chunk id
    full = full[:, :, :n_chunks].reshape(B, T, n_chunks * n_pts, 3)
    return full[..., :2].contiguous(), full[..., 2] > 0.5
