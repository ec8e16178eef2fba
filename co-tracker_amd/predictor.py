"""Predictor wrappers with the reference's public API (cotracker/predictor.py).

``CoTrackerPredictor.forward(video, queries, segm_mask, grid_size, grid_query_frame,
backward_tracking)`` (predictor.py:36-68) and ``CoTrackerOnlinePredictor.forward(video_chunk,
is_first_step, queries, grid_size, grid_query_frame, add_support_grid)`` (predictor.py:230-309)
keep their signatures, defaults, return values and quirks: the offline predictor
thresholds visibility alone at 0.9, the online one thresholds visibility*confidence at 0.6, the
query-frame prediction is overwritten with the query itself, dense mode derives its step from
the raw video width.  All tensor work stays on the GPU (the reference's per-batch Python fix-up
loop, predictor.py:177-185, is a single indexed write here).
"""
import torch
import torch.nn.functional as F

from .build_cotracker import build_cotracker


def get_points_on_a_grid(size, extent, center=None, device="cpu"):
    """size x size points covering an (H, W) extent with margin W/64, row-major, as (x, y)
    (behaviour of cotracker/models/core/model_utils.py:83-139)."""
    H, W = float(extent[0]), float(extent[1])
    if size == 1:
        return torch.tensor([W / 2, H / 2], device=device)[None, None]
    cy, cx = (H / 2, W / 2) if center is None else (float(center[0]), float(center[1]))
    m = W / 64
    # endpoints in the reference's evaluation order (python doubles are not associative)
    ys = torch.linspace(m - H / 2 + cy, H / 2 + cy - m, size, device=device)
    xs = torch.linspace(m - W / 2 + cx, W / 2 + cx - m, size, device=device)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack([gx, gy], dim=-1).reshape(1, -1, 2)


def _cat(a, b, dim):
    return b if a is None else torch.cat([a, b], dim=dim)


class CoTrackerPredictor(torch.nn.Module):
    def __init__(self, checkpoint="./checkpoints/scaled_offline.pth", offline=True, v2=False, window_len=60):
        super().__init__()
        self.v2 = v2
        self.support_grid_size = 6
        model = build_cotracker(checkpoint, v2=v2, offline=offline, window_len=window_len)
        self.interp_shape = model.model_resolution
        self.model = model
        self.model.eval()

    @torch.no_grad()
    def forward(self, video, queries: torch.Tensor = None, segm_mask: torch.Tensor = None, grid_size: int = 0,
                grid_query_frame: int = 0, backward_tracking: bool = False):
        if queries is None and grid_size == 0:
            return self._compute_dense_tracks(video, grid_query_frame=grid_query_frame,
                                              backward_tracking=backward_tracking)
        return self._compute_sparse_tracks(video, queries, segm_mask, grid_size,
                                           add_support_grid=(grid_size == 0 or segm_mask is not None),
                                           grid_query_frame=grid_query_frame, backward_tracking=backward_tracking)

    # dense mode: grid_step^2 independent point chunks (predictor.py:70-98).  With `dense_group` set (a process group, or
    # True for the default group) the chunks are dealt out over the ranks and all-gathered (sharding.dense_sharded);
    # otherwise they are tracked in sequence on this device, as in the reference.
    dense_group = None

    def _dense_layout(self, video, grid_size=80):
        H, W = video.shape[-2:]
        step = W // grid_size  # the reference derives the step from the raw video WIDTH for both axes (predictor.py:73)
        return step * step, (W // step) * (H // step)

    def _dense_chunk(self, video, offset, grid_query_frame, grid_size=80, backward_tracking=False):
        H, W = video.shape[-2:]
        step = W // grid_size
        gw, gh = W // step, H // step
        pts = torch.zeros(video.shape[0], gw * gh, 3, device=video.device)
        pts[:, :, 0] = grid_query_frame
        pts[:, :, 1] = torch.arange(gw, device=video.device).repeat(gh) * step + offset % step
        pts[:, :, 2] = torch.arange(gh, device=video.device).repeat_interleave(gw) * step + offset // step
        return self._compute_sparse_tracks(video=video, queries=pts, backward_tracking=backward_tracking)

    def _compute_dense_tracks(self, video, grid_query_frame, grid_size=80, backward_tracking=False):
        if self.dense_group is not None:
            from .sharding import dense_sharded
            return dense_sharded(self, video, grid_query_frame, grid_size, backward_tracking,
                                 group=None if self.dense_group is True else self.dense_group)
        tracks = vis = None
        for offset in range(self._dense_layout(video, grid_size)[0]):
            t_step, v_step = self._dense_chunk(video, offset, grid_query_frame, grid_size, backward_tracking)
            tracks = _cat(tracks, t_step, 2)
            vis = _cat(vis, v_step, 2)
        return tracks, vis

    def _resize(self, video):
        B, T, C, H, W = video.shape
        v = F.interpolate(video.reshape(B * T, C, H, W).float(), tuple(self.interp_shape), mode="bilinear",
                          align_corners=True)
        return v.reshape(B, T, 3, self.interp_shape[0], self.interp_shape[1])

    def _compute_sparse_tracks(self, video, queries, segm_mask=None, grid_size=0, add_support_grid=False,
                               grid_query_frame=0, backward_tracking=False):
        B, T, C, H, W = video.shape
        ih, iw = self.interp_shape
        video = self._resize(video)  # predictor.py:112-116
        if queries is not None:
            assert queries.shape[2] == 3
            queries = queries.clone().float()
            queries[:, :, 1:] *= queries.new_tensor([(iw - 1) / (W - 1), (ih - 1) / (H - 1)])
        elif grid_size > 0:
            pts = get_points_on_a_grid(grid_size, self.interp_shape, device=video.device)
            if segm_mask is not None:
                segm_mask = F.interpolate(segm_mask, tuple(self.interp_shape), mode="nearest")
                keep = segm_mask[0, 0][pts[0, :, 1].round().long(), pts[0, :, 0].round().long()].bool()
                pts = pts[:, keep]
            queries = torch.cat([torch.full_like(pts[:, :, :1], float(grid_query_frame)), pts], dim=2).repeat(B, 1, 1)
        if add_support_grid:
            g = get_points_on_a_grid(self.support_grid_size, self.interp_shape, device=video.device)
            g = torch.cat([torch.zeros_like(g[:, :, :1]), g], dim=2).repeat(B, 1, 1)
            queries = torch.cat([queries, g], dim=1)

        tracks, vis, *_ = self.model.forward(video=video, queries=queries, iters=6)

        if backward_tracking:
            tracks, vis = self._compute_backward_tracks(video, queries, tracks, vis)
            if add_support_grid:
                queries[:, -self.support_grid_size ** 2:, 0] = T - 1
        if add_support_grid:
            tracks = tracks[:, :, : -self.support_grid_size ** 2]
            vis = vis[:, :, : -self.support_grid_size ** 2]
        vis = vis > 0.9  # confidence is ignored by the offline predictor (predictor.py:170-171)

        # the query point itself is the prediction at its query frame, and visible (predictor.py:177-185)
        n = tracks.shape[2]
        qt = queries[:, :n, 0].long()
        bi = torch.arange(B, device=tracks.device)[:, None].expand(B, n)
        ni = torch.arange(n, device=tracks.device)[None, :].expand(B, n)
        tracks[bi, qt, ni] = queries[:, :n, 1:]
        vis[bi, qt, ni] = True

        tracks = tracks * tracks.new_tensor([(W - 1) / (iw - 1), (H - 1) / (ih - 1)])
        return tracks, vis

    def _compute_backward_tracks(self, video, queries, tracks, vis):  # predictor.py:192-209
        T = video.shape[1]
        inv_q = queries.clone()
        inv_q[:, :, 0] = T - inv_q[:, :, 0] - 1
        inv_tracks, inv_vis, *_ = self.model(video=video.flip(1).contiguous(), queries=inv_q, iters=6)
        inv_tracks, inv_vis = inv_tracks.flip(1), inv_vis.flip(1)
        before = torch.arange(T, device=queries.device)[None, :, None] < queries[:, None, :, 0]
        tracks = torch.where(before[..., None], inv_tracks, tracks)
        vis = torch.where(before, inv_vis, vis)
        return tracks, vis


class CoTrackerOnlinePredictor(torch.nn.Module):
    def __init__(self, checkpoint="./checkpoints/scaled_online.pth", offline=False, v2=False, window_len=16):
        super().__init__()
        self.v2 = v2
        self.support_grid_size = 6
        model = build_cotracker(checkpoint, v2=v2, offline=False, window_len=window_len)
        self.interp_shape = model.model_resolution
        self.step = model.window_len // 2
        self.model = model
        self.model.eval()
        self.model.hip_graph = True  # streaming: replay the captured window graph per chunk (configs[3])

    def finish(self):
        """Not in the reference (its stream has no end marker): examine the f16-range check of the LAST chunk, which graph
        streaming defers to the next call (behaviours table).  Raises FloatingPointError like that next call
        would; a no-op for CoTracker2 (unguarded) and when nothing is pending."""
        resolve = getattr(self.model, "_resolve_deferred_range_check", None)
        if resolve is not None:
            resolve()

    @torch.no_grad()
    def forward(self, video_chunk, is_first_step: bool = False, queries: torch.Tensor = None, grid_size: int = 5,
                grid_query_frame: int = 0, add_support_grid=False):
        B, T, C, H, W = video_chunk.shape
        ih, iw = self.interp_shape
        if is_first_step:  # predictor.py:242-274: reset state, remember the queries, no tracking yet
            self.model.init_video_online_processing()
            self._prev_chunk = None
            if queries is not None:
                assert queries.shape[2] == 3
                self.N = queries.shape[1]
                queries = queries.clone().float()
                queries[:, :, 1:] *= queries.new_tensor([(iw - 1) / (W - 1), (ih - 1) / (H - 1)])
                if add_support_grid:
                    g = get_points_on_a_grid(self.support_grid_size, self.interp_shape, device=video_chunk.device)
                    queries = torch.cat([queries, torch.cat([torch.zeros_like(g[:, :, :1]), g], dim=2)], dim=1)
            elif grid_size > 0:
                pts = get_points_on_a_grid(grid_size, self.interp_shape, device=video_chunk.device)
                self.N = grid_size ** 2
                queries = torch.cat([torch.full_like(pts[:, :, :1], float(grid_query_frame)), pts], dim=2)
            self.queries = queries
            return (None, None)

        # streaming feature cache (opt-in, model.online_feature_cache): prove ON THE HOST that this chunk's first T - step frames
        # are the memory of the previous chunk's last T - step frames (a view of the same resident video advanced by `step`
        # frames) and tell the model -- it only ever sees the freshly resized tensor below.  No device work, no wait.
        if getattr(self.model, "online_feature_cache", False):
            from .model import tail_aliases
            prev = getattr(self, "_prev_chunk", None)
            self.model._overlap_hint = bool(tail_aliases(prev, video_chunk, 1, self.step))
            self._prev_chunk = video_chunk  # a reference: keeps the allocation alive until the next call
            try:
                video_chunk._ctk_version = video_chunk._version
            except Exception:
                pass
        v = F.interpolate(video_chunk.reshape(B * T, C, H, W).float(), tuple(self.interp_shape), mode="bilinear",
                          align_corners=True).reshape(B, T, 3, ih, iw)
        if self.v2:  # CoTracker2 returns (tracks, visibility, train_data): no confidence (predictor.py:283-286)
            tracks, vis, _ = self.model(video=v, queries=self.queries, iters=6, is_online=True)
            conf = None
        else:
            tracks, vis, conf, _ = self.model(video=v, queries=self.queries, iters=6, is_online=True)
        if add_support_grid:
            tracks, vis = tracks[:, :, :self.N], vis[:, :, :self.N]
            conf = conf[:, :, :self.N] if conf is not None else None
        if conf is not None:
            vis = vis * conf  # predictor.py:297-298
        return tracks * tracks.new_tensor([(W - 1) / (iw - 1), (H - 1) / (ih - 1)]), vis > 0.6
