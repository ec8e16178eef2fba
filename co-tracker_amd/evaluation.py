"""Accuracy-evaluation front end of the reference on the HIP models.

``EvaluationPredictor`` mirrors cotracker/models/evaluation_predictor.py:25-213 (same constructor, ``forward(video,
queries) -> (tracks [B,T,N,2] raw-video px, visibility*confidence [B,T,N])``): the TAP-Vid protocol tracks every query
point either on its own together with a local 8x8 support grid and a global 5x5 grid (``single_point=True``, one model
call per query) or jointly with the global grid.  ``compute_tapvid_metrics`` is the TAP-Vid metric of
cotracker/evaluation/core/eval_utils.py:12-138 (occlusion accuracy, <delta^x, Jaccard) restated on numpy.

The datasets and the ``Evaluator`` loop (TAP-Vid / Dynamic Replica readers, hydra configs) stay out of scope: they are
storage / IO, not part of the tracking path.  SIFT support points need OpenCV (absent here): ``sift_size > 0`` raises.
"""
from typing import Mapping, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .predictor import get_points_on_a_grid


def get_uniformly_sampled_pts(size: int, num_frames: int, extent, device="cpu") -> torch.Tensor:
    """`size` random (t, x, y) queries (model_utils.py:22-33); consumes the torch RNG exactly as the reference does."""
    t = torch.randint(low=0, high=num_frames, size=(size, 1), device=device)
    xy = torch.rand(size, 2, device=device) * torch.tensor([extent[1], extent[0]], device=device)
    return torch.cat((t, xy), dim=1)[None]


class EvaluationPredictor(torch.nn.Module):
    def __init__(self, cotracker_model, interp_shape: Tuple[int, int] = (384, 512), grid_size: int = 5,
                 local_grid_size: int = 8, single_point: bool = True, sift_size: int = 0,
                 num_uniformly_sampled_pts: int = 0, n_iters: int = 6, local_extent: int = 50) -> None:
        super().__init__()
        if sift_size > 0:
            raise NotImplementedError("SIFT support points need OpenCV (get_sift_sampled_pts, model_utils.py:60-80)")
        self.grid_size = grid_size
        self.local_grid_size = local_grid_size
        self.sift_size = sift_size
        self.single_point = single_point
        self.interp_shape = interp_shape
        self.n_iters = n_iters
        self.num_uniformly_sampled_pts = num_uniformly_sampled_pts
        self.model = cotracker_model
        self.local_extent = local_extent
        self.model.eval()

    def _support_queries(self, video, query_xy=None):
        """Extra queries appended after the evaluated ones: local grid around the query (single-point mode only), the
        global grid, uniformly sampled points -- in the reference's order (evaluation_predictor.py:151-199 / :88-112)."""
        dev = video.device
        extra = []
        if query_xy is not None and self.local_grid_size > 0:
            g = get_points_on_a_grid(self.local_grid_size, (self.local_extent, self.local_extent),
                                     [query_xy[1], query_xy[0]], device=dev)
            extra.append(torch.cat([torch.zeros_like(g[:, :, :1]), g], dim=2))
        if self.grid_size > 0:
            g = get_points_on_a_grid(self.grid_size, video.shape[3:], device=dev)
            extra.append(torch.cat([torch.zeros_like(g[:, :, :1]), g], dim=2))
        if self.num_uniformly_sampled_pts > 0:
            extra.append(get_uniformly_sampled_pts(self.num_uniformly_sampled_pts, video.shape[1], video.shape[3:], device=dev))
        return extra

    @torch.no_grad()
    def forward(self, video, queries):
        queries = queries.clone().float()
        B, T, C, H, W = video.shape
        assert queries.shape[2] == 3 and B == 1
        ih, iw = self.interp_shape
        video = F.interpolate(video.reshape(B * T, C, H, W).float(), (ih, iw), mode="bilinear", align_corners=True)
        video = video.reshape(B, T, 3, ih, iw)
        queries[:, :, 1] *= (iw - 1) / (W - 1)
        queries[:, :, 2] *= (ih - 1) / (H - 1)
        N = queries.shape[1]
        if self.single_point:  # one model call per query: the query, its local grid, the global grid
            traj = torch.zeros(B, T, N, 2, device=video.device)
            vis = torch.zeros(B, T, N, device=video.device)
            conf = torch.zeros(B, T, N, device=video.device)
            for i in range(N):
                q = queries[:, i:i + 1]
                q = torch.cat([q] + self._support_queries(video, (float(q[0, 0, 1]), float(q[0, 0, 2]))), dim=1)
                out = self.model(video=video, queries=q, iters=self.n_iters)
                traj[:, :, i] = out[0][:, :, 0, :2]
                vis[:, :, i] = out[1][:, :, 0]
                conf[:, :, i] = out[2][:, :, 0] if len(out) > 3 else 1.0
            conf = conf if len(out) > 3 else None
        else:  # all queries jointly, plus the global grid / random points
            extra = self._support_queries(video)
            n_extra = sum(e.shape[1] for e in extra)
            out = self.model(video=video, queries=torch.cat([queries] + extra, dim=1), iters=self.n_iters)
            traj, vis = out[0][:, :, :N].clone() if n_extra else out[0], out[1][:, :, :N] if n_extra else out[1]
            conf = (out[2][:, :, :N] if n_extra else out[2]) if len(out) > 3 else None
        traj = traj * traj.new_tensor([(W - 1) / float(iw - 1), (H - 1) / float(ih - 1)])
        if conf is not None:
            vis = vis * conf
        return traj, vis


def compute_tapvid_metrics(query_points: np.ndarray, gt_occluded: np.ndarray, gt_tracks: np.ndarray,
                           pred_occluded: np.ndarray, pred_tracks: np.ndarray, query_mode: str) -> Mapping[str, np.ndarray]:
    """TAP-Vid metrics per video (eval_utils.py:12-138).  query_points [b,n,3] = (t, y, x); gt_occluded / pred_occluded
    [b,n,t] bool; gt_tracks / pred_tracks [b,n,t,2] = (x, y) in 256x256 raster units.  query_mode "first": only frames
    AFTER the query frame count; "strided": every frame but the query frame.  Returns occlusion_accuracy,
    pts_within_{1,2,4,8,16}, jaccard_{1,2,4,8,16}, average_pts_within_thresh, average_jaccard (arrays of length b)."""
    if query_mode not in ("first", "strided"):
        raise ValueError("Unknown query mode " + query_mode)
    T = gt_tracks.shape[2]
    frames = np.arange(T)
    qf = np.round(query_points[..., 0]).astype(np.int32)[..., None]     # [b,n,1]
    counted = (frames > qf) if query_mode == "first" else (frames != qf)  # [b,n,t]
    total = lambda m: np.sum(m & counted, axis=(1, 2))  # noqa: E731
    # the reference divides by the evaluated points of the WHOLE batch here (eval_utils.py:75-78), not per video
    out = {"occlusion_accuracy": total(pred_occluded == gt_occluded) / np.sum(counted)}
    gt_vis, pred_vis = ~gt_occluded.astype(bool), ~pred_occluded.astype(bool)
    d2 = np.sum(np.square(pred_tracks - gt_tracks), axis=-1)
    n_gt = total(gt_vis)
    within_all, jac_all = [], []
    for thr in (1, 2, 4, 8, 16):
        close = d2 < thr * thr
        hit = close & gt_vis
        out[f"pts_within_{thr}"] = total(hit) / n_gt
        false_pos = total(pred_vis & ~hit)  # predicted visible where the truth is occluded, or too far from it
        out[f"jaccard_{thr}"] = total(hit & pred_vis) / (n_gt + false_pos)
        within_all.append(out[f"pts_within_{thr}"])
        jac_all.append(out[f"jaccard_{thr}"])
    out["average_jaccard"] = np.mean(np.stack(jac_all, axis=1), axis=1)
    out["average_pts_within_thresh"] = np.mean(np.stack(within_all, axis=1), axis=1)
    return out
