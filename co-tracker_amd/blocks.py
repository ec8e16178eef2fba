"""Host-side mirror of cotracker/models/core/cotracker/blocks.py:284-362 -- ``CorrBlock`` (CoTracker2's 4-D
correlation-volume sampler) on the MI355X path.

Same constructor kwargs and the same two-step protocol as the reference (``corr(targets)`` then
``sample(coords)`` -> ``[B*N, S, num_levels*(2r+1)^2]``), but ``corr`` only remembers the track features and
``sample`` runs ONE fused HIP kernel (``ctk_corrblock_sample``) that forms the <=9x9 footprint dot products under
every 7x7 tap lattice and blends them exactly like ATen's ``grid_sampler_2d``: the [B,S,N,H,W] volumes of
blocks.py:357-362 are never materialised.  GPU only, no fallback.
"""
import torch

from . import ops


class CorrBlock:
    def __init__(self, fmaps, num_levels=4, radius=4, multiple_track_feats=False, padding_mode="zeros"):
        if num_levels != 4 or radius != 3 or multiple_track_feats or padding_mode != "border":
            raise NotImplementedError("HIP CorrBlock is specialised to num_levels=4, radius=3, padding_mode='border', "
                                      "single track feature (how CoTracker2 builds it, cotracker.py:119-124)")
        if not fmaps.is_cuda:
            raise RuntimeError("cotracker_amd runs on an MI355X GPU only: there is no CPU path")
        B, S, C, H, W = fmaps.shape
        if C != 128:
            raise NotImplementedError("latent_dim must be 128 (cotracker.py:44)")
        self.S, self.C, self.H, self.W = S, C, H, W
        self.padding_mode = padding_mode
        self.num_levels = num_levels
        self.radius = radius
        self.multiple_track_feats = multiple_track_feats
        # NHWC pyramid per batch element: level 0 = a layout change of fmaps, level l = 2x2 average pooling
        # (blocks.py:300-307; ctk_avg_pool2_nhwc is bit-identical to F.avg_pool2d).  Not normalised.
        self.pyramids = []
        for b in range(B):
            f0 = fmaps[b].float().permute(0, 2, 3, 1).contiguous()
            self.pyramids.append(ops.build_pyramid(f0, num_levels))
        self.targets = None

    def corr(self, targets):  # blocks.py:342-362
        B, S, N, C = targets.shape
        assert C == self.C
        assert S == self.S
        self.targets = targets.float().contiguous()

    def sample(self, coords):  # blocks.py:309-340
        B, S, N, D = coords.shape
        assert D == 2
        assert self.targets is not None and self.targets.shape[:3] == (B, S, N), "call corr(targets) first"
        coords = coords.float().contiguous()
        outs = [ops.corrblock_sample(self.pyramids[b], self.targets[b], coords[b]) for b in range(B)]
        return outs[0] if B == 1 else torch.cat(outs, dim=0)  # [B*N, S, LRR]
