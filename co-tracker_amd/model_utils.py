"""The sampling helpers of cotracker/models/core/model_utils.py on the HIP path, with the reference's signatures.

  bilinear_sampler   <- model_utils.py:191-255  (4-D and 5-D input, align_corners True / False, padding "border" / "zeros")
  sample_features4d  <- model_utils.py:258-290
  sample_features5d  <- model_utils.py:293-323
  get_points_on_a_grid is re-exported from predictor.py (model_utils.py:64-107).

All three are bit-identical to the reference on the CPU (tests/test_gpu_parity.py::test_bilinear_sampler_*), run in one kernel
(csrc/sampler.hip) on the reference's own NCHW layout, and -- like everything in this package -- have no CPU path: tensors must
be on the GPU.  The tracker's hot path does not go through these (its samplers are fused into the correlation kernels); they
are for callers that used the reference's helpers directly.
"""
import torch

from . import ops
from .predictor import get_points_on_a_grid  # noqa: F401  (same import path as the reference's model_utils)


def bilinear_sampler(input, coords, align_corners=True, padding_mode="border"):
    if not input.is_cuda:
        raise RuntimeError("cotracker_amd runs on an MI355X GPU only: move the tensors to 'cuda'. There is no CPU path.")
    return ops.bilinear_sampler(input.float(), coords.float(), align_corners=align_corners, padding_mode=padding_mode)


def sample_features4d(input, coords):
    """input [B,C,H,W], coords [B,R,2] = (x, y) -> [B,R,C]   (model_utils.py:258-290)."""
    B = input.shape[0]
    feats = bilinear_sampler(input, coords.unsqueeze(2))  # B C R 1
    return feats.permute(0, 2, 1, 3).reshape(B, -1, feats.shape[1] * feats.shape[3])


def sample_features5d(input, coords):
    """input [B,T,C,H,W], coords [B,R1,R2,3] = (t, x, y) -> [B,R1,R2,C]   (model_utils.py:293-323)."""
    B = input.shape[0]
    feats = bilinear_sampler(input.permute(0, 2, 1, 3, 4).contiguous(), coords.unsqueeze(3))  # B C R1 R2 1
    return feats.permute(0, 2, 3, 1, 4).reshape(B, feats.shape[2], feats.shape[3], feats.shape[1])
