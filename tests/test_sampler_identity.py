"""Replay of the reference's only unit test (tests/test_bilinear_sample.py:16-47, align_corners=True
cases): identity sampling on integer grids must reproduce the input -- against the oracle on CPU and,
under -m gpu, against the HIP sampler."""
import numpy as np
import pytest

from oracle import cotracker_oracle as O


def _identity_case(T, H, W, seed):
    r = np.random.RandomState(seed)
    base = r.standard_normal((H, W)).astype(np.float32)
    vol = np.stack([base + np.float32(k) for k in range(T)], axis=0)  # [T,H,W]
    tt, xx, yy = np.meshgrid(np.arange(T), np.arange(W), np.arange(H), indexing="ij")
    coords = np.stack([tt, xx, yy], axis=-1).astype(np.float32).transpose(0, 2, 1, 3)  # [T,H,W,3] (t,x,y)
    return vol, coords


@pytest.mark.parametrize("T,H,W", [(1, 4, 5), (3, 4, 5)])
def test_identity_oracle(T, H, W):
    vol, coords = _identity_case(T, H, W, 0)
    out = O.bilinear_sampler_5d(vol[None, None], coords[None])
    np.testing.assert_allclose(out[0, 0], vol, rtol=1.3e-6, atol=1e-5)  # torch.testing.assert_close defaults


@pytest.mark.gpu
def test_identity_hip():
    import torch
    from cotracker_amd import ops
    H, W, S = 4, 5, 3
    r = np.random.RandomState(1)
    fm = r.standard_normal((S, H, W, 128)).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pts = np.stack([xx, yy], -1).reshape(-1, 2).astype(np.float32)  # N = H*W points on the pixel grid
    coords = np.broadcast_to(pts[None], (S, H * W, 2)).copy()
    out = ops.sample_patches(torch.from_numpy(fm).cuda(), torch.from_numpy(coords).cuda(), 0).cpu().numpy()
    centre = out[:, :, 24, :].reshape(S, H, W, 128)  # tap (dx=0, dy=0)
    np.testing.assert_allclose(centre, fm, rtol=1.3e-6, atol=1e-5)
