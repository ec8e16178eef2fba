"""Synthetic inputs of BASELINE.md §3: a smooth-noise texture that translates by
(0.5*t, 0.25*t) px per frame plus 2 % i.i.d. noise, float32 in [0,255], [1,T,3,H,W].

Deterministic (numpy RandomState + elementwise float32 math), so the golden
generator, the tests and bench.py all see the same video for the same arguments.
"""
import numpy as np
import torch


def _smooth(tex: np.ndarray, k: int = 9, passes: int = 2) -> np.ndarray:
    """Separable box filter, edge-replicated (pure numpy: identical on every host)."""
    pad = k // 2
    for _ in range(passes):
        for axis in (1, 2):
            t = np.concatenate([np.repeat(np.take(tex, [0], axis=axis), pad, axis=axis), tex,
                                np.repeat(np.take(tex, [-1], axis=axis), pad, axis=axis)], axis=axis)
            c = np.cumsum(t, axis=axis, dtype=np.float64)
            c = np.concatenate([np.zeros_like(np.take(c, [0], axis=axis)), c], axis=axis)
            n = tex.shape[axis]
            hi = np.take(c, np.arange(k, k + n), axis=axis)
            lo = np.take(c, np.arange(0, n), axis=axis)
            tex = ((hi - lo) / k).astype(np.float32)
    return tex


def synthetic_video(T: int, H: int, W: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    r = np.random.RandomState(seed)
    m = 64
    tex = r.uniform(0.0, 1.0, size=(3, H + m, W + m)).astype(np.float32)
    tex = _smooth(tex)
    lo, hi = tex.min(), tex.max()
    tex = (tex - lo) / max(hi - lo, 1e-6)
    frames = np.empty((T, 3, H, W), dtype=np.float32)
    for t in range(T):
        sx, sy = (0.5 * t) % (m - 1), (0.25 * t) % (m - 1)
        ix, iy = int(np.floor(sx)), int(np.floor(sy))
        fx, fy = np.float32(sx - ix), np.float32(sy - iy)
        a = tex[:, iy:iy + H, ix:ix + W]
        b = tex[:, iy:iy + H, ix + 1:ix + 1 + W]
        c = tex[:, iy + 1:iy + 1 + H, ix:ix + W]
        d = tex[:, iy + 1:iy + 1 + H, ix + 1:ix + 1 + W]
        top = a * (1 - fx) + b * fx
        bot = c * (1 - fx) + d * fx
        frames[t] = top * (1 - fy) + bot * fy
    noise = r.uniform(-0.02, 0.02, size=frames.shape).astype(np.float32)
    frames = np.clip(frames + noise, 0.0, 1.0) * np.float32(255.0)
    return torch.from_numpy(frames)[None].to(device)


def grid_queries(grid_size: int, extent=(384, 512), query_frame: int = 0, device="cpu") -> torch.Tensor:
    """Same points as predictor `grid_size=G` (model_utils.py:83-139): [1,G*G,3]=(t,x,y)."""
    from .predictor import get_points_on_a_grid

    pts = get_points_on_a_grid(grid_size, extent, device=device)
    return torch.cat([torch.full_like(pts[:, :, :1], float(query_frame)), pts], dim=2)
