"""Deterministic synthetic checkpoints (no network, no released weights on disk).

``fill_synthetic_`` writes a reproducible pseudo-random value into every entry of
a module's ``state_dict`` using only the entry's *name* and *shape*, so the same
call produces identical weights for the reference model (golden generation,
tests/golden/make_golden.py) and for this package's model (same key set,
SURVEY §8b).  Distributions follow the reference initialisers
(cotracker.py:464-481 xavier-uniform linears, heads trunc-normal std 1e-3;
blocks.py:173-180 kaiming fan-out convs) except that biases are non-zero so the
bias paths are exercised, and the two output heads are scaled ("stress init",
BASELINE.md §3) so that tracks move several pixels over 6 iterations.
"""
import zlib

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2**32))


def synthetic_tensor(name: str, shape, seed: int = 0, head_scale: float = 8.0) -> torch.Tensor:
    r = _rng(name, seed)
    shape = tuple(shape)
    is_head = ".flow_head." in name or ".vis_conf_head." in name
    if name.endswith("virual_tracks"):
        v = r.standard_normal(shape)
    elif name.endswith("norm_context.weight"):
        v = 1.0 + 0.1 * r.standard_normal(shape)
    elif name.endswith("norm_context.bias"):
        v = 0.05 * r.standard_normal(shape)
    elif name.endswith(".weight") and len(shape) == 4:  # conv, kaiming normal fan_out
        fan_out = shape[0] * shape[2] * shape[3]
        v = r.standard_normal(shape) * np.sqrt(2.0 / fan_out)
    elif name.endswith(".weight") and len(shape) == 2:
        if is_head:
            v = r.standard_normal(shape) * 0.001 * head_scale
        else:
            a = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = r.uniform(-a, a, size=shape)
    elif name.endswith(".weight") and len(shape) == 1:  # affine norm scale (CoTracker2's GroupNorm, cotracker.py:79)
        v = 1.0 + 0.1 * r.standard_normal(shape)
    elif name.endswith(".bias"):
        v = r.standard_normal(shape) * (0.01 if is_head else 0.02)
    else:
        raise KeyError(f"no synthetic rule for {name} {shape}")
    return torch.from_numpy(np.asarray(v, dtype=np.float32))


@torch.no_grad()
def fill_synthetic_(module: torch.nn.Module, seed: int = 0, head_scale: float = 8.0):
    """In-place synthetic checkpoint for any module with the CoTracker3 key set."""
    sd = module.state_dict()
    for name, t in sd.items():
        if name in ("time_emb", "pos_emb"):  # deterministic buffers (embeddings.py:11-84), keep
            continue
        t.copy_(synthetic_tensor(name, t.shape, seed, head_scale).to(t.device, t.dtype))
    inval = getattr(module, "invalidate_packed_weights", None)
    if inval is not None:  # cotracker_amd models cache device-side repacked weights
        inval()
    return module


# CoTracker2 at BASELINE scale (tests/golden/make_golden_scale.py "v2_c2", tests/test_gpu_scale.py): with xavier-random weights
# the CoTracker2 update (coordinates AND track features fed back through 6 + 6 transformer layers, cotracker.py:131-171) is a
# chaotic map -- the unmodified reference moves 0.76 px between 8 and 3 CPU threads at the CoTracker3 head scale and 0.08 px at
# 0.25 x -- so six iterations can only be pinned with a damped feedback: flow-head scale 0.02 and track_feat_updater x 0.1
# (reference 8 vs 3 threads: ~1e-4 px; the tracks still move up to ~3 px per window).
V2_DAMP = {"head_scale": 0.02, "updater_scale": 0.1}


@torch.no_grad()
def fill_synthetic_v2_damped_(module: torch.nn.Module, seed: int = 0):
    """fill_synthetic_ for a CoTracker2 (reference or cotracker_amd) with the V2_DAMP feedback damping."""
    fill_synthetic_(module, seed=seed, head_scale=V2_DAMP["head_scale"])
    module.track_feat_updater[0].weight.mul_(V2_DAMP["updater_scale"])
    module.track_feat_updater[0].bias.mul_(V2_DAMP["updater_scale"])
    inval = getattr(module, "invalidate_packed_weights", None)
    if inval is not None:
        inval()
    return module
