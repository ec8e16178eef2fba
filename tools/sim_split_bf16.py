"""Numerics study (CPU, build container): what does replacing every exact-f32 Linear of the update
path by a split-bf16 contraction (a = a_hi + a_lo, 3 bf16 MFMAs: hi*hi + hi*lo + lo*hi, f32
accumulate) cost against the reference goldens?  Uses the numpy oracle as the harness (this is a
tool, not product code).  Usage: python tools/sim_split_bf16.py [terms]   terms in {1,3,6}; optional 2nd arg f16 = split into IEEE half instead of bf16.
Two-term study (3rd arg): "2a:<min_k>" drops the hi_w*lo_a term (activations rounded to one half) and "2w:<min_k>" the
lo_w*hi_a term (weights rounded to one half) in every Linear whose K (input width) is >= min_k; the others keep 3 terms."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cotracker_oracle as O  # noqa: E402
import test_oracle_golden as TG  # noqa: E402

TERMS = int(sys.argv[1]) if len(sys.argv) > 1 else 3


DT = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.bfloat16
TWO = sys.argv[3].split(":") if len(sys.argv) > 3 else None   # e.g. ["2a", "1536"]


def split(x, n):
    parts, r = [], torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    for _ in range(n):
        h = r.to(DT).float()
        parts.append(h)
        r = r - h
    return parts


def linear_split(x, w, b=None):
    sh = x.shape
    x2 = np.asarray(x, np.float32).reshape(-1, sh[-1])
    if TERMS == 1:
        xa, wa = split(x2, 1), split(w, 1)
        y = xa[0] @ wa[0].T
    elif TERMS == 3:
        xa, wa = split(x2, 2), split(w, 2)
        if TWO is not None and w.shape[1] >= int(TWO[1]) and (len(TWO) < 3 or w.shape[1] <= int(TWO[2])):
            y = (xa[0] @ wa[1].T if TWO[0] == "2a" else xa[1] @ wa[0].T) + xa[0] @ wa[0].T
        else:
            y = xa[1] @ wa[0].T + xa[0] @ wa[1].T + xa[0] @ wa[0].T
    else:
        xa, wa = split(x2, 3), split(w, 3)
        y = (xa[2] @ wa[0].T + xa[0] @ wa[2].T + xa[1] @ wa[1].T) + (xa[1] @ wa[0].T + xa[0] @ wa[1].T) + xa[0] @ wa[0].T
    y = y.numpy().reshape(*sh[:-1], w.shape[0])
    if b is not None:
        y = y + np.asarray(b, np.float32)
    return y.astype(np.float32)


def golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))


def report(tag, a, b):
    print(f"  {tag:28s} max|err| = {np.abs(a - b).max():.3e}")


def run(label):
    print(label)
    g = golden("ops")
    p = TG.oracle_params(3)
    d = O.update_former(g["uf_x"], p)
    report("update_former delta", d, g["uf_delta"])
    pyr = [g[f"fmaps{i}"] for i in range(4)]
    sup = [g[f"support{i}"] for i in range(4)]
    B, S, N, _ = g["coords"].shape
    cinit = np.broadcast_to(g["queried_coords"].reshape(B, 1, N, 2), (B, S, N, 2))
    trace = []
    O.forward_window(pyr, cinit, sup, g["fw_vis_init"], g["fw_conf_init"], p, iters=3, model_resolution=(96, 128), trace=trace)
    report("forward_window coords px it2", trace[2]["coords"] * 4.0, g["fw_coords2"])
    report("forward_window vis logit it2", trace[2]["vis"][..., 0], g["fw_vis2"])
    report("forward_window conf logit it2", trace[2]["conf"][..., 0], g["fw_conf2"])
    g = golden("model_online")
    p = TG.oracle_params(1)
    fm = O.normalize_fmaps(g["on_fnet"][None])[0]
    T = fm.shape[0]
    c, v, f = O.model_forward_online(TG._pad_fmaps(fm, 8, T)[None], g["on_queries"], p, window_len=8, iters=4,
                                     model_resolution=(64, 96), T=T)
    report("online sliding coords px", c, g["on_coords"])
    report("online sliding vis logit", TG.logit(v), TG.logit(g["on_vis"]))
    report("online sliding conf logit", TG.logit(f), TG.logit(g["on_conf"]))
    g = golden("model_offline")
    p = TG.oracle_params(2)
    fm = O.normalize_fmaps(g["off_fnet"][None])
    c, v, f = O.model_forward_offline(fm, g["off_queries"], p, iters=4, model_resolution=(64, 96))
    report("offline coords px", c, g["off_coords"])
    report("offline vis logit", TG.logit(v), TG.logit(g["off_vis"]))
    report("offline conf logit", TG.logit(f), TG.logit(g["off_conf"]))


run("exact f32 oracle vs reference goldens")
O.linear = linear_split
run(f"split-bf16 Linear layers, {TERMS} MFMA terms, vs reference goldens")
