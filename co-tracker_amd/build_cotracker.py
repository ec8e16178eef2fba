"""Model factory with the reference's signature (cotracker/models/build_cotracker.py:26-45)."""
import torch

from .model import CoTrackerThreeOffline, CoTrackerThreeOnline
from .model_v2 import CoTracker2


def build_cotracker(checkpoint=None, offline=True, window_len=16, v2=False):
    if v2:
        model = CoTracker2(stride=4, window_len=window_len)
    else:
        cls = CoTrackerThreeOffline if offline else CoTrackerThreeOnline
        model = cls(stride=4, corr_radius=3, window_len=window_len)
    if checkpoint is not None:
        with open(checkpoint, "rb") as f:
            state_dict = torch.load(f, map_location="cpu")
        if "model" in state_dict:
            state_dict = state_dict["model"]
        model.load_state_dict(state_dict)
    return model
