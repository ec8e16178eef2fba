"""torch.hub entry points with the reference's names and kwargs (hubconf.py:72-119).

There is no network in the build environment, so ``pretrained=True`` loads from a local file:
``checkpoint=<path>`` kwarg, else ``./checkpoints/scaled_{offline,online}.pth`` (the paths
predictor.py:17,215 default to).  CoTracker2 entry points raise NotImplementedError (out of the
hot-path scope).
"""
import os

dependencies = ["torch"]


def _make(*, pretrained=True, online=False, version="3", checkpoint=None, **kwargs):
    from cotracker_amd.predictor import CoTrackerOnlinePredictor, CoTrackerPredictor

    if version != "3":
        raise NotImplementedError("only the CoTracker3 entry points are provided by the MI355X hot path")
    if pretrained and checkpoint is None:
        checkpoint = "./checkpoints/scaled_online.pth" if online else "./checkpoints/scaled_offline.pth"
        if not os.path.exists(checkpoint):
            raise FileNotFoundError(f"pretrained=True needs {checkpoint} (no network access to download it)")
    if not pretrained:
        checkpoint = None
    if online:
        return CoTrackerOnlinePredictor(checkpoint=checkpoint, window_len=16)
    return CoTrackerPredictor(checkpoint=checkpoint, window_len=60)


def cotracker3_offline(*, pretrained: bool = True, **kwargs):
    return _make(pretrained=pretrained, online=False, version="3", **kwargs)


def cotracker3_online(*, pretrained: bool = True, **kwargs):
    return _make(pretrained=pretrained, online=True, version="3", **kwargs)


def cotracker2(*, pretrained: bool = True, **kwargs):
    return _make(pretrained=pretrained, online=False, version="2", **kwargs)


def cotracker2_online(*, pretrained: bool = True, **kwargs):
    return _make(pretrained=pretrained, online=True, version="2", **kwargs)


def cotracker2v1(*, pretrained: bool = True, **kwargs):
    return _make(pretrained=pretrained, online=False, version="2.1", **kwargs)


def cotracker2v1_online(*, pretrained: bool = True, **kwargs):
    return _make(pretrained=pretrained, online=True, version="2.1", **kwargs)
