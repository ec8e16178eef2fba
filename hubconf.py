"""torch.hub entry points with the reference's names and kwargs (hubconf.py:72-119).

There is no network in the build environment, so ``pretrained=True`` loads from a local file:
``checkpoint=<path>`` kwarg, else ``./checkpoints/scaled_{offline,online}.pth`` (the paths
predictor.py:17,215 default to) for CoTracker3 and ``./checkpoints/cotracker2{,v1}.pth`` for the CoTracker2
entry points (window 8 / 16, hubconf.py:27-45).
"""
import os

dependencies = ["torch"]


def _make(*, pretrained=True, online=False, version="3", checkpoint=None, **kwargs):
    from cotracker_amd.predictor import CoTrackerOnlinePredictor, CoTrackerPredictor

    if version not in ("2", "2.1", "3"):
        raise Exception("Provided version does not exist")
    v2 = version != "3"
    if pretrained and checkpoint is None:
        if v2:
            checkpoint = "./checkpoints/cotracker2.pth" if version == "2" else "./checkpoints/cotracker2v1.pth"
        else:
            checkpoint = "./checkpoints/scaled_online.pth" if online else "./checkpoints/scaled_offline.pth"
        if not os.path.exists(checkpoint):
            raise FileNotFoundError(f"pretrained=True needs {checkpoint} (no network access to download it)")
    if not pretrained:
        checkpoint = None
    window_len = {"2": 8, "2.1": 16}.get(version, 16 if online else 60)  # hubconf.py:27-45
    cls = CoTrackerOnlinePredictor if online else CoTrackerPredictor
    return cls(checkpoint=checkpoint, window_len=window_len, v2=v2)


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
