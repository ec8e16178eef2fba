"""MI355X-native CoTracker3 iterative-update hot path (gfx950 HIP kernels behind a C-ABI).

Public surface mirrors the reference (facebookresearch/co-tracker):
  cotracker_amd.predictor.CoTrackerPredictor / CoTrackerOnlinePredictor   (cotracker/predictor.py)
  cotracker_amd.build_cotracker.build_cotracker                            (cotracker/models/build_cotracker.py:26-45)
  cotracker_amd.model.CoTrackerThreeOnline / CoTrackerThreeOffline         (cotracker3_online.py / cotracker3_offline.py)
Import it as ``cotracker_amd`` (see ../cotracker_amd/__init__.py).
"""
__version__ = "0.1.0"
