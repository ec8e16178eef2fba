#!/usr/bin/env python
"""Sustained MFMA peak of this MI355X under its power budget (register-only loops)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cotracker_amd import _lib as L  # noqa: E402

lib = L.load()
scratch = torch.zeros(16, device="cuda")
for kind, name, iters in ((0, "v_mfma_f32_32x32x2_f32", 20000), (1, "v_mfma_f32_32x32x16_bf16", 100000)):
    for rep in range(3):
        fl = C.c_double(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.ctk_probe_mfma(kind, iters, scratch.data_ptr(), C.byref(fl), torch.cuda.current_stream().cuda_stream), "probe")
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        print(f"{name}: {ms:.2f} ms  {fl.value / ms / 1e9:.1f} TFLOP/s")
