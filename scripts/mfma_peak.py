#!/usr/bin/env python
"""Sustained f32 MFMA rate of this MI355X (lr_mfma_f32_probe): the ceiling the MFMA kernels' fractions should be
read against besides the 157.3 TFLOP/s spec figure (2.4 GHz).  usage: python scripts/mfma_peak.py"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from librecommender_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
out = torch.zeros(4, device=dev)
for wps in (1, 2):
    for iters in (20_000, 100_000):
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops._call("lr_mfma_f32_probe", iters, wps, out.data_ptr(), ops._stream())
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        ms = min(ts)
        n_mfma = 256 * wps * 4 * iters * 8                       # workgroups x waves x MFMAs per wave
        tf = n_mfma * 2.0 * 32 * 32 * 2 / ms / 1e9
        clk = n_mfma * 64 / 1024 / (ms * 1e-3) / 1e9             # 64 issue cycles per MFMA on each of 1,024 SIMDs
        print(f"waves/SIMD {wps}  iters {iters:6d}  {ms:8.3f} ms  {tf:6.1f} TFLOP/s ({tf / 157.3 * 100:.1f}% of 157.3)  "
              f"implied clock {clk:.2f} GHz")
