#!/usr/bin/env python3
"""Block-at-a-time rate of MultiBandExtractorAccu.process_stereo_chunk (upx_stream_chunk: overlap-add ring on the device):
microseconds per block and x real time at 48 kHz for a few STFT sizes."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upmix_amd as ux  # noqa: E402

rng = np.random.default_rng(0)
for n in (256, 1024, 4096, 8192):
    hop = n // 4
    bex = ux.MultiBandExtractorAccu(n, 0.75, ux.make_blackman_harris, 300.0, 3000.0, 48000, "raised_cosine", 75.0, 750.0)
    l = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal(n).astype(np.float32)
    for _ in range(20):
        bex.process_stereo_chunk(l, r)
    reps = 500
    t0 = time.perf_counter()
    for _ in range(reps):
        bex.process_stereo_chunk(l, r)
    dt = (time.perf_counter() - t0) / reps
    print(f"N={n:5d} hop={hop:5d}: {dt * 1e6:7.1f} us per block = {hop / 48000 / dt:8.1f} x real time at 48 kHz", flush=True)
    bex.close()
