#!/usr/bin/env python3
"""How the WAV pipeline's uploads and kernels share the card: C3's signal as PCM16 from page-locked memory through
upx_wav_pipeline, a few times; prints wall ms and the library's begin / tail / finish split.  Run under
`rocprofv3 --kernel-trace --stats` to see whether the runtime copies with blit kernels (__amd_rocclr_copyBuffer...)
or with the SDMA engines, and under HSA_ENABLE_SDMA=0 / 1 to compare."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench            # noqa: E402
import upmix_amd as ux  # noqa: E402

sr, seconds = 48000, float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
n = int(sr * seconds)
bands = ux.chain_bands(bench.EDGES, 0.75, ux.make_blackman_harris, sr, max_block_size=8192, verbose=False)
plan = ux.DevicePlan(bands)
x = bench.synth(n, 2)
pcm = np.clip(np.rint(x * 32767.0), -32768, 32767).astype("<i2")
pinned = plan.host_empty(pcm.nbytes).view("<i2").reshape(pcm.shape)
pinned[...] = pcm
for rep in range(4):
    t0 = time.perf_counter()
    plan.wav_pipeline(pinned, 16, 2, n, "stereo_sum", 16)
    dt = (time.perf_counter() - t0) * 1e3
    print(f"rep {rep}: {dt:.2f} ms  {plan.wav_pipeline_times_ms()}", flush=True)
plan.close()
