#!/usr/bin/env python3
"""
How long after the first launch does a step reach its steady time?  Prints the wall time of consecutive groups of
steps of the C3 workload (one process_device call each, inputs resident) from a cold process, then the same after an
idle pause.  The card leaves its idle power state over the first tens of milliseconds of work.

    python scripts/clock_ramp_check.py [group=5] [groups=60] [pause_s=2.0]
"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import upmix_amd as ux

group = int(sys.argv[1]) if len(sys.argv) > 1 else 5
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 60
pause = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
sr, n = 48000, 48000 * 600
bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, sr, max_block_size=8192, verbose=False)
plan = ux.DevicePlan(bands, device=0)
rng = np.random.default_rng(2)
x = rng.standard_normal((n, 2), dtype=np.float32) * 0.25
d_in = plan.alloc(n * 8)
d_out = [plan.alloc(n * 4) for _ in range(3)]
plan.h2d(d_in, x)
plan.sync()
for label in ("cold", "after %.1f s idle" % pause):
    out = []
    for g in range(groups):
        t0 = time.perf_counter()
        for _ in range(group):
            plan.process_device(d_in, n, n, d_out[0], d_out[1], d_out[2], n)
        plan.sync()
        out.append((time.perf_counter() - t0) / group * 1e3)
    print(label, "ms/step per group of %d:" % group, " ".join("%.3f" % v for v in out), flush=True)
    time.sleep(pause)
plan.close()
