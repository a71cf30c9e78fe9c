#!/usr/bin/env python3
"""
What do the per-launch HIP events of a timed call cost?  C3 workload resident in HBM, 200 calls with upx_plan_enable_timing on and
off, three times (MI355X: 1.449 vs 1.425 ms per call = 24 us for the ten events of a step).  bench.py therefore records them on
every 4th step only.

    python scripts/timing_cost_check.py
"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import upmix_amd as ux
sr, n = 48000, 48000 * 600
bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, sr, max_block_size=8192, verbose=False)
plan = ux.DevicePlan(bands, device=0)
x = np.random.default_rng(2).standard_normal((n, 2), dtype=np.float32) * 0.25
d_in = plan.alloc(n * 8); d_out = [plan.alloc(n * 4) for _ in range(3)]
plan.h2d(d_in, x); plan.sync()
def run(k):
    t0 = time.perf_counter()
    for _ in range(k):
        plan.process_device(d_in, n, n, d_out[0], d_out[1], d_out[2], n)
    plan.sync()
    return (time.perf_counter() - t0) / k * 1e3
run(150)
for rep in range(3):
    for timing in (True, False):
        plan.enable_timing(timing)
        run(20)
        print("timing", timing, "%.4f ms/step" % run(200), flush=True)
plan.close()
