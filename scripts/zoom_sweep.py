#!/usr/bin/env python3
"""Developer sweep on a GPU box: per-band times of a plan under different UPX_* settings (one process each).
Usage: python scripts/zoom_sweep.py "<ENV=VAL ENV=VAL>" ...   (each argument = one configuration; "" = defaults)
       UPX_SWEEP_PLAN=c3|c4|default selects the workload."""
import os
os.environ.setdefault("UPX_TUNING", "1")   # round 6: the library reads its UPX_* knobs only in a process that opts in
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import upmix_amd as ux
which = os.environ.get("UPX_SWEEP_PLAN", "c3")
sr, mx, secs = {"c3": (48000, 8192, 600), "c4": (96000, 8192, 900), "default": (48000, 65536, 600)}[which]
total = sr * secs
bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, sr, max_block_size=mx, verbose=False)
plan = ux.DevicePlan(bands)
rng = np.random.default_rng(2)
x = (0.1 * rng.standard_normal((total, 2))).astype(np.float32)
d_in = plan.alloc(total * 8); d_out = [plan.alloc(total * 4) for _ in range(3)]
plan.h2d(d_in, x)
plan.enable_timing(True)
acc = []
for r in range(6):
    plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
    plan.sync()
    if r: acc.append(plan.band_times_ms())
ms = np.median(np.array(acc), axis=0)
names = [plan.band_kernel_name(i).split("upx_")[-1][:40] for i in range(len(bands))]
print("  bands", np.round(ms, 3).tolist(), "sum %%.3f ms" %% ms.sum(), [plan.band_info(b)["blocks_per_stream"] for b in range(len(bands))], flush=True)
''' % ROOT

for cfg in sys.argv[1:] or [""]:
    env = dict(os.environ)
    for kv in cfg.split():
        k, v = kv.split("=", 1)
        env[k] = v
    print(f"[{cfg or 'defaults'}]", flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
