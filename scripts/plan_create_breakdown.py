#!/usr/bin/env python3
"""
Where a fresh process' `runtime_and_plan_ms` goes (bench.py: e2e.drop_in_entry_float64_views.fresh_process): dlopen of the
library, the HIP runtime coming up, upx_plan_create (its own breakdown on stderr: UPX_PLAN_TIMING=1 under UPX_TUNING=1),
upx_plan_reserve, and a second plan of the same bands in the same process (what is per process, what per plan).

    python3 scripts/plan_create_breakdown.py [--max-stft 8192]
"""
import argparse
import os
import sys
import time

os.environ["UPX_TUNING"] = "1"
os.environ["UPX_PLAN_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-stft", type=int, default=8192)
    args = ap.parse_args()
    t0 = time.perf_counter()
    import upmix_amd as ux
    from upmix_amd import _lib
    t1 = time.perf_counter()
    _lib.load()
    t2 = time.perf_counter()
    n = _lib.device_count()
    t3 = time.perf_counter()
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 48000, max_block_size=args.max_stft,
                           verbose=False)
    t4 = time.perf_counter()
    plan = ux.DevicePlan(bands)
    t5 = time.perf_counter()
    plan.reserve(28_800_000, 28_800_000, 28_800_000)
    t6 = time.perf_counter()
    plan2 = ux.DevicePlan(bands)
    t7 = time.perf_counter()
    print(f"import {1e3*(t1-t0):.1f} ms, dlopen {1e3*(t2-t1):.1f}, device_count (runtime up) {1e3*(t3-t2):.1f} [{n} device(s)], "
          f"chain_bands {1e3*(t4-t3):.1f}, DevicePlan {1e3*(t5-t4):.1f}, reserve(10 min) {1e3*(t6-t5):.1f}, "
          f"second DevicePlan {1e3*(t7-t6):.1f}", flush=True)
    plan.close()
    plan2.close()


if __name__ == "__main__":
    main()
