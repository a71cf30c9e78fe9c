#!/usr/bin/env python3
"""
Round 5, lever (c): do launches of DIFFERENT kinds run better side by side than one after another?

The C3 signal is cut into two time tiles on the shard grid; each tile has its own plan (own HIP stream, own scratch and
seam buffers) whose launch geometry is cut for `n_cu` compute units (UPX_N_CU) and whose launch groups start `rotate`
groups into the list (UPX_BAND_ROTATE), so that while tile 0 runs a band-limited (LDS-bound) launch tile 1 runs a fused
(VALU-bound) one.  Timing only: the two tiles write their own planes and the seam between them is not added.

    python3 scripts/r5_two_tiles.py [--steps 40] [--seconds 600]

Prints one line per configuration: ms per step of the whole signal (both tiles), and the single-plan reference.
"""
import argparse
import json
import os
os.environ.setdefault("UPX_TUNING", "1")   # round 6: the library reads its UPX_* knobs only in a process that opts in
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench   # noqa: E402  (synth, EDGES)


def make_plan(ux, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        bands = ux.chain_bands(bench.EDGES, 0.75, ux.make_blackman_harris, 48000, max_block_size=8192, threshold_factor=32,
                               verbose=False, device=0)
        return ux.DevicePlan(bands, device=0), bands
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--seconds", type=float, default=600.0)
    args = ap.parse_args()
    import upmix_amd as ux
    from upmix_amd import sharding

    total = int(48000 * args.seconds)
    x = bench.synth(total, 2)
    base, bands = make_plan(ux, {})
    d_in = base.alloc(total * 8)
    base.h2d(d_in, x)
    del x
    geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
    d_whole = [base.alloc(total * 4) for _ in range(3)]

    def timed(step, sync, steps):
        for _ in range(60):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        return (time.perf_counter() - t0) / steps * 1e3

    ref = timed(lambda: base.process_device(d_in, total, total, *d_whole, total), base.sync, args.steps)
    print(json.dumps({"config": "one plan, whole signal", "ms_per_step": round(ref, 4)}), flush=True)

    shards = geo.plan(total, 2)
    d_tile = [[base.alloc(s.t_out * 4) for _ in range(3)] for s in shards]
    configs = [(256, 0, 3), (256, 2, 3), (128, 0, 3), (128, 2, 3), (128, 1, 3), (128, 3, 3), (128, 2, 0), (160, 2, 3),
               (192, 2, 3), (96, 2, 3)]
    for n_cu, rot, prio in configs:
        pa, _ = make_plan(ux, {"UPX_N_CU": n_cu, "UPX_PRIO_YOUNG": prio})
        pb, _ = make_plan(ux, {"UPX_N_CU": n_cu, "UPX_BAND_ROTATE": rot, "UPX_PRIO_YOUNG": prio})
        plans = [pa, pb]

        def step():
            for pl, s, d in zip(plans, shards, d_tile):
                pl.process_device(d_in + s.start * 8, s.t_in, s.own_len, d[0], d[1], d[2], s.t_out)

        def sync():
            pa.sync()
            pb.sync()
        ms = timed(step, sync, args.steps)
        # the same two calls on ONE stream (plan A for both tiles): what the cut alone costs
        one = timed(lambda: [pa.process_device(d_in + s.start * 8, s.t_in, s.own_len, d[0], d[1], d[2], s.t_out)
                             for s, d in zip(shards, d_tile)], pa.sync, args.steps)
        print(json.dumps({"config": {"n_cu": n_cu, "rotate_tile1": rot, "prio_young": prio},
                          "two_streams_ms_per_step": round(ms, 4), "same_cut_one_stream_ms": round(one, 4),
                          "vs_one_plan": round(ms / ref, 4)}), flush=True)
        pa.close()
        pb.close()
    again = timed(lambda: base.process_device(d_in, total, total, *d_whole, total), base.sync, args.steps)
    print(json.dumps({"config": "one plan, whole signal (again)", "ms_per_step": round(again, 4)}), flush=True)


if __name__ == "__main__":
    main()
