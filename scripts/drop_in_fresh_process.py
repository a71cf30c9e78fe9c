#!/usr/bin/env python3
"""
What ONE call of the documented drop-in entry costs a fresh process (the reference's flow: main.py:23, 67-80 - import,
chain_bands, one extract_center_left_right_multi_band_in_memory call on two float64 column views of the decoded file).
Run as a child of bench.py (e2e.drop_in_entry_float64_views.fresh_process) or by hand; prints one JSON line.

    python3 scripts/drop_in_fresh_process.py [--seconds 600] [--sr 48000] [--max-stft 8192]
"""
import argparse
import json
import os
import sys
import time

t_start = time.perf_counter()
import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--sr", type=int, default=48000)
    ap.add_argument("--max-stft", type=int, default=8192)
    args = ap.parse_args()
    total = int(args.sr * args.seconds)
    rng = np.random.default_rng(2)
    wave = np.empty((total, 2), dtype=np.float64)      # what soundfile.read returns (main.py:43)
    m, s = rng.standard_normal(total), rng.standard_normal(total)
    wave[:, 0] = 0.1 * (m + 0.5 * s)
    wave[:, 1] = 0.1 * (m - 0.5 * s)
    del m, s
    t0 = time.perf_counter()
    import upmix_amd.center_extraction as ce          # INTEGRATION.md option A: the import main.py:23 is swapped for
    t_import = time.perf_counter() - t0
    t0 = time.perf_counter()
    bands = ce.chain_bands([0, 30, 120, 480, 1920, 7680], overlap=0.75, window_func=ce.make_blackman_harris, sr=args.sr,
                           xover_mode="raised_cosine", max_block_size=args.max_stft, verbose=False)
    t_chain = time.perf_counter() - t0
    L, R = wave[:, 0], wave[:, 1]                       # main.py:49-50
    # what the first call spends before any sample moves: the HIP runtime comes up (first runtime call of the process), the
    # library's code objects load, the plan's tables and device buffers are made - timed apart by creating the plan the entry
    # is about to look up in its cache (extractor._checked_out_plan)
    from upmix_amd import extractor
    t0 = time.perf_counter()
    with extractor._checked_out_plan(bands, 0):
        pass
    t_plan = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = ce.extract_center_left_right_multi_band_in_memory(L, R, args.sr, bands)
    t_first = time.perf_counter() - t0
    keep = [a.copy() for a in res]
    from upmix_amd import hostmem
    pageable = not any(hostmem.is_pinned(a) for a in res)
    del res                                              # a caller that is done with a result lets it go: its blocks are reused
    t0 = time.perf_counter()
    res = ce.extract_center_left_right_multi_band_in_memory(L, R, args.sr, bands)
    t_second = time.perf_counter() - t0
    same = all(np.array_equal(a, b) for a, b in zip(res, keep))
    del res
    steady = []
    for _ in range(3):
        t0 = time.perf_counter()
        res = ce.extract_center_left_right_multi_band_in_memory(L, R, args.sr, bands)
        steady.append(time.perf_counter() - t0)
        del res
    # the host cast + interleave this entry used to make before the first byte moved (round 4: extractor.py:521)
    t0 = time.perf_counter()
    np.stack([np.asarray(L, dtype=np.float32), np.asarray(R, dtype=np.float32)], axis=1)
    t_stack = time.perf_counter() - t0
    print(json.dumps({
        "samples": total, "import_ms": round(t_import * 1e3, 1), "chain_bands_ms": round(t_chain * 1e3, 1),
        "runtime_and_plan_ms": round(t_plan * 1e3, 1),
        "first_call_ms": round(t_first * 1e3, 1), "first_call_results_pageable": bool(pageable),
        "second_call_ms": round(t_second * 1e3, 1), "steady_call_ms": round(min(steady) * 1e3, 1),
        "one_shot_total_ms": round((t_import + t_chain + t_plan + t_first) * 1e3, 1),
        "host_cast_and_interleave_it_replaces_ms": round(t_stack * 1e3, 1),
        "identical_results": bool(same),
        "note": "runtime_and_plan = HIP runtime start + code objects + plan tables / device buffers (once per process and band "
                "list); first call -> pageable result arrays (a one-shot process pays no pinning); second call pins its result "
                "blocks; steady = pooled page-locked results, each result dropped before the next call"}))


if __name__ == "__main__":
    main()
