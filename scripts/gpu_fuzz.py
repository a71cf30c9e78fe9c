"""One-off wider randomised sweep on the GPU (not part of the test suite): many random plans, incl. same-size
neighbours (merged launches), streaming with small chunks, and overlaps that route to every kernel path.

    python scripts/gpu_fuzz.py <seed> <trials> [live]

Round 6: every tenth trial draws its length from 5-30 M samples and runs as ONE device-resident launch per band group
(upx_process_device on the whole signal - the full-chip launch geometry: stream tables, shorter edge streams, XCD dealing,
which lengths below 300 k never select); the oracle is compared on windows - the head, the tail, two centred on stream seams
read off the launch geometry, two seeded - behind its own fade-in."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import upmix_amd as ux
from oracle import upmix_oracle as orc

def rms(a): return float(np.sqrt(np.mean(np.square(a)))) if a.size else 0.0
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
sizes = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536]
overlaps = [0.5, 0.75, 0.875, 0.6, 0.7, 0.9, 0.75, 0.75]
LIVE = len(sys.argv) > 3 and sys.argv[3] == "live"   # aim at the live-slot flavours: fused sizes, hop N/4, narrow bands
if LIVE:
    sizes, overlaps = [1024, 2048, 1024, 2048, 256, 512], [0.75]
windows = sorted(ux.WINDOW_FUNCS)
worst = 0.0
t0 = time.time()
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n_bands = int(rng.integers(1, 6))
    edges = np.sort(rng.uniform(10.0, 20000.0, size=n_bands + 1))
    if LIVE:
        edges = np.sort(rng.uniform(10.0, 9000.0, size=n_bands + 1))
    overlap = overlaps[int(rng.integers(len(overlaps)))]
    wname = windows[int(rng.integers(len(windows)))]
    mode = ["raised_cosine", "hard_zero"][int(rng.integers(2))]
    long_run = trial % 10 == 9 and not LIVE
    total = int(rng.integers(5_000_000, 30_000_000)) if long_run else int(rng.integers(1, 300000))
    if long_run:
        overlap = [0.5, 0.75, 0.875][int(rng.integers(3))]      # hops N / 2^k: the bands share a frame grid for the windows
    gb, ob, prev = [], [], 0.0
    n_prev = None
    for lo, hi in zip(edges[:-1], edges[1:]):
        n = n_prev if (n_prev and rng.random() < 0.5) else sizes[int(rng.integers(len(sizes)))]   # repeats -> merged launches
        n_prev = n
        if int(n * (1 - overlap)) < 1 or -(-n // int(n * (1 - overlap))) > 64:
            continue
        width = 0.25 * hi
        gb.append(ux.MultiBandExtractorAccu(n, overlap, ux.WINDOW_FUNCS[wname], lo, hi, 44100, mode, prev, width))
        ob.append(orc.Band(n, overlap, lo, hi, 44100, mode, prev, width, window=orc.WINDOWS[wname]))
        prev = width
    if not gb:
        continue
    x = orc.synthetic_stereo(total, (11, trial))
    if long_run:
        plan = ux.DevicePlan(gb)
        d_in, d_out = plan.alloc(total * 8), [plan.alloc(total * 4) for _ in range(3)]
        try:
            plan.h2d(d_in, x)
            plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
            got = [np.empty(total, np.float32) for _ in range(3)]
            for o, d in zip(got, d_out):
                plan.d2h(o, d)
            n_max = max(b.block_size for b in gb)
            grid = 2 * max(b.hop_size for b in gb)
            length = 4 * n_max + 60000
            seams = []
            for b in range(len(gb)):
                if plan.band_group(b)[0] == b:
                    st = np.unique(plan.band_stream_starts(b).astype(np.int64))
                    st = st[st > 0][:-1]
                    if len(st):
                        seams.append(int(st[int(rng.integers(len(st)))]) * gb[b].hop_size - length // 2)
            starts = [0, total - length] + seams[:2] + [int(v) for v in rng.integers(0, total - length, size=2)]
            errs = []
            for a in starts:
                a = max(0, min(a // grid * grid, (total - length) // grid * grid))
                seg = x[a:a + length].astype(np.float64)
                ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
                lo = 0 if a == 0 else n_max
                errs += [rms(g[a + lo:a + length - n_max].astype(np.float64) - r[lo:length - n_max]) for g, r in zip(got, ref)]
            fills = [plan.band_fill(b) for b in range(len(gb)) if plan.band_group(b)[0] == b]
        finally:
            for d in [d_in] + d_out:
                plan.free(d)
            plan.close()
        bad = any((not np.all(np.isfinite(g[::101]))) for g in got)
        worst = max(worst, max(errs))
        flag = "BAD" if (bad or max(errs) > 1e-5) else "ok"
        print(f"{trial:3d} {flag} N={[b.block_size for b in gb]} ov={overlap} {wname} {mode} T={total} LONG one launch per group, "
              f"{len(starts)} oracle windows ({len(seams[:2])} at stream seams), fill {[round(f['workgroups'] / max(f['slots'], 1), 2) for f in fills]} "
              f"err={max(errs):.2e}", flush=True)
        del got, x
        continue
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    # the drop-in entry on the caller's own arrays (upx_process_lr): float32 column views, float64 column views (main.py:49-50),
    # two contiguous float64 arrays - in turn; the cast and the interleave happen on the device
    kind = trial % 3
    if kind == 0:
        L, R = x[:, 0], x[:, 1]
    else:
        x64 = x.astype(np.float64)
        L, R = (x64[:, 0], x64[:, 1]) if kind == 1 else (x64[:, 0].copy(), x64[:, 1].copy())
    got = ux.extract_center_left_right_multi_band_in_memory(L, R, 44100, gb)
    errs = [rms(g.astype(np.float64) - r) for g, r in zip(got, ref)]
    plan = ux.DevicePlan(gb)
    try:
        chunk = int(rng.integers(1, max(2, total)))
        try:
            gc = plan.process_chunked(x, chunk)
            errs += [rms(g.astype(np.float64) - r) for g, r in zip(gc, ref)]
        except NotImplementedError:
            pass   # hops that do not share a shard grid
    finally:
        plan.close()
    bad = any((not np.all(np.isfinite(g))) for g in got)
    worst = max(worst, max(errs))
    flag = "BAD" if (bad or max(errs) > 1e-5) else "ok"
    kn = ""
    if LIVE:
        q = ux.DevicePlan(gb)
        kn = " " + ",".join(q.band_kernel_name(i).split("Live<")[-1].rstrip(">") if "Live<" in q.band_kernel_name(i) else "-" for i in range(len(gb)))
        q.close()
    print(f"{trial:3d} {flag} N={[b.block_size for b in gb]} ov={overlap} {wname} {mode} T={total} err={max(errs):.2e}{kn}", flush=True)
print("worst", worst, "elapsed", time.time() - t0)
