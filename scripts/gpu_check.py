#!/usr/bin/env python3
"""Developer check on a GPU box: parity of each kernel size vs the oracle + C3 timing per band."""
import os
os.environ.setdefault("UPX_TUNING", "1")   # round 6: the library reads its UPX_* knobs only in a process that opts in
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import upmix_amd as ux  # noqa: E402
from oracle import upmix_oracle as orc  # noqa: E402


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


def parity():
    worst = 0.0
    for n, total, lo, hi, wl, wh in [(256, 30000, 7680., 24000., 480., 6000.), (512, 30000, 3000., 24000., 480., 6000.),
                                     (1024, 50000, 1920., 7680., 480., 1920.), (2048, 50000, 0., 24000., 0., 6000.),
                                     (4096, 100000, 480., 1920., 120., 480.), (8192, 150000, 120., 480., 30., 120.)]:
        for ov in (0.75, 0.5, 0.875):
            x = orc.synthetic_stereo(total, n)
            ob = orc.Band(n, ov, lo, hi, 48000, "raised_cosine", wl, wh)
            ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
            gb = ux.MultiBandExtractorAccu(n, ov, ux.make_blackman_harris, lo, hi, 48000, "raised_cosine", wl, wh)
            got = gb.process_all_blocks(x[:, 0], x[:, 1])
            errs = [rms(g.astype(np.float64) - r) for g, r in zip(got, ref)]
            worst = max(worst, *errs)
            print(f"N={n} ov={ov} T={total} rms err {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}", flush=True)
    print("worst", worst)
    return worst


def c3_timing(seconds=600, reps=5):
    sr = 48000
    total = sr * seconds
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, sr, max_block_size=8192,
                           verbose=False)
    plan = ux.DevicePlan(bands)
    x = orc.synthetic_stereo(total, 2)
    d_in = plan.alloc(total * 8)
    d_out = [plan.alloc(total * 4) for _ in range(3)]
    plan.h2d(d_in, x)
    plan.enable_timing(True)
    for r in range(reps):
        t0 = time.perf_counter()
        plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
        plan.sync()
        dt = time.perf_counter() - t0
        ms = plan.band_times_ms()
        print(f"rep {r}: wall {dt*1e3:.2f} ms  bands {np.round(ms, 3).tolist()} sum {ms.sum():.2f} ms "
              f"-> {total/dt/1e6:.1f} Msamples/s, {20*6*total/ms.sum()/1e6:.1f} GB/s algorithmic", flush=True)
    print([plan.band_info(b) for b in range(6)])
    # parity on a 2 s prefix of the same signal vs the oracle
    pre = sr * 2
    outs = [np.empty(pre, np.float32) for _ in range(3)]
    for o, d in zip(outs, d_out):
        plan.d2h(o, d)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, sr, max_block_size=8192)
    # the prefix of the full run equals a run on a longer prefix except near its end: compare the first second
    ref = orc.extract_multi_band(x[:pre + 8192, 0].astype(np.float64), x[:pre + 8192, 1].astype(np.float64), ob)
    for name, g, r in zip("C L R".split(), outs, ref):
        print(f"C3 prefix {name}: rms err {rms(g[:pre-8192].astype(np.float64) - r[:pre-8192]):.2e}")


def default_plan_timing(seconds=600, reps=3):
    """The reference's own default plan (STFT up to 65536) at C3 length."""
    sr = 48000
    total = sr * seconds
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, sr, verbose=False)
    plan = ux.DevicePlan(bands)
    x = orc.synthetic_stereo(total, 2)
    d_in = plan.alloc(total * 8)
    d_out = [plan.alloc(total * 4) for _ in range(3)]
    plan.h2d(d_in, x)
    plan.enable_timing(True)
    for r in range(reps):
        t0 = time.perf_counter()
        plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
        plan.sync()
        dt = time.perf_counter() - t0
        print(f"default plan rep {r}: wall {dt*1e3:.2f} ms bands {np.round(plan.band_times_ms(), 3).tolist()} "
              f"sizes {[b.block_size for b in bands]}", flush=True)
    pre = sr * 4
    outs = [np.empty(pre, np.float32) for _ in range(3)]
    for o, d in zip(outs, d_out):
        plan.d2h(o, d)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, sr)
    ref = orc.extract_multi_band(x[:pre + 65536, 0].astype(np.float64), x[:pre + 65536, 1].astype(np.float64), ob)
    for name, g, r in zip("C L R".split(), outs, ref):
        print(f"default plan prefix {name}: rms err {rms(g[:pre-65536].astype(np.float64) - r[:pre-65536]):.2e}")


def e2e_timing(seconds=600, reps=4):
    """PCIe-inclusive rates: host float32 planes (upx_process) and the PCM16 WAV pipeline."""
    from upmix_amd import _lib
    sr = 48000
    total = sr * seconds
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, sr, max_block_size=8192,
                           verbose=False)
    plan = ux.DevicePlan(bands)
    x = orc.synthetic_stereo(total, 2)
    for r in range(reps):
        t0 = time.perf_counter()
        res = plan.process(x)        # fresh result arrays: their pages are faulted in during the call
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        del res                      # (returning 345 MB to the OS is the caller's cost, timed separately)
        df = time.perf_counter() - t1
        print(f"upx_process (host f32 in, 3 fresh f32 planes out; streamed) rep {r}: {dt*1e3:.1f} ms -> "
              f"{total/dt/1e6:.1f} Msamples/s   [freeing the result: {df*1e3:.1f} ms]", flush=True)
    import ctypes as C
    outs = [np.empty(total, np.float32) for _ in range(3)]
    for chunk in (1 << 20, 1 << 21, 1 << 22, 1 << 23, 1 << 24):
        for r in range(2):
            t0 = time.perf_counter()
            _lib.check(plan._lib.upx_process_chunked(plan.handle, x.ctypes.data_as(_lib.f32p), total,
                                                     *(o.ctypes.data_as(_lib.f32p) for o in outs), chunk))
            dt = time.perf_counter() - t0
        print(f"upx_process_chunked chunk {chunk} into warm buffers: {dt*1e3:.1f} ms -> {total/dt/1e6:.1f} Msamples/s", flush=True)
    os.environ["UPX_STREAM_CHUNK"] = "0"
    for r in range(2):
        t0 = time.perf_counter()
        plan.process(x)
        dt = time.perf_counter() - t0
        print(f"upx_process one shot (UPX_STREAM_CHUNK=0) rep {r}: {dt*1e3:.1f} ms -> {total/dt/1e6:.1f} Msamples/s", flush=True)
    del os.environ["UPX_STREAM_CHUNK"]
    pcm = np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.int16)
    for r in range(reps):
        t0 = time.perf_counter()
        res = plan.wav_pipeline(pcm, _lib.PCM16, 2, total, "stereo_sum", _lib.PCM16)
        dt = time.perf_counter() - t0
        del res
        print(f"upx_wav_pipeline (PCM16 in, PCM16 stereo_sum out) rep {r}: {dt*1e3:.1f} ms -> {total/dt/1e6:.1f} Msamples/s "
              f"{plan.wav_pipeline_times_ms()}", flush=True)


if __name__ == "__main__":
    print("variant", os.environ.get("UPX_KERNEL_VARIANT", "0"))
    if "parity" in sys.argv:
        parity()
    if "c3" in sys.argv:
        c3_timing()
    if "default" in sys.argv:
        default_plan_timing()
    if "e2e" in sys.argv:
        e2e_timing()
