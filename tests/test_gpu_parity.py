"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI (libupmix_hip.so via upmix_amd), against
  * the golden fixtures generated from the reference (tests/golden), and
  * the CPU oracle on the same seeded inputs,
at the tolerance BASELINE.json states: 1e-5 RMS absolute on Ls/C/Rs (float32).
Full-size runs (BASELINE configs) are checked on windows the oracle can afford
plus size-independent properties (silence -> zeros, shard/seam invariance,
homogeneity).
"""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden, rms

pytestmark = pytest.mark.gpu

TOL = 1e-5   # RMS, absolute (BASELINE.json north_star)


@pytest.fixture(scope="module")
def ux():
    import upmix_amd
    from upmix_amd import _lib
    assert _lib.device_count() >= 1, "no GPU visible"
    return upmix_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import upmix_oracle
    return upmix_oracle


def close(got, ref, tol=TOL):
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.all(np.isfinite(got))
    err = rms(got.astype(np.float64) - ref.astype(np.float64))
    assert err <= tol, err
    return err


def gpu_chain(ux, edges, sr, max_block, tf, mode="raised_cosine", window=None, overlap=0.75):
    return ux.chain_bands(edges, overlap, window or ux.make_blackman_harris, sr, mode, max_block_size=max_block,
                          threshold_factor=tf, verbose=False)


def seam_positions(plan, bands, per_group=3, seed=0):
    """Sample positions of stream seams of the LAST process_device call, a few per launch group (seeded choice among the
    interior seams): {group leader: [sample, ...]} (upx_plan_band_stream_starts: frames -> samples through the band's hop)."""
    rng = np.random.default_rng(seed)
    out = {}
    for b in range(len(bands)):
        leader, size = plan.band_group(b)
        if leader != b:
            continue
        starts = plan.band_stream_starts(b).astype(np.int64)
        inner = np.unique(starts[(starts > 0)])[:-1]            # (the last entry is the end frame)
        assert len(inner) >= per_group, (b, len(inner))
        pick = rng.choice(inner, size=per_group, replace=False)
        out[b] = [int(f) * int(bands[b].hop_size) for f in pick]
    return out


def check_oracle_windows(orc, ob, x, out, starts, length, margin, grid):
    """Oracle restarted at each window start (a multiple of `grid` = 2 hop_max, so every band's frames coincide with the
    signal's own): behind its fade-in (`margin` = the largest STFT) and before its tail both must agree.  Returns the
    number of samples compared."""
    total, checked = len(x), 0
    for a in starts:
        a = max(0, min(int(a) // grid * grid, (total - length) // grid * grid))
        seg = x[a:a + length].astype(np.float64)
        ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
        lo = 0 if a == 0 else margin
        for got, r in zip(out, ref):
            close(got[a + lo:a + length - margin], r[lo:length - margin])
        checked += length - margin - lo
    return checked


def test_golden_single_frame_through_chunk_api(ux):
    z = load_golden("f3_frames.npz")
    for n in (256, 2048, 8192):
        lo, hi, wl, wh = z[f"N{n}_params"]
        bex = ux.MultiBandExtractorAccu(n, 0.75, ux.make_blackman_harris, lo, hi, 48000, "raised_cosine", wl, wh)
        x = z[f"N{n}_x"]
        c, l, r = bex.process_stereo_chunk(x[:, 0], x[:, 1])
        fc, fl, fr = bex.flush_final()
        assert len(c) == n // 4 and len(fc) == n
        for got, tail, key in ((c, fc, "rec_c"), (l, fl, "rec_l"), (r, fr, "rec_r")):
            close(np.concatenate([got, tail])[:n], z[f"N{n}_{key}"])
        assert not bex.accumC.any()


def test_golden_one_band_framing(ux):
    z = load_golden("f4_oneband.npz")
    for tag in ("T12345", "T1000", "T512", "T2048", "T1"):
        bex = ux.MultiBandExtractorAccu(2048, 0.75, ux.make_blackman_harris, 0.0, 24000.0, 48000, "raised_cosine",
                                        0.0, 6000.0)
        x = z[f"{tag}_x"]
        out = bex.process_all_blocks(x[:, 0], x[:, 1])
        for got, k in zip(out, "clr"):
            close(got, z[f"{tag}_{k}"])


def test_golden_other_overlaps_and_windows(ux):
    z = load_golden("f4_oneband.npz")
    for tag, ov, wname, n in (("ov50_sqrt_hann", 0.5, "sqrt_hann", 1024), ("ov875_hann", 0.875, "hann", 1024)):
        bex = ux.MultiBandExtractorAccu(n, ov, ux.WINDOW_FUNCS[wname], 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0)
        x = z[f"{tag}_x"]
        out = bex.process_all_blocks(x[:, 0], x[:, 1])
        for got, k in zip(out, "clr"):
            close(got, z[f"{tag}_{k}"])


def test_unsupported_shapes_fail_loudly(ux):
    x = np.zeros((4000, 2), np.float32)
    # sizes outside 64..65536 are not covered
    bex = ux.MultiBandExtractorAccu(32, 0.75, ux.make_hann, 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0)
    with pytest.raises(NotImplementedError):
        bex.process_all_blocks(x[:, 0], x[:, 1])
    # more than 64 frames overlapping one sample (hop = 2 of 256) is refused, not silently slow
    bex = ux.MultiBandExtractorAccu(256, 0.995, ux.make_hann, 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0)
    with pytest.raises(NotImplementedError):
        bex.process_all_blocks(x[:, 0], x[:, 1])
    with pytest.raises(ValueError):
        ux.MultiBandExtractorAccu(4, 0.9, ux.make_hann, 0.0, 100.0, 48000)


def test_arbitrary_overlap_unfused_path(ux, orc, monkeypatch):
    """hop = int(N (1 - overlap)) that does not divide N, K = 16, and the unfused path forced."""
    z = load_golden("f4_oneband.npz")
    bex = ux.MultiBandExtractorAccu(512, 0.6, ux.make_hamming, 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0)
    assert bex.hop_size == 204
    x = z["ov60_hamming_x"]
    for got, k in zip(bex.process_all_blocks(x[:, 0], x[:, 1]), "clr"):
        close(got, z[f"ov60_hamming_{k}"])
    x = orc.synthetic_stereo(300000, 13)
    for n, ov, wname in ((64, 0.75, "hann"), (128, 0.75, "blackman_harris"), (1024, 0.7, "hann"), (256, 0.9375, "hann"), (2048, 0.35, "sqrt_hann"),
                         (4096, 0.8, "blackman_harris"), (8192, 0.9, "blackman"), (16384, 0.6, "hann")):
        gb = ux.MultiBandExtractorAccu(n, ov, ux.WINDOW_FUNCS[wname], 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0)
        ob = orc.Band(n, ov, 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0, window=orc.WINDOWS[wname])
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
        for got, r in zip(gb.process_all_blocks(x[:, 0], x[:, 1]), ref):
            close(got, r)
    # the unfused pipeline on a plan the fused kernels normally take (merged bands included)
    monkeypatch.setenv("UPX_FORCE_UNFUSED", "1")
    bands = gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], 48000, 8192, 32)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000, max_block_size=8192)
    x = orc.synthetic_stereo(200000, 14)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    plan = ux.DevicePlan(bands)
    for got, r in zip(plan.process(x), ref):
        close(got, r)
    plan.close()


def test_unfused_wide_bands_at_large_stft_one_pass_tail(ux, orc, monkeypatch):
    """Round 6: wide bands at STFT 32 768 / 65 536 - e.g. chain_bands([0, 3000]) with the reference's
    default max_block_size - cannot take the band-limited path and run the unfused pipeline.  Its last two kernels (step 2 of
    the inverse transforms, then the gather overlap-add) are ONE pass now for hop N/2, N/4, N/8 (upx_big_tail_kernel:
    the overlap-add state of a column in registers).  Against the oracle, and against the two-kernel route of rounds 1-5
    (UPX_BIG_TAIL=0): the same float32 additions in the same order.  Small chunks (UPX_BIG_CHUNK_LOG2) put several chunks
    and ragged ranges of emitted blocks into a short signal."""
    x = orc.synthetic_stereo(900_000, 15)
    cases = [
        ("default wide", dict(edges=[0, 3000], overlap=0.75, max_block=65536, mode="raised_cosine")),
        ("K = 2", dict(edges=[0, 2500], overlap=0.5, max_block=32768, mode="hard_zero")),
        ("K = 8", dict(edges=[0, 2000], overlap=0.875, max_block=65536, mode="raised_cosine")),
    ]
    for chunk_log2 in ("24", "20"):
        monkeypatch.setenv("UPX_BIG_CHUNK_LOG2", chunk_log2)
        for name, c in cases:
            ob = orc.plan_bands(c["edges"], c["overlap"], orc.win_blackman_harris, 48000, c["mode"], max_block_size=c["max_block"])
            ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
            outs = {}
            for tail in ("1", "0"):
                monkeypatch.setenv("UPX_BIG_TAIL", tail)
                bands = gpu_chain(ux, c["edges"], 48000, c["max_block"], 32, mode=c["mode"], overlap=c["overlap"])
                assert bands[0].block_size == c["max_block"]
                plan = ux.DevicePlan(bands)
                assert "unfused" in plan.band_kernel_name(0), name
                outs[tail] = plan.process(x)
                plan.close()
                for got, r in zip(outs[tail], ref):
                    close(got, r)
            for a, b in zip(outs["1"], outs["0"]):
                assert np.array_equal(a, b), (name, chunk_log2, rms(a.astype(np.float64) - b))
    # two merged 65 536 bands (one transform, per-bin gain list) whose pass bands are too wide for the band-limited path
    monkeypatch.setenv("UPX_BIG_CHUNK_LOG2", "24")
    monkeypatch.delenv("UPX_BIG_TAIL")
    gb = [ux.MultiBandExtractorAccu(65536, 0.75, ux.make_blackman_harris, lo, hi, 48000, "raised_cosine", wl, wh)
          for lo, hi, wl, wh in ((0.0, 900.0, 0.0, 225.0), (900.0, 4000.0, 225.0, 1000.0))]
    ob = [orc.Band(65536, 0.75, lo, hi, 48000, "raised_cosine", wl, wh) for lo, hi, wl, wh in ((0.0, 900.0, 0.0, 225.0), (900.0, 4000.0, 225.0, 1000.0))]
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    plan = ux.DevicePlan(gb)
    assert plan.band_group(0) == (0, 2) and "unfused" in plan.band_kernel_name(0)
    for got, r in zip(plan.process(x), ref):
        close(got, r)
    plan.close()


def test_golden_multi_band(ux):
    z = load_golden("f5_multiband.npz")
    plans = {
        "c3_6band_8192_48k": gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], 48000, 8192, 32),
        "c2_3band_4096_48k": gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64),
        "c4_6band_8192_96k": gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], 96000, 8192, 32),
    }
    for name, bands in plans.items():
        x = z[f"{name}_x"]
        out = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], bands[0].sr, bands)
        for got, k in zip(out, "clr"):
            close(got, z[f"{name}_{k}"])


def test_reference_default_plan_stft_65536(ux, orc):
    """chain_bands with the reference's own defaults -> [65536, 65536, 16384, 4096, 1024, 256]."""
    z = load_golden("f5_multiband.npz")
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 48000, verbose=False)
    assert [b.block_size for b in bands] == [65536, 65536, 16384, 4096, 1024, 256]
    x = z["default_65536_x"]
    out = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
    for got, k in zip(out, "clr"):
        close(got, z[f"default_65536_{k}"])
    # longer signal (several chunks per band), 96 kHz plan [65536 x2, 32768, 8192, 2048, 512], vs the oracle
    x = orc.synthetic_stereo(1_500_000, 12)
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 96000, verbose=False)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 96000)
    assert [b.block_size for b in bands] == [65536, 65536, 32768, 8192, 2048, 512]
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    for got, r in zip(ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 96000, bands), ref):
        close(got, r)


def test_golden_degenerate_inputs(ux):
    z = load_golden("f6_degenerate.npz")
    for tag in ("silence", "l_eq_r", "r_zero", "l_zero", "tiny", "antiphase"):
        bands = gpu_chain(ux, [0, 300, 3000], 48000, 1024, 32)
        x = z[f"{tag}_x"]
        out = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
        for got, k in zip(out, "clr"):
            close(got, z[f"{tag}_{k}"])
        if tag == "silence":
            assert not any(o.any() for o in out)          # exact zeros, no NaN
        if tag == "r_zero":
            # the reference gives exactly 0 here; L and R share one complex FFT on the GPU, so R's spectrum
            # is rounding noise (~1e-8 |L|) instead of 0 and a centre of that order appears
            assert rms(out[0]) < 1e-7
        if tag == "l_eq_r":
            assert rms(out[1]) < 1e-7 and rms(out[2]) < 1e-7   # mono -> everything in the centre


def test_hard_zero_and_unknown_mode(ux, orc):
    x = orc.synthetic_stereo(20000, 77)
    for mode in ("hard_zero", "no_such_mode"):
        bex = ux.MultiBandExtractorAccu(1024, 0.75, ux.make_blackman_harris, 300.0, 3000.0, 48000, mode, 75.0, 750.0)
        ob = orc.Band(1024, 0.75, 300.0, 3000.0, 48000, mode, 75.0, 750.0)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
        for got, r in zip(bex.process_all_blocks(x[:, 0], x[:, 1]), ref):
            close(got, r)


def test_c1_full_vs_oracle(ux, orc):
    """BASELINE configs[0]: 10 s, 48 kHz, 1 band, STFT 2048, full size, seed 0."""
    x = orc.synthetic_stereo(480000, 0)
    bex = ux.MultiBandExtractorAccu(2048, 0.75, ux.make_blackman_harris, 0.0, 24000.0, 48000, "raised_cosine", 0.0, 6000.0)
    ob = orc.Band(2048, 0.75, 0.0, 24000.0, 48000, "raised_cosine", 0.0, 6000.0)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    for got, r in zip(bex.process_all_blocks(x[:, 0], x[:, 1]), ref):
        close(got, r)


def test_c2_prefix_vs_oracle(ux, orc):
    """BASELINE configs[1] plan [4096, 4096, 1024], 2 s prefix, seed 1."""
    x = orc.synthetic_stereo(96000, 1)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    ob = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=4096, threshold_factor=64)
    assert [b.block_size for b in bands] == [4096, 4096, 1024]
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    for got, r in zip(ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands), ref):
        close(got, r)


def test_c2_full_size(ux, orc):
    """BASELINE configs[1] whole: 60 s of 48 kHz stereo (2.88 M samples), 3 bands (crossovers 300 / 3000 Hz), STFT
    [4096, 4096, 1024], seed 1: oracle windows at the head, in the interior and at the tail, time shards + seams ==
    single launch, silence -> exact zeros."""
    total = 2_880_000
    x = orc.synthetic_stereo(total, 1)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    ob = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=4096, threshold_factor=64)
    assert [b.block_size for b in bands] == [4096, 4096, 1024]
    out = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
    assert all(o.shape == (total,) and np.all(np.isfinite(o)) for o in out)
    n = 120000
    ref = orc.extract_multi_band(x[:n + 4096, 0].astype(np.float64), x[:n + 4096, 1].astype(np.float64), ob)
    for got, r in zip(out, ref):
        close(got[:n], r[:n])
    for a in (1024 * 700, 1024 * 1999):          # interior windows on the hop_max grid (the restarted oracle fades in)
        seg = x[a:a + 100000].astype(np.float64)
        ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
        for got, r in zip(out, ref):
            close(got[a + 4096:a + 100000 - 4096], r[4096:100000 - 4096])
    a = (total // 1024 - 60) * 1024
    seg = x[a:].astype(np.float64)
    ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
    for got, r in zip(out, ref):
        close(got[a + 4096:], r[4096:])
    from upmix_amd.sharding import process_sharded_single_device
    plan = ux.DevicePlan(bands)
    for a, b in zip(out, process_sharded_single_device(plan, x, max_shard=700_000)):
        assert rms(a.astype(np.float64) - b) < 1e-8 and float(np.max(np.abs(a - b))) < 1e-6
    for o in plan.process(np.zeros((total, 2), np.float32)):
        assert not o.any()
    plan.close()


@pytest.fixture(scope="module")
def c3_full(ux, orc):
    """BASELINE configs[2] at full size: 10 min, 48 kHz, 6 bands, STFT <= 8192, seed 2."""
    x = orc.synthetic_stereo(28_800_000, 2)
    bands = gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], 48000, 8192, 32)
    out = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
    return x, bands, out


def test_c3_full_size_windows_vs_oracle(ux, orc, c3_full):
    x, bands, out = c3_full
    total = len(x)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000, max_block_size=8192)
    assert [b.block_size for b in ob] == [8192, 8192, 8192, 4096, 1024, 256]
    assert all(o.shape == (total,) and np.all(np.isfinite(o)) for o in out)
    # head (includes the no-pre-roll fade-in of SURVEY 3.3)
    n = 60000
    ref = orc.extract_multi_band(x[:n + 8192, 0].astype(np.float64), x[:n + 8192, 1].astype(np.float64), ob)
    for got, r in zip(out, ref):
        close(got[:n], r[:n])
    # interior windows aligned to hop_max = 2048: the oracle restarted at `a` has its own fade-in over the first
    # 0.75 N samples, after that both agree
    for a in (2048 * 5000, 2048 * 11111):
        seg = x[a:a + 90000].astype(np.float64)
        ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
        for got, r in zip(out, ref):
            close(got[a + 8192:a + 90000 - 8192], r[8192:90000 - 8192])
    # tail: the end of the signal is fully covered and trimmed to T
    a = (total // 2048 - 40) * 2048
    seg = x[a:].astype(np.float64)
    ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
    for got, r in zip(out, ref):
        close(got[a + 8192:], r[8192:])


def test_c3_full_size_one_launch_per_band_windows_at_stream_seams(ux, orc, c3_full):
    """BASELINE configs[2] as bench.py runs it - the whole 28.8 M samples resident, ONE launch
    per band group, i.e. the full-chip launch geometry (stream tables, shorter edge streams, XCD dealing) that only long
    signals select.  Oracle windows at >= 16 seeded offsets, half of them centred on stream seams read off the launch
    geometry of each group (upx_plan_band_stream_starts), the head and the tail: >= 5 % of the signal, 1e-5 RMS each.  Plus,
    over ALL samples: the streamed drop-in entry (4 M-sample chunks: other seams) agrees to float32 seam rounding,
    homogeneity f(2x) = 2 f(x), channel swap."""
    x, bands, streamed = c3_full
    total = len(x)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000, max_block_size=8192)
    plan = ux.DevicePlan(bands)
    d_in = plan.alloc(total * 8)
    d_out = [plan.alloc(total * 4) for _ in range(3)]

    def run(sig):
        plan.h2d(d_in, sig)
        plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
        res = [np.empty(total, np.float32) for _ in range(3)]
        for o, d in zip(res, d_out):
            plan.d2h(o, d)
        return res
    try:
        out = run(x)
        # the geometry under test IS the full-chip one: every group fills (nearly) all its workgroup slots
        for b in (0, 3, 4, 5):
            fill = plan.band_fill(b)
            assert fill["workgroups"] >= 0.9 * fill["slots"], (b, fill)
        seams = seam_positions(plan, bands, per_group=3, seed=6)
        assert sorted(seams) == [0, 3, 4, 5]
        length, margin = 100_000, 8192
        centred = [p - length // 2 for g in sorted(seams) for p in seams[g]]                 # 12 windows across seams
        rng = np.random.default_rng(66)
        anywhere = [int(v) for v in rng.integers(0, total - length, size=8)]                # 8 seeded offsets
        checked = check_oracle_windows(orc, ob, x, out, [0] + centred + anywhere + [total - length], length, margin, 4096)
        assert checked >= 0.05 * total, checked
        # every seam window really holds its seam behind the oracle's fade-in
        for g in seams:
            for p_ in seams[g]:
                a = max(0, (p_ - length // 2) // 4096 * 4096)
                assert a + margin < p_ < a + length - margin
        for a, b in zip(out, streamed):
            assert rms(a.astype(np.float64) - b) < 1e-8 and float(np.max(np.abs(a - b))) < 1e-6
        # homogeneity over all 28.8 M samples: the mask is scale invariant up to EPS = 1e-12 in its denominators
        twice = run(x * np.float32(2.0))
        for a, b in zip(out, twice):
            assert rms(2.0 * a.astype(np.float64) - b) < 1e-7
            assert float(np.max(np.abs(2.0 * a.astype(np.float64) - b))) < 1e-5
        del twice
        # channel swap over all samples: C stays, Ls and Rs trade places
        sw = run(np.ascontiguousarray(x[:, ::-1]))
        assert rms(sw[0].astype(np.float64) - out[0]) < 1e-7
        assert rms(sw[1].astype(np.float64) - out[2]) < 1e-7 and rms(sw[2].astype(np.float64) - out[1]) < 1e-7
    finally:
        for d in [d_in] + d_out:
            plan.free(d)
        plan.close()


def test_c3_full_size_float64_views_equal_the_float32_call(ux, orc, c3_full):
    """BASELINE configs[2] at full size through the documented drop-in call: a float64 [T, 2] parent, its two column views
    (main.py:49-50), seven streamed chunks, cast + interleave on the device - bit for bit the planes of the float32 call."""
    x, bands, out = c3_full
    wave = x.astype(np.float64)
    got = ux.extract_center_left_right_multi_band_in_memory(wave[:, 0], wave[:, 1], 48000, bands)
    for g, o in zip(got, out):
        assert g.dtype == np.float32 and np.array_equal(g, o)
    del wave


def test_c3_properties_full_size(ux, orc, c3_full):
    x, bands, out = c3_full
    total = len(x)
    # (1) time sharding + on-device seam == single launch (same arithmetic up to float32 association at seams)
    from upmix_amd.sharding import process_sharded_single_device, ShardGeometry
    plan = ux.DevicePlan(bands)
    sharded = process_sharded_single_device(plan, x, max_shard=7_000_000)
    geo = ShardGeometry(plan.block_sizes, plan.hops)
    assert geo.hop_max == 2048 and geo.spill == 6144
    for a, b in zip(out, sharded):
        assert rms(a.astype(np.float64) - b) < 1e-8
        assert float(np.max(np.abs(a - b))) < 1e-6
    # (2) homogeneity: the mask is scale invariant (up to EPS), so f(2x) = 2 f(x)
    y = ux.extract_center_left_right_multi_band_in_memory(2 * x[:400000, 0], 2 * x[:400000, 1], 48000, bands)
    for a, b in zip(out, y):
        assert rms(2.0 * a[:300000].astype(np.float64) - b[:300000]) < 1e-6
    # (3) silence in -> exact zeros out, at full size
    z = np.zeros((total, 2), np.float32)
    for o in plan.process(z):
        assert not o.any()
    # (4) swapping channels swaps Ls/Rs and keeps C
    sw = ux.extract_center_left_right_multi_band_in_memory(x[:400000, 1], x[:400000, 0], 48000, bands)
    assert rms(sw[0][:300000].astype(np.float64) - out[0][:300000]) < 1e-7
    assert rms(sw[1][:300000].astype(np.float64) - out[2][:300000]) < 1e-7
    plan.close()


def test_run_to_run_deterministic(ux, orc):
    x = orc.synthetic_stereo(300000, 9)
    bands = gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], 48000, 8192, 32)
    a = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
    b = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)   # gather overlap-add, no float atomics


def test_blocks_per_stream_only_moves_seam_rounding(ux, orc):
    """A different stream cut moves the seams (float32 association on the K-1 blocks after each); nothing else."""
    x = orc.synthetic_stereo(200000, 10)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    plan = ux.DevicePlan(bands)
    base = plan.process(x)
    for f in (4, 16, 1000):
        plan.set_blocks_per_stream(f)
        for u, v in zip(base, plan.process(x)):
            assert float(np.max(np.abs(u - v))) < 1e-6, f
            assert rms(u.astype(np.float64) - v) < 1e-8, f
    plan.close()


def test_timing_pause_keeps_recorded_calls(ux, orc):
    """upx_plan_enable_timing 2 / 3 (resume / pause): a loop may time every n-th call; the ring keeps the timed ones only."""
    x = orc.synthetic_stereo(150000, 12)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    plan = ux.DevicePlan(bands)
    d_in = plan.alloc(x.nbytes)
    d_out = [plan.alloc(len(x) * 4) for _ in range(3)]
    plan.h2d(d_in, x)
    plan.enable_timing(True)
    for i in range(6):
        plan.pause_timing(i % 3 != 0)       # calls 0 and 3 are timed
        plan.process_device(d_in, len(x), len(x), d_out[0], d_out[1], d_out[2], len(x))
    per_call = plan.band_times_calls_ms(2)
    assert per_call.shape == (2, len(bands)) and np.all(per_call.sum(axis=1) > 0)
    with pytest.raises(Exception):
        plan.band_times_calls_ms(3)         # only two timed calls were recorded
    plan.enable_timing(True)                # on again: forgets them
    with pytest.raises(Exception):
        plan.band_times_calls_ms(1)
    plan.close()


def test_device_helpers_absmax_scale(ux, orc):
    x = orc.synthetic_stereo(100001, 11)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 1024, 32)
    plan = ux.DevicePlan(bands)
    v = np.ascontiguousarray(x[:, 0])
    d = plan.alloc(v.nbytes)
    plan.h2d(d, v)
    assert plan.absmax(d, len(v)) == float(np.max(np.abs(v)))
    plan.scale(d, len(v), 0.5)
    back = np.empty_like(v)
    plan.d2h(back, d)
    assert np.array_equal(back, v * np.float32(0.5))
    # np.max(np.abs(.)) semantics (main.py:53, :85-88): a NaN anywhere gives NaN, an infinity wins over every number
    for bad, want in ((np.nan, "nan"), (-np.inf, "inf")):
        w = v.copy()
        w[77777] = bad
        plan.h2d(d, w)
        got = plan.absmax(d, len(w))
        assert (np.isnan(got) if want == "nan" else got == np.inf), (bad, got)
    plan.free(d)
    plan.close()


def test_results_live_in_pooled_page_locked_memory(ux, orc, monkeypatch):
    """DevicePlan.process returns NEW arrays per call (the reference returns fresh NumPy arrays, center_extraction.py:503-513)
    that live in pooled page-locked blocks (upmix_amd/hostmem.py): a block is reused only after every array on it is gone,
    results stay intact while held, and beyond the pool's limit the arrays are plain pageable ones - same numbers."""
    import gc
    from upmix_amd import hostmem
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    plan = ux.DevicePlan(bands)
    x = orc.synthetic_stereo(300000, 21)
    y = orc.synthetic_stereo(300000, 22)
    # the FIRST call that asks for a size gets plain NumPy arrays: a one-shot process (the reference's flow, main.py:78-80)
    # must not pay for pinning blocks it never reuses; the second call has proven reuse and pins (hostmem.PinnedPool.take)
    # (earlier tests of this process may have asked for the same size class: forget that, and count pinned bytes from here)
    # ... and left idle blocks of it behind, which any call may use: release them)
    gc.collect()
    held0 = hostmem.POOL.reset_for_tests(plan.handle)["held"]
    first = plan.process(x)
    assert not any(hostmem.is_pinned(o) for o in first) and hostmem.POOL.pinned_bytes() == held0
    a = plan.process(x)
    assert all(hostmem.is_pinned(o) for o in a) and all(np.array_equal(o, q) for o, q in zip(a, first))
    del first
    keep = [o.copy() for o in a]
    b = plan.process(y)                                # `a` is still held: `b` must not land on its blocks
    assert all(np.array_equal(o, k) for o, k in zip(a, keep))
    assert not any(np.shares_memory(o, q) for o in a for q in b)
    addr = {o.ctypes.data for o in a}
    view = a[0][1000:2000]                             # a view keeps its block alive
    del a
    gc.collect()
    c = plan.process(x)                                # the freed blocks come back (two of them: one is still viewed)
    assert len(addr & {o.ctypes.data for o in c}) == 2
    assert np.array_equal(view, keep[0][1000:2000]) and all(np.array_equal(o, k) for o, k in zip(c, keep))
    held = hostmem.POOL.pinned_bytes()
    assert held - held0 >= 5 * 300000 * 4              # at least the live planes of this test are pinned (b, c, the viewed block of a)
    # a pool without room hands out pageable arrays; the numbers do not change
    monkeypatch.setattr(hostmem.POOL, "limit", 0)
    d = plan.process(x)
    assert all(np.array_equal(o, k) for o, k in zip(d, keep)) and hostmem.POOL.pinned_bytes() == held
    left = c[1]
    left *= 2.0                                        # results are the caller's to scale in place (main.py:95-97)
    assert np.array_equal(c[1], keep[1] * np.float32(2.0))
    plan.close()


def test_drop_in_entry_takes_the_callers_arrays_as_they_are(ux, orc):
    """
    main.py:49-50, 78-80 hands extract_center_left_right_multi_band_in_memory two float64 COLUMN VIEWS of one [T, 2]
    array.  They go to the library as they are (upx_process_lr: cast to float32 and interleaved on the device); the
    result is bit-identical to the host cast + interleave + upx_process, for float64 / float32, column views / separate
    contiguous arrays, short signals and signals that stream in chunks; other dtypes and strides take the host cast.
    Values beyond the float32 range become infinite exactly as np.asarray(x, float32) makes them.
    """
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    plan = ux.DevicePlan(bands)
    for total in (1, 777, 300001, (1 << 23) + 4097 + 1):          # the last one streams (two chunks of 2^22 and a ragged rest)
        wave = orc.synthetic_stereo(total, 5).astype(np.float64)
        wave[total // 2:, 1] *= 1.0 + 2.0 ** -30                  # values that are not float32 numbers: the cast rounds
        ref = plan.process(wave.astype(np.float32))
        cases = {
            "f64 column views": (wave[:, 0], wave[:, 1]),
            "f64 contiguous": (wave[:, 0].copy(), wave[:, 1].copy()),
            "f32 column views": (wave.astype(np.float32)[:, 0], wave.astype(np.float32)[:, 1]),
            "f32 contiguous": (wave[:, 0].astype(np.float32), wave[:, 1].astype(np.float32)),
            "columns of a [T, 3] array (other strides: host cast)": (np.pad(wave, ((0, 0), (0, 1)))[:, 0], np.pad(wave, ((0, 0), (0, 1)))[:, 1]),
            "int16 (host cast)": ((wave[:, 0] * 1000).astype(np.int16), (wave[:, 1] * 1000).astype(np.int16)),
        }
        for name, (L, R) in cases.items():
            got = plan.process_lr(L, R)
            if name.startswith("int16"):
                want = plan.process(np.stack([L.astype(np.float32), R.astype(np.float32)], axis=1))
            else:
                want = ref
            for g, w in zip(got, want):
                assert g.dtype == np.float32 and g.shape == (total,) and np.array_equal(g, w), (name, total)
    # the module-level entry and the per-band entry go the same way
    wave = orc.synthetic_stereo(50000, 6).astype(np.float64)
    ref = plan.process(wave.astype(np.float32))
    for g, w in zip(ux.extract_center_left_right_multi_band_in_memory(wave[:, 0], wave[:, 1], 48000, bands), ref):
        assert np.array_equal(g, w)
    one = bands[2].process_all_blocks(wave[:, 0], wave[:, 1])
    one_ref = ux.DevicePlan([bands[2]])
    for g, w in zip(one, one_ref.process(wave.astype(np.float32))):
        assert np.array_equal(g, w)
    one_ref.close()
    # beyond the float32 range: infinity on the device as on the host (documented: non-finite output on those frames)
    big = wave.copy()
    big[100, 0] = 1e300
    got = plan.process_lr(big[:, 0], big[:, 1])
    with np.errstate(over="ignore"):
        want = plan.process(big.astype(np.float32))
    for g, w in zip(got, want):
        assert np.array_equal(g, w, equal_nan=True)
    with pytest.raises(ValueError):
        plan.process_lr(wave[:, 0], wave[:-1, 1])
    assert all(o.shape == (0,) for o in plan.process_lr(np.zeros(0), np.zeros(0)))
    plan.close()


def test_plan_reserve_prepares_a_call_shape(ux, orc):
    """upx_plan_reserve: everything the first process_device call of a shape allocates / uploads / loads, up front - no
    signal buffer is touched, the result is bit for bit the one without it, any shape may be reserved (and re-reserved)."""
    x = orc.synthetic_stereo(400000, 12)
    bands = gpu_chain(ux, [0, 120, 480, 4000], 48000, 8192, 32)
    outs = []
    for reserve in (False, True):
        plan = ux.DevicePlan(bands)
        total = x.shape[0]
        if reserve:
            plan.reserve(total, total, total)
            plan.reserve(0, 0, 0)
            plan.reserve(12345, 12345, 12345)          # another shape in between: the tables follow the last one asked for
            plan.reserve(total, total, total)
        d_in = plan.alloc(total * 8)
        d = [plan.alloc(total * 4) for _ in range(3)]
        plan.h2d(d_in, x)
        plan.process_device(d_in, total, total, d[0], d[1], d[2], total)
        got = [np.empty(total, np.float32) for _ in range(3)]
        for g, p in zip(got, d):
            plan.d2h(g, p)
        outs.append(got)
        for p in d + [d_in]:
            plan.free(p)
        plan.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        ux.DevicePlan(bands).reserve(-1, 0, 0)


POISON = {"UPX_FIRST_BAND": "4", "UPX_BAND_ROTATE": "2", "UPX_DUAL": "1", "UPX_SEAM_INKERNEL": "1", "UPX_KERNEL_VARIANT": "2",
          "UPX_N_CU": "128", "UPX_FORCE_UNFUSED": "1", "UPX_ZOOM": "0", "UPX_EDGE_PERCENT": "60", "UPX_ZOOM_ONCE": "0",
          "UPX_NO_BAND_MERGE": "1", "UPX_NO_LIVE_FLAVOUR": "1", "UPX_STREAM_CHUNK": "50000", "UPX_PRIO_YOUNG": "0",
          "UPX_ZOOM_SCRATCH_MB": "1", "UPX_MIN_STREAM_FRAMES": "16", "UPX_SEAM_VEC": "0"}


def test_a_poisoned_environment_changes_neither_kernels_nor_bits(ux, orc, monkeypatch):
    """round 5's library read ~37 UPX_* variables at plan creation, among them experiments that
    changed the band-sum association (UPX_FIRST_BAND, UPX_BAND_ROTATE) and swapped in kernels no BASELINE test covers
    (UPX_DUAL, UPX_SEAM_INKERNEL, UPX_KERNEL_VARIANT).  The experiments are compiled out of the product library now
    (-DUPX_EXPERIMENTS builds only) and every remaining knob is read only in a process that opts in with UPX_TUNING=1: without
    it, whatever the environment holds, a plan selects the same kernels, cuts the same streams and returns the same bits -
    on the drop-in entry and on the streamed host call alike."""
    x = orc.synthetic_stereo(700000, 8)
    edges = [0, 30, 120, 480, 1920, 7680]

    def run():
        bands = gpu_chain(ux, edges, 48000, 8192, 32)
        plan = ux.DevicePlan(bands)
        names = [(plan.band_kernel_name(i), plan.band_phase_kernel_name(i, 0), plan.band_phase_kernel_name(i, 1))
                 for i in range(len(bands))]
        out = plan.process(x)
        info = [plan.band_info(i) for i in range(len(bands))]
        plan.close()
        wave = x.astype(np.float64)
        entry = ux.extract_center_left_right_multi_band_in_memory(wave[:, 0], wave[:, 1], 48000, bands)
        return names, info, out, entry

    monkeypatch.delenv("UPX_TUNING")
    clean = run()
    for k, v in POISON.items():
        monkeypatch.setenv(k, v)
    dirty = run()
    assert clean[0] == dirty[0] and clean[1] == dirty[1]
    assert not any("dual" in n or "Cfg<13, 4, 16>" in n for row in dirty[0] for n in row)
    for a, b in zip(clean[2] + clean[3], dirty[2] + dirty[3]):
        assert np.array_equal(a, b)
    # ... and the same variables DO reach an opted-in process (the gate is the opt-in, not deaf knobs): the unfused pipeline
    # as a second implementation, within rounding of the default kernels
    monkeypatch.setenv("UPX_TUNING", "1")
    for k in POISON:
        monkeypatch.delenv(k)
    monkeypatch.setenv("UPX_FORCE_UNFUSED", "1")
    bands = gpu_chain(ux, edges, 48000, 8192, 32)
    plan = ux.DevicePlan(bands)
    assert all("unfused" in plan.band_kernel_name(i) for i in range(len(bands)))
    forced = plan.process(x)
    plan.close()
    for a, b in zip(clean[2], forced):
        assert not np.array_equal(a, b) and rms(a.astype(np.float64) - b) < 1e-7


def test_rccl_single_rank_communicator(ux, orc):
    """dlopen of RCCL, unique id, ncclCommInitRank, and the seam pack -> ncclAllReduce -> add kernels on real sizes."""
    from upmix_amd import sharding
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 1024, 32)
    plan = ux.DevicePlan(bands)
    votes = []
    seam = sharding.RcclSeam(plan, 0, 1, broadcast=lambda b: b, all_ok=lambda ok=True, message="": votes.append((ok, message)))
    assert votes == [(True, ""), (True, "")]          # before and after RCCL's blocking init
    own, spill = 50000, 6144
    rng = np.random.default_rng(5)
    host = [rng.standard_normal(own + spill).astype(np.float32) for _ in range(3)]
    d = [plan.alloc((own + spill) * 4) for _ in range(3)]
    for p, h in zip(d, host):
        plan.h2d(p, h)
    seam.exchange(d, own, spill)          # one rank: nothing to exchange
    seam.wait()                           # nothing pending: returns at once
    from upmix_amd import _lib
    _lib.check(_lib.load().upx_comm_reserve(seam.handle, spill))
    seam.selftest(d, own, spill, 8, 5)    # 8-row seam, my spill in row 5, all-reduce, add row 5 onto my head
    seam.wait(30.0)                       # bounded wait for the queued exchange (upx_comm_wait)
    plan.sync()
    for p, h in zip(d, host):
        got = np.empty_like(h)
        plan.d2h(got, p)
        want = h.copy()
        want[:spill] += h[own:own + spill]
        assert np.array_equal(got, want)
    # the timeout path of upx_comm_wait: an exchange queued behind ~0.2 s of other work on the plan's stream has not run when a
    # 5 ms wait expires - as if a peer had not arrived - so the communicator is aborted and the call says so
    x = orc.synthetic_stereo(2_000_000, 3)
    d_in = plan.alloc(x.shape[0] * 8)
    planes = [plan.alloc(x.shape[0] * 4) for _ in range(3)]
    plan.h2d(d_in, x)
    plan.sync()
    for _ in range(400):
        plan.process_device(d_in, x.shape[0], x.shape[0], planes[0], planes[1], planes[2], x.shape[0])
    seam.selftest(d, own, spill, 8, 5)
    import time
    t0 = time.perf_counter()
    with pytest.raises(Exception, match="did not finish within|aborted"):
        seam.wait(0.005)
    assert time.perf_counter() - t0 < 5.0
    plan.sync()                           # the stream itself drains: nothing is left hanging
    for p in planes + [d_in]:
        plan.free(p)
    # the abort path: the communicator is gone afterwards and says so (a rank whose peer never arrived ends here)
    seam.abort()
    seam.abort()                          # idempotent
    with pytest.raises(Exception, match="aborted"):
        seam.selftest(d, own, spill, 8, 5)
    with pytest.raises(Exception, match="aborted"):
        seam.wait(1.0)
    seam.close()
    for p in d:
        plan.free(p)
    plan.close()


def test_wav_pipeline_device_codec_and_cli(ux, orc, tmp_path):
    """WAV bytes in -> exported WAV out with codec, peak scale and layouts on the GPU."""
    import json
    import os
    from conftest import GOLDEN
    from upmix_amd import wav, export, _lib, cli
    z = load_golden("f7_main.npz")
    meta = json.load(open(os.path.join(GOLDEN, "f7_main.json")))
    x = z["x"]                                     # float32 [T,2]; main.py itself produced the golden outputs from it
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 48000, verbose=False)
    plan = ux.DevicePlan(bands)
    for mode in ("stereo_sum", "split", "AB"):
        payloads, stats = plan.wav_pipeline(x, _lib.F32, 2, len(x), mode, _lib.F32)
        assert f"Applying scale_factor = {stats['scale_factor']:.4f}" in meta[mode]["log_tail"]
        names = export.export_file_names("eyes", mode, bands, 0.75)
        for key, raw in payloads.items():
            got = raw.view(np.float32).reshape(-1, 2)
            want = z[f"{mode}:{os.path.join('out', names[key])}"]
            assert rms(got.astype(np.float64) - want) <= 3e-5      # 1e-5 x scale_factor 2.8
    # integer PCM: decode and quantise on the device == the host codec
    rng = np.random.default_rng(3)
    pcm = (rng.standard_normal((50000, 2)) * 3000).astype(np.int16)
    payloads, stats = plan.wav_pipeline(pcm, _lib.PCM16, 2, len(pcm), "stereo_sum", _lib.PCM16)
    xf = pcm.astype(np.float64) / 32768.0
    c, l, r = plan.process(xf.astype(np.float32))
    scale, _ = export.scale_to_input_peak(c, l, r, export.input_peak(xf))
    assert abs(scale - stats["scale_factor"]) <= 1e-6 * scale
    ref = export.export_arrays("stereo_sum", c, l, r)["Sum"]
    q = np.clip(np.rint(ref.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    got = payloads["Sum"].view(np.int16).reshape(-1, 2)
    assert np.max(np.abs(got.astype(np.int32) - q)) <= 1           # identical up to ties of float rounding
    plan.close()
    # the CLI end to end: mono 24-bit file in, three split files out
    os.makedirs(tmp_path / "in")
    mono = rng.uniform(-0.5, 0.5, 30000)
    wav.write(str(tmp_path / "in" / "tone.wav"), mono, 44100, "PCM_24")
    written = cli.run("tone.wav", "split", str(tmp_path / "in"), str(tmp_path / "out"), max_stft=4096)
    assert sorted(written) == ["C", "Ls", "Rs"]
    c_wav, sr = wav.read(written["C"])
    ls_wav, _ = wav.read(written["Ls"])
    assert sr == 44100 and c_wav.shape == (30000, 2) and np.array_equal(c_wav[:, 0], c_wav[:, 1])
    assert not ls_wav[:, 1].any() and np.max(np.abs(ls_wav[:, 0])) < 1e-3     # mono input: everything is centre
    host = cli.run("tone.wav", "split", str(tmp_path / "in"), str(tmp_path / "out_host"), max_stft=4096, host_export=True)
    c_host, _ = wav.read(host["C"])
    assert np.max(np.abs(c_host - c_wav)) <= 2.0 / 32768


def test_device_export_equals_host_export_byte_for_byte(ux, orc, tmp_path):
    """the CLI's default (codec, peak scale, export layout and quantisation on the GPU) and --host-export
    (NumPy, pinned to main.py by fixture F7) must write the same files: every export mode x every subtype, from an integer
    PCM file and from a float file (AB: the original sum L + R is formed in float64 from the decoded samples, as
    main.py:112-114 does)."""
    import os
    from upmix_amd import wav, cli
    rng = np.random.default_rng(5)
    os.makedirs(tmp_path / "in")
    x = np.clip(0.2 * rng.standard_normal((40000, 2)), -0.99, 0.99)
    wav.write(str(tmp_path / "in" / "a16.wav"), x, 48000, "PCM_16")
    wav.write(str(tmp_path / "in" / "a24.wav"), x, 48000, "PCM_24")
    wav.write(str(tmp_path / "in" / "af.wav"), x, 48000, "FLOAT")
    n = 0
    for name in ("a16.wav", "a24.wav", "af.wav"):
        for mode in ("AB", "split", "stereo_sum"):
            for subtype in ("PCM_16", "PCM_24", "PCM_32", "FLOAT"):
                kw = dict(max_stft=4096, subtype=subtype)
                dev = cli.run(name, mode, str(tmp_path / "in"), str(tmp_path / "dev"), **kw)
                host = cli.run(name, mode, str(tmp_path / "in"), str(tmp_path / "host"), host_export=True, **kw)
                assert sorted(dev) == sorted(host)
                for key in dev:
                    a, b = open(dev[key], "rb").read(), open(host[key], "rb").read()
                    assert a == b, (name, mode, subtype, key)
                    n += 1
    assert n == 3 * (1 + 3 + 1) * 4


def test_random_plans_vs_oracle(ux, orc):
    """Seeded random band plans / overlaps / windows / lengths through whichever kernel path they select."""
    rng = np.random.default_rng(2024)
    sizes = [256, 512, 1024, 2048, 4096, 8192, 16384]
    overlaps = [0.5, 0.75, 0.875, 0.6, 0.7, 0.9]
    windows = sorted(ux.WINDOW_FUNCS)
    for trial in range(12):
        n_bands = int(rng.integers(1, 4))
        edges = np.sort(rng.uniform(20.0, 15000.0, size=n_bands + 1))
        overlap = overlaps[int(rng.integers(len(overlaps)))]
        wname = windows[int(rng.integers(len(windows)))]
        mode = ["raised_cosine", "hard_zero"][int(rng.integers(2))]
        total = int(rng.integers(3000, 120000))
        gb, ob, prev = [], [], 0.0
        for lo, hi in zip(edges[:-1], edges[1:]):
            n = sizes[int(rng.integers(len(sizes)))]
            if int(n * (1 - overlap)) < 1 or -(-n // int(n * (1 - overlap))) > 64:
                continue
            width = 0.25 * hi
            gb.append(ux.MultiBandExtractorAccu(n, overlap, ux.WINDOW_FUNCS[wname], lo, hi, 44100, mode, prev, width))
            ob.append(orc.Band(n, overlap, lo, hi, 44100, mode, prev, width, window=orc.WINDOWS[wname]))
            prev = width
        if not gb:
            continue
        x = orc.synthetic_stereo(total, (7, trial))
        ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
        got = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 44100, gb)
        for g, r in zip(got, ref):
            close(g, r)


def test_band_limited_path_random_plans(ux, orc, monkeypatch):
    """The two-kernel band-limited path (upx_zoom.h; every band whose pass band allows a decimation D >= 8) on random
    band-limited plans: STFT 4096 .. 65536, K = 2 / 4 / 8, all windows, both crossover modes, repeated sizes (merged
    launches), ragged lengths, also streamed in chunks; 1e-5 RMS vs the oracle."""
    monkeypatch.setenv("UPX_ZOOM", "8")
    rng = np.random.default_rng(77)
    wnames = ["blackman_harris", "hann", "sqrt_hann", "hamming", "blackman"]
    n_zoom = 0
    for trial in range(14):
        overlap = [0.75, 0.5, 0.875, 0.75][int(rng.integers(4))]
        wname = wnames[int(rng.integers(len(wnames)))]
        mode = ["raised_cosine", "hard_zero"][int(rng.integers(2))]
        n_bands = int(rng.integers(1, 5))
        gb, ob, prev, lo = [], [], 0.0, 0.0
        for b in range(n_bands):
            n = [4096, 8192, 8192, 16384, 32768, 65536][int(rng.integers(6))]
            if b and rng.random() < 0.4:
                n = gb[-1].block_size                               # same size as the previous band: merged launch
            top = (44100.0 / n) * float(rng.integers(20, min(400, n // 20)))   # pass band ends between bin 20 and bin 400
            hi = max(lo + 44100.0 / n, top)
            width = 0.25 * hi
            gb.append(ux.MultiBandExtractorAccu(n, overlap, ux.WINDOW_FUNCS[wname], lo, hi, 44100, mode, prev, width))
            ob.append(orc.Band(n, overlap, lo, hi, 44100, mode, prev, width, window=orc.WINDOWS[wname]))
            prev, lo = width, hi
        total = int(rng.integers(1, 4 * max(b.block_size for b in gb) + 70000))
        x = orc.synthetic_stereo(total, (9, trial))
        ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
        plan = ux.DevicePlan(gb)
        n_zoom += sum("zoom" in plan.band_kernel_name(i) for i in range(len(gb)))
        outs = [plan.process(x)]
        geo_ok = all(max(b.hop_size for b in gb) % b.hop_size == 0 for b in gb)
        if geo_ok and total > 2:
            outs.append(plan.process_chunked(x, int(rng.integers(1, total))))
        plan.close()
        for got in outs:
            for g, r in zip(got, ref):
                close(g, r)
    assert n_zoom >= 10   # the plans above are band-limited: most bands must have taken the path under test


def test_band_limited_launch_geometries(ux, orc, monkeypatch):
    """The band-limited synthesis' stream tables (every workgroup slot once, centre streams longer than the Ls/Rs ones, shorter
    streams at the signal's edges), the analysis' pairs dealt by age and the priority turns are placement only: against the
    oracle, and against the geometry of rounds 1-2 (UPX_ZOOM_ONCE=0) up to seam rounding; also with a scratch so small that
    a call takes many launch pairs (every launch gets its own stream lists), and with the placement knobs at odd values."""
    x = orc.synthetic_stereo(48000 * 25 + 321, 21)
    edges = [0, 30, 120, 480, 1920, 7680]
    ref_bands = orc.plan_bands(edges, 0.75, orc.win_blackman_harris, 48000, max_block_size=8192)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ref_bands)

    def run(env):
        for k in ("UPX_ZOOM_ONCE", "UPX_ZOOM_SCRATCH_MB", "UPX_ZOOM_C_COST", "UPX_ZOOM_EDGE_PERCENT", "UPX_ZOOM_A_AGE",
                  "UPX_PRIO_YOUNG", "UPX_EDGE_PERCENT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        bands = gpu_chain(ux, edges, 48000, 8192, 32)
        plan = ux.DevicePlan(bands)
        assert "zoom" in plan.band_kernel_name(0) and "zoom" in plan.band_kernel_name(3)
        out = plan.process(x)
        again = plan.process(x)           # the cached tables of the second call
        for u, v in zip(out, again):
            assert np.array_equal(u, v)
        plan.close()
        return out

    base = run({})
    for g, r in zip(base, ref):
        close(g, r)
    # calls of alternating length back to back, no synchronisation in between: the stream tables follow in stream order
    bands = gpu_chain(ux, edges, 48000, 8192, 32)
    plan = ux.DevicePlan(bands)
    n, short = len(x), len(x) - 123457
    d_in = plan.alloc(x.nbytes)
    d_out = [plan.alloc(n * 4) for _ in range(3)]
    plan.h2d(d_in, x)
    for t in (n, short, n, short, n):
        plan.process_device(d_in, t, t, d_out[0], d_out[1], d_out[2], t)
    for d, b in zip(d_out, base):
        got = np.empty(n, dtype=np.float32)
        plan.d2h(got, d)
        assert np.array_equal(got, b)
    plan.close()
    for env in ({"UPX_ZOOM_ONCE": "0"},
                {"UPX_ZOOM_SCRATCH_MB": "1"},                       # ~170 frames of P = 512 per launch pair
                {"UPX_ZOOM_SCRATCH_MB": "1", "UPX_ZOOM_ONCE": "0"},
                {"UPX_ZOOM_C_COST": "0.9", "UPX_ZOOM_EDGE_PERCENT": "50", "UPX_ZOOM_A_AGE": "30"},
                {"UPX_PRIO_YOUNG": "0", "UPX_EDGE_PERCENT": "100", "UPX_ZOOM_EDGE_PERCENT": "100", "UPX_ZOOM_A_AGE": "0"}):
        got = run(env)
        for g, b, r in zip(got, base, ref):
            close(g, r)
            assert float(np.max(np.abs(g - b))) < 1e-6, env          # another cut: seam rounding only
            assert rms(g.astype(np.float64) - b) < 1e-8, env


def test_process_rank_single_gpu(ux, orc):
    """The per-rank entry of the multi-GPU path with world = 1 equals the plain call."""
    from upmix_amd import sharding
    x = orc.synthetic_stereo(150000, 15)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    plan = ux.DevicePlan(bands)
    shard, outs = sharding.process_rank(plan, x, 0, 1)
    assert shard.own_len == len(x)
    for a, b in zip(outs, plan.process(x)):
        assert np.array_equal(a, b)
    plan.close()


def test_pathological_signals(ux, orc):
    """Impulses, DC, full-scale square-ish content, large and small amplitudes: relative 1e-5 RMS vs the oracle."""
    total = 60000
    t = np.arange(total)
    rng = np.random.default_rng(21)
    cases = {
        "impulse": np.stack([(t == 12345) * 1.0, (t == 12345) * 0.5], axis=1),
        "dc": np.stack([np.full(total, 0.25), np.full(total, -0.1)], axis=1),
        "square": np.stack([np.sign(np.sin(2 * np.pi * 440 * t / 48000)), np.sign(np.sin(2 * np.pi * 443 * t / 48000))], axis=1) * 0.9,
        "loud_1e6": rng.standard_normal((total, 2)) * 1e6,
        "quiet_1e-5": rng.standard_normal((total, 2)) * 1e-5,
        "sine_left_only": np.stack([np.sin(2 * np.pi * 1000 * t / 48000), np.zeros(total)], axis=1),
    }
    bands = gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], 48000, 8192, 32)
    ob = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000, max_block_size=8192)
    for name, x in cases.items():
        x = x.astype(np.float32)
        ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
        got = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
        scale = max(float(np.max(np.abs(x))), 1e-30)
        for g, r in zip(got, ref):
            assert np.all(np.isfinite(g)), name
            assert rms((g.astype(np.float64) - r) / scale) <= TOL, name


def test_kernel_flavours_agree(ux, orc, monkeypatch):
    """Wide streams (the fused kernel for STFT 4096 / 8192) and the band-limited two-kernel path (upx_zoom.h; UPX_ZOOM =
    smallest decimation that takes it, 0 = never) are different routings of the same arithmetic: each within tolerance of
    the oracle, and within float32 rounding of one another.  (The plain Stockham schedule and the 8-points-per-lane
    kernels of rounds 1-5 live in experiment builds only: csrc/experiments/.)"""
    x = orc.synthetic_stereo(150000, 21)
    edges, tf = [0, 120, 480, 4000], 32
    ref_bands = orc.plan_bands(edges, 0.75, orc.win_blackman_harris, 48000, max_block_size=8192, threshold_factor=tf)
    assert sorted({b.block_size for b in ref_bands}) == [512, 4096, 8192]
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ref_bands)
    outs, names = {}, {}
    for variant, zoom in (("0", "0"), ("default", None)):
        monkeypatch.setenv("UPX_KERNEL_VARIANT", "2")        # deaf in the product library: no such kernels in it
        if zoom is None:
            monkeypatch.delenv("UPX_ZOOM", raising=False)
        else:
            monkeypatch.setenv("UPX_ZOOM", zoom)
        bands = gpu_chain(ux, edges, 48000, 8192, tf)
        plan = ux.DevicePlan(bands)
        names[variant] = [plan.band_kernel_name(i) for i in range(len(bands))]
        outs[variant] = plan.process(x)
        plan.close()
        for got, r in zip(outs[variant], ref):
            close(got, r)
    assert any("WideCfg<13, 4>" in n for n in names["0"]) and any("WideCfg<12, 4>" in n for n in names["0"])
    # default: the merged 8192 bands (pass band below bin 128: P = 256, D = 32) take the band-limited path,
    # the 4096 band (bins up to 426: P = 1024, D = 4: below the smallest decimation the path is built for) stays fused
    assert any("zoom_synthesis_kernel<upx::ZoomCfg<8, 16, 4>>" in n for n in names["default"])
    assert any("WideCfg<12, 4>" in n for n in names["default"])
    for a, b in zip(outs["0"], outs["default"]):
        assert rms(a.astype(np.float64) - b) < 1e-7


def test_streamed_host_call(ux, orc, monkeypatch):
    """upx_process_chunked: upload / kernels / download of consecutive chunks overlap, each chunk's overlap-add
    tail is added onto the next on the device.  Any chunk length gives the one-shot result up to the float32
    association at the chunk seams; upx_process streams by itself once a signal spans two chunks."""
    x = orc.synthetic_stereo(700001, 31)
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    plan = ux.DevicePlan(bands)
    monkeypatch.setenv("UPX_STREAM_CHUNK", "0")
    base = plan.process(x)
    ref_bands = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=4096, threshold_factor=64)
    ref = orc.extract_multi_band(x[:150000, 0].astype(np.float64), x[:150000, 1].astype(np.float64), ref_bands)
    for chunk in (1, 5000, 65536, 200000, 699999, 1 << 22):
        got = plan.process_chunked(x, chunk)
        for u, v, r in zip(base, got, ref):
            assert np.all(np.isfinite(v))
            assert float(np.max(np.abs(u - v))) < 1e-6, chunk
            close(v[:140000], r[:140000])
    monkeypatch.setenv("UPX_STREAM_CHUNK", "100000")   # upx_process streams from two chunks on
    for u, v in zip(base, plan.process(x)):
        assert float(np.max(np.abs(u - v))) < 1e-6
    # ragged tails: last chunk shorter than the spill, signal shorter than a chunk
    for total in (3, 4097, 12288 + 5, 24576 + 4095):
        xs = x[:total]
        one = plan.process(xs) if total < 200000 else None
        monkeypatch.setenv("UPX_STREAM_CHUNK", "0")
        one = plan.process(xs)
        got = plan.process_chunked(xs, 4096)
        for u, v in zip(one, got):
            assert u.shape == v.shape == (total,) and float(np.max(np.abs(u - v))) < 1e-6, total
    plan.close()


def test_signal_longer_than_one_launch(ux, orc):
    """More than 2^29 samples (a launch's 32-bit byte offsets end there): upx_process streams the signal through
    the device in chunks, so the length is bounded by host memory only.  Oracle windows at the start, across the
    2^29 boundary and at the ragged end."""
    try:
        import psutil
        if psutil.virtual_memory().available < 40 * (1 << 30):
            pytest.skip("needs ~12 GB of host memory")
    except ImportError:
        pass
    total = (1 << 29) + 12345
    base = orc.synthetic_stereo(1 << 22, 41)
    x = np.tile(base, (total // len(base) + 1, 1))[:total]
    x *= np.linspace(0.2, 1.0, total, dtype=np.float32)[:, None]   # no two windows alike
    bands = gpu_chain(ux, [0, 300, 3000], 48000, 4096, 64)
    ref_bands = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=4096, threshold_factor=64)
    got = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], 48000, bands)
    assert all(g.shape == (total,) for g in got)
    span, pad = 60000, 8192    # slice starts on the frame grid of every band; compare what lies a frame inside it
    for start in (0, (1 << 29) - 40960, total - span):
        s0 = max(0, (start - pad) // 4096 * 4096)
        sl = x[s0:min(total, s0 + span + 2 * pad)]
        ref = orc.extract_multi_band(sl[:, 0].astype(np.float64), sl[:, 1].astype(np.float64), ref_bands)
        lo = 0 if s0 == 0 else pad
        hi = len(sl) if s0 + len(sl) == total else len(sl) - pad
        for g, r in zip(got, ref):
            close(g[s0 + lo:s0 + hi], r[lo:hi].astype(np.float32))
    for g in got:
        assert np.all(np.isfinite(g[::4099]))


def test_c4_plan_96k_prefix_vs_oracle(ux, orc):
    """BASELINE configs[3] plan: 96 kHz, same edges -> STFT [8192 x4, 2048, 512], 2 s prefix, seed 3,
    whole and cut into two time shards on one device (the multi-GPU arithmetic without RCCL)."""
    from upmix_amd import sharding
    sr, edges = 96000, [0, 30, 120, 480, 1920, 7680]
    x = orc.synthetic_stereo(2 * sr, 3)
    bands = gpu_chain(ux, edges, sr, 8192, 32)
    ob = orc.plan_bands(edges, 0.75, orc.win_blackman_harris, sr, max_block_size=8192, threshold_factor=32)
    assert [b.block_size for b in bands] == [8192, 8192, 8192, 8192, 2048, 512]
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
    whole = ux.extract_center_left_right_multi_band_in_memory(x[:, 0], x[:, 1], sr, bands)
    for got, r in zip(whole, ref):
        close(got, r)
    plan = ux.DevicePlan(bands)
    sharded = sharding.process_sharded_single_device(plan, x, max_shard=len(x) // 2 + 1)
    for got, r, w in zip(sharded, ref, whole):
        close(got, r)
        assert float(np.max(np.abs(got - w))) < 1e-6
    plan.close()


def test_c4_share_full_size_on_one_device(ux, orc):
    """BASELINE configs[3], one GPU's share at full size: 15 min of 96 kHz stereo (86.4 M samples), plan
    [8192 x4, 2048, 512], as ONE device-resident launch per band (what bench.py --workload c4share times and what
    each of the 8 ranks runs).  The oracle cannot afford the whole signal: windows at the head, in the interior and at
    the tail, plus size-independent properties (time shards + seams == single launch, streamed == single launch,
    silence -> exact zeros)."""
    sr, total = 96000, 86_400_000
    edges = [0, 30, 120, 480, 1920, 7680]
    bands = gpu_chain(ux, edges, sr, 8192, 32)
    ob = orc.plan_bands(edges, 0.75, orc.win_blackman_harris, sr, max_block_size=8192)
    assert [b.block_size for b in bands] == [8192, 8192, 8192, 8192, 2048, 512]
    rng = np.random.default_rng((3, 0))          # C4 shard g = seed (3, g)
    x = np.empty((total, 2), np.float32)
    step = 1 << 23
    for a in range(0, total, step):              # in pieces: no 86.4 M-sample float64 temporaries
        n = min(step, total - a)
        m, sd = rng.standard_normal(n), rng.standard_normal(n)
        x[a:a + n, 0] = 0.1 * (m + 0.5 * sd)
        x[a:a + n, 1] = 0.1 * (m - 0.5 * sd)
    plan = ux.DevicePlan(bands)
    assert "zoom" in plan.band_kernel_name(0) and plan.band_group(0) == (0, 4)   # the four 8192 bands: one band-limited launch
    d_in = plan.alloc(total * 8)
    d_out = [plan.alloc(total * 4) for _ in range(3)]
    out = [np.empty(total, np.float32) for _ in range(3)]
    try:
        plan.h2d(d_in, x)
        plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
        for o, d in zip(out, d_out):
            plan.d2h(o, d)
        assert all(np.all(np.isfinite(o[::97])) for o in out)
        # oracle windows: head (with its fade-in), two interior windows on the hop_max grid, the tail
        n = 60000
        ref = orc.extract_multi_band(x[:n + 8192, 0].astype(np.float64), x[:n + 8192, 1].astype(np.float64), ob)
        for got, r in zip(out, ref):
            close(got[:n], r[:n])
        for a in (2048 * 9000, 2048 * 33333):
            seg = x[a:a + 90000].astype(np.float64)
            ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
            for got, r in zip(out, ref):
                close(got[a + 8192:a + 90000 - 8192], r[8192:90000 - 8192])
        a = (total // 2048 - 40) * 2048
        seg = x[a:].astype(np.float64)
        ref = orc.extract_multi_band(seg[:, 0], seg[:, 1], ob)
        for got, r in zip(out, ref):
            close(got[a + 8192:], r[8192:])
        # windows centred on stream seams of every launch group of THIS call's geometry (the four merged
        # 8192 bands' band-limited pair, the fused 2048 and 512 launches) + seeded offsets: >= 5 % of the 86.4 M samples
        seams = seam_positions(plan, bands, per_group=8, seed=4)
        assert sorted(seams) == [0, 4, 5]
        length, margin = 100_000, 8192
        centred = [p_ - length // 2 for g in sorted(seams) for p_ in seams[g]]               # 24 across seams
        anywhere = [int(v) for v in np.random.default_rng(44).integers(0, total - length, size=30)]
        checked = check_oracle_windows(orc, ob, x, out, centred + anywhere, length, margin, 4096)
        assert checked >= 0.05 * total, checked
        # streamed through the host entry (2^22-sample chunks, seams added on the device) == single launch
        streamed = plan.process(x)
        for u, v in zip(out, streamed):
            assert rms(u.astype(np.float64) - v) < 1e-8
            assert float(np.max(np.abs(u - v))) < 1e-6
        del streamed
        # silence in -> exact zeros out, at full size, on the device-resident path
        plan.memset(d_in, 0, total * 8)
        plan.process_device(d_in, total, total, d_out[0], d_out[1], d_out[2], total)
        z = np.empty(total, np.float32)
        for d in d_out:
            plan.d2h(z, d)
            assert not z.any()
    finally:
        plan.free(d_in)
        for d in d_out:
            plan.free(d)
        plan.close()


def test_golden_band_edge_corner_cases_and_nonfinite_samples(ux, orc):
    """Fixture F8 (generated from the reference): swapped band edges, edges above Nyquist, clipped / zero-width fades;
    NaN and Inf samples; float64 samples beyond the float32 range."""
    from test_oracle_golden import EDGE_BANDS
    z = load_golden("f8_edges.npz")
    x = z["x"]
    for tag, (n, lo, hi, mode, wl, wh) in EDGE_BANDS.items():
        bex = ux.MultiBandExtractorAccu(n, 0.75, ux.make_blackman_harris, lo, hi, 48000, mode, wl, wh)
        for got, k in zip(bex.process_all_blocks(x[:, 0], x[:, 1]), "clr"):
            close(got, z[f"{tag}_{k}"])
        if tag == "both_above_rc":
            assert not any(o.any() for o in bex.process_all_blocks(x[:, 0], x[:, 1]))   # zero gain: exact silence
        bex.close()
    bands = gpu_chain(ux, [0, 3000], 48000, 1024, 32)
    # NaN / Inf: poisoned exactly where the reference is poisoned (the frames of every band that contain the sample,
    # all three outputs), equal to the reference elsewhere
    y = z["nonfinite_x"]
    with np.errstate(all="ignore"):
        out = ux.extract_center_left_right_multi_band_in_memory(y[:, 0], y[:, 1], 48000, bands)
    for got, k in zip(out, "clr"):
        ref = z[f"nonfinite_{k}"]
        bad_ref, bad = ~np.isfinite(ref), ~np.isfinite(got)
        if k == "c":
            # the centre signals of a frame PAIR come out of one inverse transform (C_a + i C_b), so a poisoned frame
            # takes its pair partner along: at most one more hop (256 = hop of the 1024 band) on either side
            grown = bad_ref.copy()
            for sh in range(1, 257):
                grown[sh:] |= bad_ref[:-sh]
                grown[:-sh] |= bad_ref[sh:]
            assert np.all(bad[bad_ref]) and not np.any(bad & ~grown)
        else:
            assert np.array_equal(bad, bad_ref), k
        ok = ~bad
        assert rms(got[ok].astype(np.float64) - ref[ok]) <= TOL
    # Samples beyond the float32 range are outside the boundary's dtype contract (float64 input is cast to float32
    # once on the host): the cast gives +-Inf and the result is non-finite on exactly the frames that
    # contain those samples, in all three outputs (L and R share one complex transform) - where the reference (float64
    # transforms, float32 cast at the end) returns values of magnitude 1e36+, overflows itself, or, in the channel
    # that has no such sample, stays ordinary.  Everything outside those frames equals the reference.
    h = z["huge_x"]
    with np.errstate(all="ignore"):
        out = ux.extract_center_left_right_multi_band_in_memory(h[:, 0], h[:, 1], 48000, bands)
    touched = np.zeros(len(h), bool)                              # frames of either band that contain a huge sample
    for n0 in (4000, 4001, 4002):
        for b in bands:
            n, hop = b.block_size, b.hop_size
            j_lo, j_hi = max(0, -(-(n0 - n + 1) // hop)), n0 // hop
            touched[j_lo * hop:min(len(h), j_hi * hop + n)] = True
    for got, k in zip(out, "clr"):
        ref = z[f"huge_{k}"]
        assert not np.isfinite(got[touched]).any(), k             # never a silently wrong finite number
        if k != "c":
            assert np.all(np.isfinite(got[~touched])), k          # (C: plus the pair partners' hop, see above)
        ok = np.isfinite(got) & ~touched
        assert ok.sum() > len(h) - 4000
        assert rms(got[ok].astype(np.float64) - ref[ok]) <= TOL


def test_baseline_plans_select_the_budgeted_kernels(ux):
    """The kernels tests/test_register_budget.py holds to their register budget ARE the ones the BASELINE plans select
    (bench.py workloads c1, c2, c3, c4share, default): every launch of every such plan names a kernel of that list."""
    import bench
    import importlib.util
    spec = importlib.util.spec_from_file_location("upx_register_budget", os.path.join(os.path.dirname(__file__),
                                                                                     "test_register_budget.py"))
    budget = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(budget)
    known = {bench.canonical_kernel_name(k) for k in list(budget.FUSED) + list(budget.ZOOM)}
    seen = set()
    for wl in ("c1", "c2", "c3", "c4share", "default"):
        sr = bench.WORKLOADS[wl][0]
        bands = bench.workload_bands(
            wl, lambda n, ov, lo, hi, sr_, mode, wlo, whi: ux.MultiBandExtractorAccu(n, ov, ux.make_blackman_harris, lo, hi,
                                                                                    sr_, mode, wlo, whi),
            lambda e, sr_, m, f: ux.chain_bands(e, 0.75, ux.make_blackman_harris, sr_, max_block_size=m, threshold_factor=f,
                                                verbose=False))
        plan = ux.DevicePlan(bands)
        try:
            x = np.zeros((sr, 2), np.float32)
            plan.process(x)                                        # (a call, so that the fill figures exist too)
            for b in range(len(bands)):
                names = [plan.band_phase_kernel_name(b, 1)]
                if plan.band_phase_kernel_name(b, 0):
                    names.append(plan.band_phase_kernel_name(b, 0))
                for n in names:
                    assert bench.canonical_kernel_name(n) in known, (wl, b, n)
                    seen.add(bench.canonical_kernel_name(n))
                fill = plan.band_fill(b)
                assert 0 < fill["workgroups"] and fill["slots"] >= 256
        finally:
            plan.close()
    assert seen == known, known - seen                            # and the list holds nothing no plan selects


@pytest.mark.parametrize("total,uniform", [(400_000, 1), (300_000, 1), (400_000, 0)])
def test_wav_pipeline_chunked_overlap_equals_single_chunk(ux, monkeypatch, total, uniform):
    """Round 4: upx_wav_shard_begin / _finish run the shard chunk by chunk (chunk c's kernels under the upload of chunk
    c + 1, export pieces under the download of the previous piece).  With a small UPX_WAV_CHUNK the same file goes through
    9-12 chunks of equal length (UPX_WAV_UNIFORM=1) or 5 that shrink geometrically (the default geometry): payload, peaks and scale equal the one-chunk pipeline's except <= 1 LSB behind chunk seams (the float32
    association of the overlap-add there, as between the chunks of upx_process_chunked); every mode and codec, a mono
    file, and the sharded form with a caller-applied seam."""
    from upmix_amd import _lib
    rng = np.random.default_rng(41)
    x = np.clip(0.2 * rng.standard_normal((total, 2)), -0.99, 0.99)
    pcm16 = np.rint(x * 32767).astype("<i2")
    f32 = x.astype(np.float32)
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 48000, max_block_size=8192,
                           verbose=False)
    monkeypatch.setenv("UPX_WAV_CHUNK", "0")
    whole = ux.DevicePlan(bands)
    monkeypatch.setenv("UPX_WAV_CHUNK", "32768")          # >= 4 x spill (6144), a multiple of the grid (4096)
    monkeypatch.setenv("UPX_WAV_UNIFORM", str(uniform))
    cut = ux.DevicePlan(bands)
    try:
        for src, in_kind, ch in ((pcm16, _lib.PCM16, 2), (f32, _lib.F32, 2), (pcm16[:, 0].copy(), _lib.PCM16, 1)):
            for mode, out_kind, dt in (("stereo_sum", _lib.PCM16, "<i2"), ("split", _lib.PCM32, "<i4"), ("AB", _lib.F32, "<f4")):
                ref, ref_stats = whole.wav_pipeline(src, in_kind, ch, total, mode, out_kind)
                got, stats = cut.wav_pipeline(src, in_kind, ch, total, mode, out_kind)
                assert stats["peak_in"] == ref_stats["peak_in"]
                assert abs(stats["overall_peak"] - ref_stats["overall_peak"]) <= 1e-6 * ref_stats["overall_peak"]
                for key in ref:
                    a = np.frombuffer(bytes(got[key]), dtype=dt).astype(np.float64)
                    b = np.frombuffer(bytes(ref[key]), dtype=dt).astype(np.float64)
                    assert a.shape == b.shape
                    tol = {"<i2": 1, "<i4": 1 << 17, "<f4": 2e-6}[dt]       # ~1e-7 of full scale behind a seam
                    assert np.max(np.abs(a - b)) <= tol, (mode, key, float(np.max(np.abs(a - b))))
                    if dt == "<i2":
                        assert np.count_nonzero(a != b) <= 64, (mode, key)
        t = cut.wav_pipeline_times_ms()
        assert t["begin"] > 0 and t["finish"] > 0 and 0 <= t["begin_tail"] <= t["begin"]
    finally:
        whole.close()
        cut.close()


def test_streaming_chunks_on_the_device_ring(ux, orc):
    """Round 4: process_stereo_chunk keeps the overlap-add accumulators on the device (upx_stream_chunk: 2 N floats up,
    3 hop floats down per block).  Driven block by block the way the reference drives it (center_extraction.py:449-460:
    one block per hop, flush at the end), it gives the whole-signal result - the oracle's within tolerance, and
    process_all_blocks' to float32 rounding (that path pairs frames for the centre transform; this one transforms one
    frame alone) -, short final blocks included; the accumulator attributes read and write the device state."""
    for n, total in ((1024, 9000), (256, 2500), (4096, 30000)):
        hop = n // 4
        x = orc.synthetic_stereo(total, (9, n))
        bex = ux.MultiBandExtractorAccu(n, 0.75, ux.make_blackman_harris, 300.0, 3000.0, 48000, "raised_cosine", 75.0, 750.0)
        ob = orc.Band(n, 0.75, 300.0, 3000.0, 48000, "raised_cosine", 75.0, 750.0)
        outs = [[], [], []]
        for idx in range(0, total, hop):                      # (the reference pads to whole blocks; short blocks are
            parts = bex.process_stereo_chunk(x[idx:idx + n, 0], x[idx:idx + n, 1])   # zero-extended by the library)
            for o, q in zip(outs, parts):
                assert q.shape == (hop,) and q.dtype == np.float32
                o.append(q)
        assert bex.accumC.shape == (n,) and bex.accumC.any()                        # state lives on, readable
        tail = bex.flush_final()
        assert not bex.accumC.any() and not bex.accumL.any()
        got = [np.concatenate(o + [t])[:total] for o, t in zip(outs, tail)]
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), ob)
        whole = bex.process_all_blocks(x[:, 0], x[:, 1])
        for g, r, w in zip(got, ref, whole):
            close(g, r)
            assert rms(g.astype(np.float64) - w) < 1e-7
        # a caller that presets an accumulator (attribute assignment) sees it emitted with the next block
        bex.accumL = np.full(n, 0.25, np.float32)
        c, l, r = bex.process_stereo_chunk(np.zeros(n, np.float32), np.zeros(n, np.float32))
        assert np.all(l == 0.25) and not c.any()
        assert np.all(bex.accumL[:n - hop] == 0.25) and not bex.accumL[n - hop:].any()
        bex.close()


def test_wav_pipeline_beyond_2_to_29_frames(ux):
    """VERDICT r3 missing 4: the device codec refused files of 2^29 frames or more (configs[3] taken literally on ONE GPU is
    691 M frames).  The shard now runs in chunks, so only a chunk has to fit a launch's 2^29 - 1 samples.  2^29 + 300 000
    mono 16-bit frames (1.07 GB in, 2.1 GB out), silent except for a burst that straddles frame 2^29: the output is silent
    where the input is (exact zeros), and the burst region equals the small file that holds the burst alone (the global
    peaks - hence the scale - are the burst's in both)."""
    from upmix_amd import _lib
    big = (1 << 29) + 300_000
    lo, hi = (1 << 29) - 150_000, (1 << 29) + 150_000          # burst: 300 000 frames across the old limit
    rng = np.random.default_rng(99)
    burst = (rng.standard_normal(hi - lo) * 4000).astype("<i2")
    bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 48000, max_block_size=8192,
                           verbose=False)
    plan = ux.DevicePlan(bands)
    try:
        pcm = np.zeros(big, dtype="<i2")
        pcm[lo:hi] = burst
        out, stats = plan.wav_pipeline(pcm, _lib.PCM16, 1, big, "stereo_sum", _lib.PCM16)
        got = out["Sum"].view("<i2").reshape(-1, 2)
        assert got.shape == (big, 2)
        assert not got[:lo - 8192].any() and not got[hi + 8192:].any()          # silence stays silence (+- one frame)
        # the burst alone, padded so that its frames sit on the same hop grid (lo is a multiple of 2 hop_max = 4096? no:
        # align the small file's start to the grid below lo)
        a = (lo - 16384) // 4096 * 4096
        small = np.zeros(hi + 16384 - a, dtype="<i2")
        small[lo - a:hi - a] = burst
        ref, ref_stats = plan.wav_pipeline(small, _lib.PCM16, 1, len(small), "stereo_sum", _lib.PCM16)
        want = ref["Sum"].view("<i2").reshape(-1, 2)
        assert abs(stats["scale_factor"] - ref_stats["scale_factor"]) <= 1e-6 * ref_stats["scale_factor"]
        seg = got[a:a + len(small)].astype(np.int32)
        assert np.max(np.abs(seg - want.astype(np.int32))) <= 1                 # chunk seams fall elsewhere: <= 1 LSB
        assert np.count_nonzero(seg != want) <= 2000
        del out, got, ref, want
    finally:
        plan.close()


def test_streamed_shard_api_misuse_and_piecewise_feed(ux):
    """upx_wav_shard_open / _feed / _seal and _finish_async / _wait_piece: pieces of any length give the one-call result
    (begin + finish) bit for bit when the chunk cut is the same, and misuse fails with ValueError instead of touching
    memory: feeding without an open shard, too many frames, sealing early, waiting for a piece that does not exist."""
    from upmix_amd import _lib
    rng = np.random.default_rng(8)
    n = 200_000
    pcm = np.rint(np.clip(0.2 * rng.standard_normal((n, 2)), -0.99, 0.99) * 32767).astype("<i2")
    raw = pcm.view(np.uint8).reshape(-1)
    bands = ux.chain_bands([0, 300, 3000], 0.75, ux.make_blackman_harris, 48000, max_block_size=4096, threshold_factor=64,
                           verbose=False)
    plan = ux.DevicePlan(bands)
    try:
        ref, stats = plan.wav_pipeline(pcm, _lib.PCM16, 2, n, "split", _lib.PCM24)
        with pytest.raises(ValueError):
            plan.wav_shard_feed(raw[:400], 100)                              # nothing open
        plan.wav_shard_open(_lib.PCM16, 2, n, n, n)
        with pytest.raises(ValueError):
            plan.wav_shard_seal()                                            # nothing fed yet
        pos = 0
        for piece in (1, 999, 65536, 7, 100_000, n):                         # ragged pieces, the last one clipped
            k = min(piece, n - pos)
            if k:
                plan.wav_shard_feed(raw[pos * 4:(pos + k) * 4], k)
                pos += k
        with pytest.raises(ValueError):
            plan.wav_shard_feed(raw[:4], 1)                                  # one frame too many
        pin, pout = plan.wav_shard_seal()
        assert pin == stats["peak_in"] and abs(pout - stats["overall_peak"]) <= 1e-7 * stats["overall_peak"]
        payloads, n_pieces, per = plan.wav_shard_finish_async(stats["scale_factor"], "split", _lib.PCM24, n, piece_frames=30_000)
        assert n_pieces == 7 and per == 30_000
        with pytest.raises(ValueError):
            plan.wav_shard_wait_piece(7)
        for k in range(n_pieces):
            plan.wav_shard_wait_piece(k)
            for key in ref:                                                  # piece k is final as soon as it has landed
                a, b = k * per * 6, min(n, (k + 1) * per) * 6
                assert bytes(payloads[key][a:b]) == bytes(ref[key][a:b]), (key, k)
        with pytest.raises(ValueError):
            plan.wav_shard_finish_async(1.0, "split", _lib.PCM24, n)        # no shard open any more
    finally:
        plan.close()


def test_wav_pipeline_full_c4_share_chunked_equals_whole(ux, orc, monkeypatch):
    """BASELINE configs[3], one GPU's share (86.4 M frames at 96 kHz, plan [8192 x4, 2048, 512]) through the device codec:
    the default chunk schedule (the shard in ~10 chunks, their kernels under the uploads) against the same call as ONE chunk -
    identical peaks and scale, payloads equal except <= 1 LSB behind chunk seams - and the payload against the in-memory
    entry + host export on a window (the codec around the kernels: PCM16 in, stereo_sum PCM16 out)."""
    from upmix_amd import _lib, export
    total, sr = 86_400_000, 96000
    rng = np.random.default_rng(13)
    pcm = np.empty((total, 2), dtype="<i2")
    for a in range(0, total, 1 << 23):                       # (in pieces: no 1.4 GB float64 temporary)
        b = min(total, a + (1 << 23))
        pcm[a:b] = np.rint(np.clip(0.2 * rng.standard_normal((b - a, 2), dtype=np.float32), -0.99, 0.99) * 32767).astype("<i2")
    bands = gpu_chain(ux, [0, 30, 120, 480, 1920, 7680], sr, 8192, 32)
    assert [b.block_size for b in bands] == [8192, 8192, 8192, 8192, 2048, 512]
    monkeypatch.setenv("UPX_WAV_CHUNK", "0")
    whole = ux.DevicePlan(bands)
    monkeypatch.delenv("UPX_WAV_CHUNK")
    cut = ux.DevicePlan(bands)
    try:
        ref, ref_stats = whole.wav_pipeline(pcm, _lib.PCM16, 2, total, "stereo_sum", _lib.PCM16)
        got, stats = cut.wav_pipeline(pcm, _lib.PCM16, 2, total, "stereo_sum", _lib.PCM16)
        assert stats["peak_in"] == ref_stats["peak_in"]
        assert abs(stats["overall_peak"] - ref_stats["overall_peak"]) <= 1e-6 * ref_stats["overall_peak"]
        a = got["Sum"].view("<i2").astype(np.int32)
        b = ref["Sum"].view("<i2").astype(np.int32)
        assert a.shape == b.shape == (2 * total,)
        # (every launch of every chunk cuts its streams anew: ~1e-7 association differences behind tens of thousands of stream
        # seams, a few of which cross a 16-bit rounding boundary: 7 048 of 172.8 M samples when this was written)
        assert np.max(np.abs(a - b)) <= 1 and np.count_nonzero(a != b) <= a.size // 10000
        # a window of the payload against the in-memory kernels + the host's export arithmetic (main.py:85-97, :140-157)
        lo, n = 40_000_000 // 4096 * 4096, 400_000
        seg = pcm[lo - 65536:lo + n + 65536].astype(np.float32) / 32768.0
        c, l, r = cut.process(seg)
        scale = np.float64(stats["scale_factor"])
        for p in (c, l, r):
            p *= scale
        want = export.export_arrays("stereo_sum", c, l, r)["Sum"][65536:65536 + n]
        q = np.clip(np.rint(want.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int32)
        seg_got = got["Sum"].view("<i2").reshape(-1, 2)[lo:lo + n].astype(np.int32)
        assert np.max(np.abs(seg_got - q)) <= 1                                     # (the restarted window fades in: 65536 skipped)
        del ref, got, a, b
    finally:
        whole.close()
        cut.close()
