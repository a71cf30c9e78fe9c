"""
CPU check of the KERNEL SOURCE: upmix_amd/csrc/upx_core.h compiled for the host
with a sequential workgroup executor (tests/emu/emu.cpp) and compared with the
oracle.  Catches indexing / LDS layout / framing bugs without a GPU; it is not a
product path (upmix_amd never loads it).
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rms
from oracle import upmix_oracle as orc

fp = ctypes.POINTER(ctypes.c_float)
PTS = [16]   # points per lane used by run_emu (switched by the `pts` fixture)


WIDE = "wide"   # wide streams for N = 4096 / 8192 (the library's default there), 16 points per lane otherwise


@pytest.fixture(autouse=True, params=[16, 8, WIDE], ids=["P16", "P8", "wide"])
def pts(request):
    PTS[0] = request.param
    return request.param


@pytest.fixture(scope="module")
def emu():
    import __graft_entry__ as ge
    path = ge.build_emulator()
    lib = ctypes.CDLL(path)
    lib.emu_band.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_longlong, fp, fp, fp, ctypes.c_longlong,
                             fp, fp, fp] + [ctypes.c_int] * 7
    lib.emu_band.restype = ctypes.c_int
    return lib


def P(a):
    return a.ctypes.data_as(fp)


def run_emu(lib, band, x, blocks_per_stream, outs=None, accumulate=0, own_len=None, t_out=None, pts=None,
            gain_table=None):
    pts = pts or PTS[0]
    n, hop = band.block_size, band.hop_size
    if pts == WIDE:
        pts = 0 if n in (4096, 8192) else 16
    k = n // hop
    t_in = len(x)
    own = t_in if own_len is None else own_len
    t_out = t_in if t_out is None else t_out
    j_hi = -(-own // hop)
    m_hi = min(j_hi + k - 1, -(-t_out // hop)) if accumulate else -(-t_out // hop)
    w_a = np.ascontiguousarray(band.analysis_window)
    w_s = (band.synthesis_window / np.float32(n)).astype(np.float32)
    gain = (0.5 * orc.band_gain(band)).astype(np.float32) if gain_table is None else gain_table
    n_gain = 1 if gain_table is None else gain_table.shape[0]
    gain = np.ascontiguousarray(gain)
    if outs is None:
        outs = [np.full(t_out, np.nan, np.float32) for _ in range(3)]
    xin = np.ascontiguousarray(x, dtype=np.float32)
    rc = lib.emu_band(int(np.log2(n)), k, pts, P(xin), t_in, P(outs[0]), P(outs[1]), P(outs[2]), t_out, P(w_a), P(w_s),
                      P(gain), 0, j_hi, 0, m_hi, blocks_per_stream, accumulate, n_gain)
    assert rc == 0
    return outs


CASES = [  # N, T, F, f_low, f_high, width_low, width_high
    (256, 3000, 5, 7680., 24000., 480., 6000.),
    (512, 5000, 9, 0., 24000., 0., 6000.),
    (1024, 9000, 7, 1920., 7680., 480., 1920.),
    (2048, 12345, 6, 0., 24000., 0., 6000.),
    (4096, 30000, 4, 480., 1920., 120., 480.),
    (8192, 40000, 3, 120., 480., 30., 120.),
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}" for c in CASES])
def test_kernel_source_matches_oracle(emu, case):
    n, total, f, lo, hi, wl, wh = case
    band = orc.Band(n, 0.75, lo, hi, 48000, "raised_cosine", wl, wh)
    x = orc.synthetic_stereo(total, n)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    got = run_emu(emu, band, x, f)
    for g, r in zip(got, ref):
        assert not np.isnan(g).any()
        assert rms(g.astype(np.float64) - r) < 1e-7


def test_other_overlaps(emu):
    for n, ov in ((256, 0.5), (1024, 0.5), (1024, 0.875), (4096, 0.875), (8192, 0.5)):
        band = orc.Band(n, ov, 200., 8000., 44100, "raised_cosine", 50., 2000., window=orc.win_hann)
        x = orc.synthetic_stereo(4 * n + 123, 3)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
        got = run_emu(emu, band, x, 5)
        for g, r in zip(got, ref):
            assert rms(g.astype(np.float64) - r) < 1e-7, (n, ov)


def test_stream_partition_changes_only_seam_rounding(emu):
    """Streams recompute nothing: a different cut only moves the seams, where the float32 association of the
    overlap-add differs (a few ulp on the K-1 blocks after each seam); everything else is bit-identical."""
    band = orc.Band(1024, 0.75, 300., 3000., 48000, "raised_cosine", 75., 750.)
    x = orc.synthetic_stereo(20000, 5)
    base = run_emu(emu, band, x, 1000)
    for f in (4, 6, 18):
        for a, b in zip(base, run_emu(emu, band, x, f)):
            assert float(np.max(np.abs(a - b))) < 1e-7, f
            differ = np.nonzero(a != b)[0]
            seam_blocks = set()
            for m in range(-1 + f, 80, f):
                seam_blocks.update(range(m, m + 3))
            assert all((int(i) // 256) in seam_blocks for i in differ), f


def test_band_accumulation_order(emu):
    """accumulate=1 adds onto the planes written by the previous band (band sum in list order)."""
    bands = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=1024)
    x = orc.synthetic_stereo(9000, 6)
    outs = None
    for i, b in enumerate(bands):
        outs = run_emu(emu, b, x, 6, outs=outs, accumulate=1 if i else 0)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), bands)
    for g, r in zip(outs, ref):
        assert rms(g.astype(np.float64) - r) < 1e-7


def test_interior_workgroups(emu):
    """band_program's interior flavour (upx_core.h: branch-free loop body, rotated phases, loads issued early) must run for
    the workgroups in the middle of a signal and give the oracle's result, overwriting (first band) and accumulating; the
    first and the last workgroups of a launch must NOT take it (their streams hold frames that do not exist)."""
    emu.emu_last_interior_wgs.restype = ctypes.c_longlong
    for n, total, f in ((256, 6000, 4), (1024, 30000, 4), (4096, 90000, 2)):
        bands = orc.plan_bands([0, 700, 5000], 0.75, orc.win_blackman_harris, 48000, max_block_size=n)
        bands = [b for b in bands if b.block_size == n][:2]
        assert len(bands) == 2
        x = orc.synthetic_stereo(total, n + 1)
        outs = None
        for i, b in enumerate(bands):
            outs = run_emu(emu, b, x, f, outs=outs, accumulate=1 if i else 0)
            hop = b.hop_size
            streams = -(-(-(-total // hop) + 1) // (f + (f & 1)))
            per_wg = max(64 // (n // 16), 1) if PTS[0] != 8 else max(64 // (n // 8), 1)
            wgs = -(-streams // per_wg)
            inner = emu.emu_last_interior_wgs()
            assert 0 < inner <= wgs - 2, (n, inner, wgs)
        ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), bands)
        for g, r in zip(outs, ref):
            assert not np.isnan(g).any()
            assert rms(g.astype(np.float64) - r) < 1e-7, n


def test_short_and_ragged_inputs(emu):
    band = orc.Band(2048, 0.75, 0., 24000., 48000, "raised_cosine", 0., 6000.)
    for total in (1, 511, 512, 513, 1000, 2048, 2049):
        x = orc.synthetic_stereo(total, total)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
        got = run_emu(emu, band, x, 3)
        for g, r in zip(got, ref):
            assert g.shape == (total,) and rms(g.astype(np.float64) - r) < 1e-7


def test_shard_arguments(emu):
    """own_len / t_out: only frames starting in the owned range, output spills N-hop past it."""
    band = orc.Band(1024, 0.75, 300., 3000., 48000, "raised_cosine", 75., 750.)
    x = orc.synthetic_stereo(8192 + 768, 7)
    own, t_out = 8192, 8192 + 768
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band, own_len=own, out_len=t_out)
    got = run_emu(emu, band, x, 4, own_len=own, t_out=t_out)
    for g, r in zip(got, ref):
        assert rms(g.astype(np.float64) - r) < 1e-7
    assert rms(ref[0][own:]) > 0   # the spill is not empty


def test_silence_is_exact_zero(emu):
    band = orc.Band(256, 0.75, 7680., 24000., 48000, "raised_cosine", 480., 6000.)
    got = run_emu(emu, band, np.zeros((3000, 2), np.float32), 5)
    assert all(not g.any() for g in got)


def run_unfused(emu, band, x, ch):
    emu.emu_big_band.argtypes = [ctypes.c_int, ctypes.c_int, fp, ctypes.c_longlong, fp, fp, fp, ctypes.c_longlong,
                                 fp, fp, fp] + [ctypes.c_int] * 7
    n, hop, total = band.block_size, band.hop_size, len(x)
    blocks = -(-total // hop)
    w_a = np.ascontiguousarray(band.analysis_window)
    w_s = (band.synthesis_window / np.float32(n)).astype(np.float32)
    gain = (0.5 * orc.band_gain(band)).astype(np.float32)
    outs = [np.full(total, np.nan, np.float32) for _ in range(3)]
    xin = np.ascontiguousarray(x)
    rc = emu.emu_big_band(int(np.log2(n)), hop, P(xin), total, P(outs[0]), P(outs[1]), P(outs[2]), total, P(w_a),
                          P(w_s), P(gain), 0, blocks, 0, blocks, ch, 0, 1)
    assert rc == 0
    return outs


def test_large_stft_four_step_path(emu):
    """STFT 16384..65536 (upx_big.h): chunked four-step transform, vs the oracle."""
    if PTS[0] != 16:
        pytest.skip("the unfused path has one build")
    for n, total, ch, lo, hi, wl, wh, ov in ((16384, 70000, 8, 120., 480., 30., 120., 0.75),
                                             (65536, 150000, 6, 0., 30., 0., 7.5, 0.75),
                                             (32768, 100000, 8, 30., 120., 7.5, 30., 0.75),
                                             (16384, 50000, 6, 120., 480., 30., 120., 0.5)):
        band = orc.Band(n, ov, lo, hi, 48000, "raised_cosine", wl, wh)
        x = orc.synthetic_stereo(total, n)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
        for g, r in zip(run_unfused(emu, band, x, ch), ref):
            assert not np.isnan(g).any()
            assert rms(g.astype(np.float64) - r) < 1e-7, n


def test_arbitrary_hops_unfused_path(emu):
    """Hops that do not divide N (hop = int(N (1 - overlap)), center_extraction.py:252) and K = 16."""
    if PTS[0] != 16:
        pytest.skip("the unfused path has one build")
    for n, ov, wname, total, ch in ((64, 0.75, "hann", 2000, 40), (128, 0.5, "blackman_harris", 3000, 20),
                                    (512, 0.6, "hamming", 6000, 12), (1024, 0.7, "hann", 9000, 10),
                                    (256, 0.9375, "hann", 3000, 40), (2048, 0.35, "sqrt_hann", 12000, 6),
                                    (4096, 0.8, "blackman_harris", 30000, 12), (16384, 0.6, "hann", 80000, 8)):
        band = orc.Band(n, ov, 200., 8000., 44100, "raised_cosine", 50., 2000., window=orc.WINDOWS[wname])
        x = orc.synthetic_stereo(total, n + 1)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
        for g, r in zip(run_unfused(emu, band, x, ch), ref):
            assert not np.isnan(g).any()
            assert rms(g.astype(np.float64) - r) < 1e-7, (n, ov, band.hop_size)


def merged_gain_table(bands):
    """gain[q][k]: the non-zero half-gains of bin k in band order, zero padded (what the library builds)."""
    g = np.stack([0.5 * orc.band_gain(b) for b in bands]).astype(np.float32)
    slots = int(max(1, (g != 0).sum(axis=0).max()))
    table = np.zeros((slots, g.shape[1]), np.float32)
    for k in range(g.shape[1]):
        nz = g[:, k][g[:, k] != 0]
        table[:len(nz), k] = nz
    return table


def test_merged_bands_equal_band_sum(emu):
    """Bands sharing N/hop/windows run as ONE launch with a per-bin gain list; equals the sum of the separate bands."""
    bands = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000, max_block_size=1024)[:4]
    assert [b.block_size for b in bands] == [1024] * 4
    x = orc.synthetic_stereo(9000, 8)
    table = merged_gain_table(bands)
    assert table.shape[0] == 2            # neighbouring raised-cosine bands overlap pairwise
    got = run_emu(emu, bands[0], x, 6, gain_table=table)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), bands)
    for g, r in zip(got, ref):
        assert rms(g.astype(np.float64) - r) < 1e-7


# ---- live-slot flavours (upx::Live<S0, S1>: pruned mask / parking / butterflies) --------------------------------
def live_slots(band, lanes):
    """Own-bin slots [s0, s1) that carry gain (slot s = bins lane + s * lanes), s1 = 8 if the Nyquist bin does."""
    g = orc.band_gain(band)
    n = band.block_size
    live = [bool(np.any(g[s * lanes:(s + 1) * lanes] != 0)) for s in range(8)]
    s0 = min(i for i, v in enumerate(live) if v)
    s1 = max(i for i, v in enumerate(live) if v) + 1
    if g[n // 2] != 0:
        s1 = 8
    return s0, s1


LIVE_CASES = [  # N, overlap, f_low, f_high, width_low, width_high, sr, instantiated (s0, s1)
    (1024, 0.75, 1920., 7680., 480., 1920., 48000, (0, 4)),    # BASELINE configs[2] band 5: bins 31..205
    (1024, 0.75, 1920., 7000., 480., 1500., 48000, (0, 3)),
    (1024, 0.75, 1920., 5000., 480., 100., 48000, (0, 2)),
    (2048, 0.75, 1920., 7680., 480., 1920., 96000, (0, 2)),    # configs[3] band 5
    (2048, 0.75, 1920., 7680., 480., 1920., 96000, (0, 3)),    # ... through a wider instantiation
    (2048, 0.75, 1920., 14000., 480., 2000., 96000, (0, 4)),
    (256, 0.75, 7680., 24000., 1920., 6000., 48000, (1, 8)),   # configs[2] band 6 (Nyquist live)
    (512, 0.75, 9000., 48000., 1920., 12000., 96000, (1, 8)),
    (1024, 0.75, 6000., 24000., 1000., 6000., 48000, (1, 8)),
    (1024, 0.75, 0., 24000., 0., 6000., 48000, (0, 8)),        # full range through the same entry point
    (256, 0.75, 0., 6000., 0., 500., 48000, (0, 5)),           # DC live, Nyquist dead
    (1024, 0.5, 4000., 15000., 400., 800., 48000, (1, 6)),
    (1024, 0.875, 40., 2000., 10., 200., 48000, (0, 1)),
    (2048, 0.75, 3800., 5300., 300., 300., 48000, (1, 2)),
]


@pytest.mark.parametrize("case", LIVE_CASES, ids=[f"N{c[0]}_{c[7][0]}_{c[7][1]}" for c in LIVE_CASES])
def test_live_slot_flavours(emu, case, pts):
    """The pruned flavour must give what the general one gives (skipped terms are exact zeros) and the oracle's result,
    for first bands and accumulating ones, interior and edge workgroups."""
    if pts != 16:
        pytest.skip("live-slot flavours exist for 16 points per lane")
    n, ov, lo, hi, wl, wh, sr, (s0, s1) = case
    band = orc.Band(n, ov, lo, hi, sr, "raised_cosine", wl, wh)
    lanes = n // 16
    a0, a1 = live_slots(band, lanes)
    assert s0 <= a0 and a1 <= s1, ("the instantiation must cover the live slots", (a0, a1))
    emu.emu_band_live.argtypes = [ctypes.c_int] * 4 + [fp, ctypes.c_longlong, fp, fp, fp, ctypes.c_longlong, fp, fp, fp] + \
        [ctypes.c_int] * 6
    emu.emu_band_live.restype = ctypes.c_int
    hop = band.hop_size
    k = n // hop
    total = 40 * hop + 77
    x = orc.synthetic_stereo(total, n + s1)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    w_a = np.ascontiguousarray(band.analysis_window)
    w_s = (band.synthesis_window / np.float32(n)).astype(np.float32)
    gain = np.ascontiguousarray((0.5 * orc.band_gain(band)).astype(np.float32))
    xin = np.ascontiguousarray(x, dtype=np.float32)
    j_hi = -(-total // hop)
    for accumulate in (0, 1):
        base = [np.full(total, 0.25 if accumulate else np.nan, np.float32) for _ in range(3)]
        m_hi = min(j_hi + k - 1, -(-total // hop)) if accumulate else -(-total // hop)
        general = run_emu(emu, band, x, 6, outs=[b.copy() for b in base], accumulate=accumulate, pts=16)
        got = [b.copy() for b in base]
        rc = emu.emu_band_live(int(np.log2(n)), k, s0, s1, P(xin), total, P(got[0]), P(got[1]), P(got[2]), total, P(w_a),
                               P(w_s), P(gain), 0, j_hi, 0, m_hi, 6, accumulate)
        assert rc == 0
        for g, q, r in zip(got, general, ref):
            assert not np.isnan(g).any()
            # (the single-band flavour evaluates the mask through mask_weight: same numbers up to float32 rounding)
            assert float(np.max(np.abs(g - q))) <= 3e-7, "pruned flavour differs from the general one"
            assert rms(g.astype(np.float64) - (r + (0.25 if accumulate else 0.0))) < 1e-7


# ---- streams of unequal length (BandArgs::stream_m0: shorter streams for the signal-edge workgroups) ---------------
@pytest.mark.parametrize("n,k", [(256, 4), (1024, 4), (2048, 4), (512, 2)])
def test_streams_of_unequal_length(emu, n, k, pts):
    """Any cut of the frame range into streams (even lengths, equal inside a workgroup) gives the same band output up to the
    float32 association behind the stream seams; first band and accumulating."""
    if pts != 16:
        pytest.skip("one executor suffices")
    emu.emu_band_uneven.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int, fp,
                                    ctypes.c_longlong, fp, fp, fp, ctypes.c_longlong, fp, fp, fp] + [ctypes.c_int] * 5
    emu.emu_band_uneven.restype = ctypes.c_int
    ov = 1.0 - 1.0 / k
    band = orc.Band(n, ov, 300., 9000., 48000, "raised_cosine", 75., 2000.)
    hop = band.hop_size
    g = 64 // (n // 16) if n // 16 < 64 else 1               # streams per workgroup
    wg_frames = [6, 10, 8, 10, 4]                             # a short first workgroup, a short last one
    total = (sum(wg_frames) * g - 1) * hop - 37               # frames -1 .. cover the signal; ragged end
    x = orc.synthetic_stereo(total, n + k)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    w_a = np.ascontiguousarray(band.analysis_window)
    w_s = (band.synthesis_window / np.float32(n)).astype(np.float32)
    gain = np.ascontiguousarray((0.5 * orc.band_gain(band)).astype(np.float32))
    xin = np.ascontiguousarray(x, dtype=np.float32)
    frames = (ctypes.c_int * len(wg_frames))(*wg_frames)
    j_hi = -(-total // hop)
    for accumulate in (0, 1):
        outs = [np.full(total, 0.5 if accumulate else np.nan, np.float32) for _ in range(3)]
        m_hi = min(j_hi + k - 1, -(-total // hop)) if accumulate else -(-total // hop)
        rc = emu.emu_band_uneven(int(np.log2(n)), k, frames, len(wg_frames), P(xin), total, P(outs[0]), P(outs[1]),
                                 P(outs[2]), total, P(w_a), P(w_s), P(gain), 0, j_hi, 0, m_hi, accumulate)
        assert rc == 0
        for got, r in zip(outs, ref):
            assert not np.isnan(got).any()
            assert rms(got.astype(np.float64) - (r + (0.5 if accumulate else 0.0))) < 1e-7
