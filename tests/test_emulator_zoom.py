"""
CPU check of the band-limited ("zoom") kernel source, upmix_amd/csrc/upx_zoom.h, compiled for the host with the
wave-by-wave executor of tests/emu/emu.cpp (waves of a workgroup run one after the other between barriers, so a
missing barrier shows up as a read of poisoned LDS) and compared with the oracle.  Test infrastructure only.
"""
import ctypes

import numpy as np
import pytest

from conftest import rms
from oracle import upmix_oracle as orc

fp = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def emu():
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_emulator())
    lib.emu_zoom_band.argtypes = [ctypes.c_int] * 3 + [fp, ctypes.c_longlong, fp, fp, fp, ctypes.c_longlong, fp, fp, fp] + [ctypes.c_int] * 8
    lib.emu_zoom_band.restype = ctypes.c_int
    ip = ctypes.POINTER(ctypes.c_int)
    lib.emu_zoom_band_streams.argtypes = lib.emu_zoom_band.argtypes + [ctypes.c_int, ip, ctypes.c_int, ip, ctypes.c_int]
    lib.emu_zoom_band_streams.restype = ctypes.c_int
    lib.emu_zoom_set_deal.argtypes = [ctypes.c_int, ip, ctypes.c_int]
    lib.emu_zoom_set_deal.restype = None
    return lib


def P(a):
    return a.ctypes.data_as(fp)


def zoom_p(gain_rows):
    """Smallest power of two > 2 kmax (at least 256): what the library picks."""
    kmax = int(np.max(np.nonzero(np.any(gain_rows != 0, axis=0))[0]))
    p = 256
    while p < 2 * (kmax + 1):
        p *= 2
    return p


def run_zoom(lib, band, x, blocks_per_stream, outs=None, accumulate=0, own_len=None, t_out=None, gain_table=None,
             log2p=None, pairs_per_wg=3, c_split=1, tables=None):
    """tables = (first frames of the Ls/Rs streams, of the centre streams), n + 1 entries each (ZoomArgs::stream_m0)"""
    n, hop = band.block_size, band.hop_size
    k = n // hop
    t_in = len(x)
    own = t_in if own_len is None else own_len
    t_out = t_in if t_out is None else t_out
    j_hi = -(-own // hop)
    m_hi = min(j_hi + k - 1, -(-t_out // hop)) if accumulate else -(-t_out // hop)
    w_a = np.ascontiguousarray(band.analysis_window)
    w_s = (band.synthesis_window / np.float32(n)).astype(np.float32)
    gain = (0.5 * orc.band_gain(band)).astype(np.float32)[None, :] if gain_table is None else gain_table
    gain = np.ascontiguousarray(gain)
    p = zoom_p(gain) if log2p is None else 1 << log2p
    assert n // p >= 4, "not a zoom case"
    if outs is None:
        outs = [np.full(t_out, np.nan, np.float32) for _ in range(3)]
    xin = np.ascontiguousarray(x, dtype=np.float32)
    ip = ctypes.POINTER(ctypes.c_int)
    if tables is None:
        tab = (None, 0, None, 0)
    else:
        t_lr, t_c = (np.ascontiguousarray(t, dtype=np.int32) for t in tables)
        tab = (t_lr.ctypes.data_as(ip), len(t_lr) - 1, t_c.ctypes.data_as(ip), len(t_c) - 1)
    rc = lib.emu_zoom_band_streams(int(np.log2(n)), k, int(np.log2(p)), P(xin), t_in, P(outs[0]), P(outs[1]), P(outs[2]),
                                   t_out, P(w_a), P(w_s), P(gain), 0, j_hi, 0, m_hi, blocks_per_stream, accumulate,
                                   gain.shape[0], pairs_per_wg, c_split, *tab)
    assert rc == 0, rc
    return outs


CASES = [  # N, T, F, f_low, f_high, width_low, width_high   -> (P, D, RG)
    (2048, 12345, 6, 480., 1920., 120., 480.),      # bins 15..102: P 256, D 8
    (4096, 30000, 4, 480., 1920., 120., 480.),      # bins 31..205: P 512, D 8
    (8192, 40000, 4, 120., 480., 30., 120.),        # bins 15..102: P 256, D 32 (two residue groups)
    (8192, 40000, 6, 480., 1920., 120., 480.),      # bins 61..410: P 1024, D 8
    (16384, 70000, 4, 120., 480., 30., 120.),       # P 512, D 32
    (65536, 150000, 2, 0., 30., 0., 7.5),           # bins 0..51: P 256, D 256 (sixteen residue groups)
]


@pytest.mark.parametrize("case", CASES, ids=[f"N{c[0]}_{int(c[3])}" for c in CASES])
def test_zoom_kernel_source_matches_oracle(emu, case):
    n, total, f, lo, hi, wl, wh = case
    band = orc.Band(n, 0.75, lo, hi, 48000, "raised_cosine", wl, wh)
    x = orc.synthetic_stereo(total, n)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    got = run_zoom(emu, band, x, f)
    for g, r in zip(got, ref):
        assert not np.isnan(g).any()
        assert rms(g.astype(np.float64) - r) < 1e-7


def test_other_overlaps(emu):
    for n, ov, lo, hi in ((4096, 0.5, 100., 700.), (8192, 0.875, 100., 700.), (16384, 0.5, 30., 300.), (2048, 0.875, 400., 1500.)):
        band = orc.Band(n, ov, lo, hi, 44100, "raised_cosine", 25., 150., window=orc.win_hann)
        x = orc.synthetic_stereo(4 * n + 123, 3)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
        got = run_zoom(emu, band, x, 6)
        for g, r in zip(got, ref):
            assert not np.isnan(g).any()
            assert rms(g.astype(np.float64) - r) < 1e-7, (n, ov)


def test_stream_partition_changes_only_seam_rounding(emu):
    """Streams recompute nothing: another cut only moves the seams (float32 association of the overlap-add on the
    K-1 blocks after each seam); the analysis grid (workgroups per XCD label) changes nothing at all."""
    band = orc.Band(4096, 0.75, 480., 1920., 48000, "raised_cosine", 120., 480.)
    x = orc.synthetic_stereo(60000, 5)
    base = run_zoom(emu, band, x, 1000)
    for ppw in (1, 2, 5):
        for a, b in zip(base, run_zoom(emu, band, x, 1000, pairs_per_wg=ppw)):
            assert np.array_equal(a, b)
    for f in (4, 6, 18):
        for a, b in zip(base, run_zoom(emu, band, x, f)):
            assert float(np.max(np.abs(a - b))) < 1e-7, f
            differ = np.nonzero(a != b)[0]
            seam_blocks = set()
            for m in range(-1 + f, 80, f):
                seam_blocks.update(range(m, m + 3))
            assert all((int(i) // 1024) in seam_blocks for i in differ), f


def test_centre_streams_of_their_own_length(emu):
    """ZoomArgs::blocks_per_stream_c: the centre plane cut into 2 or 4 streams per Ls/Rs stream.  Ls / Rs do not change at
    all; the centre differs from the unsplit run only on the K-1 blocks after the additional seams, by rounding."""
    band = orc.Band(4096, 0.75, 480., 1920., 48000, "raised_cosine", 120., 480.)
    x = orc.synthetic_stereo(70000, 6)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    for f, split in ((8, 2), (16, 2), (16, 4), (24, 2)):
        base = run_zoom(emu, band, x, f)
        got = run_zoom(emu, band, x, f, c_split=split)
        assert np.array_equal(base[1], got[1]) and np.array_equal(base[2], got[2])
        assert not np.isnan(got[0]).any()
        assert rms(got[0].astype(np.float64) - ref[0]) < 1e-7
        fc = f // split
        differ = np.nonzero(base[0] != got[0])[0]
        seam_blocks = set()
        for m in range(-1 + fc, 80, fc):
            seam_blocks.update(range(m, m + 3))
        assert all((int(i) // 1024) in seam_blocks for i in differ), (f, split)
    # accumulate onto existing planes, merged gain list, two residue groups (N = 8192, P = 256)
    band = orc.Band(8192, 0.75, 120., 480., 48000, "raised_cosine", 30., 120.)
    x = orc.synthetic_stereo(90000, 7)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    outs = [np.full(len(x), 0.25, np.float32) for _ in range(3)]
    got = run_zoom(emu, band, x, 8, outs=outs, accumulate=1, c_split=2)
    for g, r in zip(got, ref):
        assert rms(g.astype(np.float64) - 0.25 - r) < 1e-7
    # a length the split does not divide is refused by the emulator's driver
    with pytest.raises(AssertionError):
        run_zoom(emu, band, x, 6, c_split=2)


def deal(first_pair, n_pairs, n):
    """Stream starts the library deals: n streams over n_pairs frame pairs, lengths within one pair of each other."""
    return [-1 + 2 * (first_pair + n_pairs * i // n) for i in range(n)] + [-1 + 2 * (first_pair + n_pairs)]


def test_stream_tables_one_workgroup_per_slot(emu):
    """ZoomArgs::stream_m0 / stream_m0_c: the geometry upx_process_device uses - streams of unequal length, the centre
    streams about twice as long as the Ls/Rs ones; equals the oracle, and equals the uniform cut away from the seams."""
    band = orc.Band(4096, 0.75, 480., 1920., 48000, "raised_cosine", 120., 480.)
    x = orc.synthetic_stereo(100000, 9)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    frames = -(-len(x) // 1024) + 1
    pairs = (frames + 1) // 2
    for n_lr, n_c in ((7, 4), (9, 3), (1, 1), (5, 5)):
        tabs = (deal(0, pairs, n_lr), deal(0, pairs, n_c))
        got = run_zoom(emu, band, x, 1000, tables=tabs)
        for g, r in zip(got, ref):
            assert not np.isnan(g).any()
            assert rms(g.astype(np.float64) - r) < 1e-7, (n_lr, n_c)
    base = run_zoom(emu, band, x, 1000)
    tabs = (deal(0, pairs, 6), deal(0, pairs, 3))
    got = run_zoom(emu, band, x, 1000, tables=tabs)
    for plane, tab in ((0, tabs[1]), (1, tabs[0]), (2, tabs[0])):
        differ = np.nonzero(base[plane] != got[plane])[0]
        seam_blocks = set()
        for m in tab[1:-1]:
            seam_blocks.update(range(m, m + 3))
        assert len(differ) > 0 and all((int(i) // 1024) in seam_blocks for i in differ), plane
    # merged gain list, two residue groups, accumulate (N = 8192, P = 256)
    band = orc.Band(8192, 0.75, 120., 480., 48000, "raised_cosine", 30., 120.)
    x = orc.synthetic_stereo(120000, 7)
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
    frames = -(-len(x) // 2048) + 1
    pairs = (frames + 1) // 2
    outs = [np.full(len(x), 0.25, np.float32) for _ in range(3)]
    got = run_zoom(emu, band, x, 1000, outs=outs, accumulate=1, tables=(deal(0, pairs, 5), deal(0, pairs, 2)))
    for g, r in zip(got, ref):
        assert rms(g.astype(np.float64) - 0.25 - r) < 1e-7
    # a table that does not cover the frames, or holds an odd length, is refused by the emulator's driver
    with pytest.raises(AssertionError):
        run_zoom(emu, band, x, 1000, tables=([-1, 8, 2 * pairs - 1], [-1, 2 * pairs - 1]))


def test_analysis_pairs_dealt_by_age_change_nothing(emu):
    """ZoomArgs::deal_rows: the XCD's workgroups take unequal numbers of pairs (rows of n_l, then runs of consecutive
    pairs).  Every pair is still transformed exactly once, so the planes do not change by a bit; a dealing that leaves
    pairs out shows up as NaN (the emulator poisons the spectra)."""
    band = orc.Band(4096, 0.75, 480., 1920., 48000, "raised_cosine", 120., 480.)
    x = orc.synthetic_stereo(200000, 5)     # one stream of 200 frames -> 100 pairs -> 13 per XCD share
    base = run_zoom(emu, band, x, 200, pairs_per_wg=3)

    def deal(rows, extras):
        tab, behind = [], 0
        for n in extras:
            tab += [behind, n]
            behind += n
        emu.emu_zoom_set_deal(rows, (ctypes.c_int * len(tab))(*tab), len(extras))

    try:
        for rows, extras in ((3, [3, 1, 0]), (2, [3, 2, 2]), (1, [5, 3, 2]), (4, [1, 0, 0]), (1, [10, 0, 0])):
            assert rows * 3 + sum(extras) >= 13
            deal(rows, extras)
            got = run_zoom(emu, band, x, 200, pairs_per_wg=3)
            for a, b in zip(base, got):
                assert np.array_equal(a, b), (rows, extras)
        deal(3, [1, 0, 0])   # too few pairs dealt: spectra left unwritten
        got = run_zoom(emu, band, x, 200, pairs_per_wg=3)
        assert any(np.isnan(g).any() for g in got)
    finally:
        emu.emu_zoom_set_deal(0, (ctypes.c_int * 2)(), 0)


def test_band_accumulation_and_merged_gain_list(emu):
    """accumulate=1 adds onto the planes of the previous band; bands that share N / hop / windows run as one launch
    with a per-bin gain list and equal the sum of the separate bands."""
    bands = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000, max_block_size=4096)[:4]
    assert [b.block_size for b in bands] == [4096] * 4
    x = orc.synthetic_stereo(30000, 8)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), bands)
    outs = None
    for i, b in enumerate(bands):
        # one decimated length for the whole group, as a merged launch would use
        outs = run_zoom(emu, b, x, 6, outs=outs, accumulate=1 if i else 0, log2p=9)
    for g, r in zip(outs, ref):
        assert rms(g.astype(np.float64) - r) < 1e-7
    g = np.stack([0.5 * orc.band_gain(b) for b in bands]).astype(np.float32)
    slots = int((g != 0).sum(axis=0).max())
    table = np.zeros((slots, g.shape[1]), np.float32)
    for k in range(g.shape[1]):
        nz = g[:, k][g[:, k] != 0]
        table[:len(nz), k] = nz
    assert table.shape[0] == 2
    for got, r in zip(run_zoom(emu, bands[0], x, 6, gain_table=table), ref):
        assert rms(got.astype(np.float64) - r) < 1e-7


def test_short_ragged_and_shard_arguments(emu):
    band = orc.Band(4096, 0.75, 480., 1920., 48000, "raised_cosine", 120., 480.)
    for total in (1, 1023, 1024, 1025, 4096, 4097, 9000):
        x = orc.synthetic_stereo(total, total)
        ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band)
        for g, r in zip(run_zoom(emu, band, x, 4), ref):
            assert g.shape == (total,) and rms(g.astype(np.float64) - r) < 1e-7
    # own_len / t_out: only frames starting in the owned range, output spills N - hop past it
    x = orc.synthetic_stereo(32768 + 3072, 7)
    own, t_out = 32768, 32768 + 3072
    ref = orc.band_process(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), band, own_len=own, out_len=t_out)
    for g, r in zip(run_zoom(emu, band, x, 4, own_len=own, t_out=t_out), ref):
        assert rms(g.astype(np.float64) - r) < 1e-7
    assert rms(ref[0][own:]) > 0


def test_silence_is_exact_zero(emu):
    band = orc.Band(8192, 0.75, 120., 480., 48000, "raised_cosine", 30., 120.)
    got = run_zoom(emu, band, np.zeros((30000, 2), np.float32), 4)
    assert all(not g.any() for g in got)
