"""
Pins oracle/upmix_oracle.py against the golden fixtures generated from the
unmodified reference (tests/golden/make_golden.py).  Bit-exact: max|diff| == 0.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from oracle import upmix_oracle as orc

WIN = orc.WINDOWS


def chain(edges, sr, max_block, tf, mode="raised_cosine", window=orc.win_blackman_harris, overlap=0.75):
    return orc.plan_bands(edges, overlap, window, sr, mode, max_block_size=max_block, threshold_factor=tf)


def test_f0_plan_tables():
    rec = json.load(open(os.path.join(GOLDEN, "f0_plan.json")))
    for f_low, sr, mx, tf, want in rec["block_size"]:
        assert orc.block_size_for_low_freq(f_low, sr, mx, tf) == want
    for f, sr, n, want in rec["freq_to_bin"]:
        assert orc.freq_to_bin(f, sr, n) == want
    for x, want in rec["next_pow2"]:
        assert orc.next_pow2(x) == want
    for ch in rec["chain"]:
        bands = orc.plan_bands(ch["edges"], 0.75, orc.win_rect, ch["sr"])
        got = [[b.block_size, b.hop_size, b.f_low, b.f_high, b.xover_width_low_hz, b.xover_width_high_hz,
                b.xover_mode] for b in bands]
        assert got == ch["bands"]


def test_f1_windows_bit_exact():
    z = load_golden("f1_windows.npz")
    for key in z.files:
        parts = key.split("_")
        if parts[-1] == "wa":
            n = int(parts[-2]); name = "_".join(parts[:-2])
            got = WIN[name](n)
        else:
            ov = float(parts[-1]); n = int(parts[-3]); name = "_".join(parts[:-3])
            got = orc.wola_synthesis_window(WIN[name](n), ov)
        assert got.dtype == np.float32
        assert np.array_equal(got, z[key]), key


def test_wola_hop_too_small_raises():
    with pytest.raises(ValueError):
        orc.wola_synthesis_window(orc.win_hann(4), 0.9)
    with pytest.raises(ValueError):
        orc.Band(4, 0.9, 0.0, 100.0, 48000)


def test_f2_gain_vectors_bit_exact():
    z = load_golden("f2_gains.npz")
    plans = {
        "c3_6band_8192_48k": chain([0, 30, 120, 480, 1920, 7680], 48000, 8192, 32),
        "c4_6band_8192_96k": chain([0, 30, 120, 480, 1920, 7680], 96000, 8192, 32),
        "c2_3band_4096_48k": chain([0, 300, 3000], 48000, 4096, 64),
        "hard_300_3000_1024": [orc.Band(1024, 0.75, 300.0, 3000.0, 48000, "hard_zero", 75.0, 750.0)],
        "unknown_mode_1024": [orc.Band(1024, 0.75, 300.0, 3000.0, 48000, "no_such_mode", 75.0, 750.0)],
    }
    seen = 0
    for name, bands in plans.items():
        for i, b in enumerate(bands):
            key = f"{name}_b{i}_N{b.block_size}"
            assert np.array_equal(orc.band_gain(b), z[key]), key
            seen += 1
    assert seen == len(z.files)
    # SURVEY 3.5 probes
    g = orc.band_gain(plans["c3_6band_8192_48k"][2])
    nz = np.nonzero(g)[0]
    assert (nz[0], nz[-1]) == (15, 102)
    g = orc.band_gain(plans["hard_300_3000_1024"][0])
    nz = np.nonzero(g)[0]
    assert (nz[0], nz[-1]) == (6, 64)


def test_f3_single_frame():
    z = load_golden("f3_frames.npz")
    for n in (256, 2048, 8192):
        lo, hi, wl, wh = z[f"N{n}_params"]
        band = orc.Band(n, 0.75, lo, hi, 48000, "raised_cosine", wl, wh)
        x = z[f"N{n}_x"].astype(np.float64)
        g = orc.band_gain(band)
        rc, rl, rr = orc.frames_to_recs(x[:, 0], x[:, 1], band, g)
        assert np.array_equal(rc, z[f"N{n}_rec_c"])
        assert np.array_equal(rl, z[f"N{n}_rec_l"])
        assert np.array_equal(rr, z[f"N{n}_rec_r"])
        sl = np.fft.rfft(x[:, 0] * band.analysis_window) * g
        assert np.array_equal(sl.astype(np.complex64), z[f"N{n}_specL_bl"])


@pytest.mark.parametrize("proc", [orc.band_process, orc.band_process_streaming])
def test_f4_one_band(proc):
    z = load_golden("f4_oneband.npz")
    for tag in ("T12345", "T1000", "T512", "T2048", "T1"):
        band = orc.Band(2048, 0.75, 0.0, 24000.0, 48000, "raised_cosine", 0.0, 6000.0)
        x = z[f"{tag}_x"].astype(np.float64)
        c, l, r = proc(x[:, 0], x[:, 1], band)
        for got, k in ((c, "c"), (l, "l"), (r, "r")):
            assert got.dtype == np.float32 and got.shape == (len(x),)
            assert np.array_equal(got, z[f"{tag}_{k}"]), (tag, k)
    for tag, ov, wname, n in (("ov50_sqrt_hann", 0.5, "sqrt_hann", 1024), ("ov875_hann", 0.875, "hann", 1024),
                              ("ov60_hamming", 0.6, "hamming", 512)):
        band = orc.Band(n, ov, 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0, window=WIN[wname])
        x = z[f"{tag}_x"].astype(np.float64)
        c, l, r = proc(x[:, 0], x[:, 1], band)
        for got, k in ((c, "c"), (l, "l"), (r, "r")):
            assert np.array_equal(got, z[f"{tag}_{k}"]), (tag, k)


def test_f5_multi_band():
    z = load_golden("f5_multiband.npz")
    plans = {
        "c3_6band_8192_48k": chain([0, 30, 120, 480, 1920, 7680], 48000, 8192, 32),
        "c2_3band_4096_48k": chain([0, 300, 3000], 48000, 4096, 64),
        "c4_6band_8192_96k": chain([0, 30, 120, 480, 1920, 7680], 96000, 8192, 32),
        "default_65536": orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000),
    }
    assert [b.block_size for b in plans["c3_6band_8192_48k"]] == [8192, 8192, 8192, 4096, 1024, 256]
    assert [b.block_size for b in plans["c2_3band_4096_48k"]] == [4096, 4096, 1024]
    assert [b.block_size for b in plans["c4_6band_8192_96k"]] == [8192, 8192, 8192, 8192, 2048, 512]
    assert [b.block_size for b in plans["default_65536"]] == [65536, 65536, 16384, 4096, 1024, 256]
    for name, bands in plans.items():
        x = z[f"{name}_x"].astype(np.float64)
        c, l, r = orc.extract_multi_band(x[:, 0], x[:, 1], bands)
        for got, k in ((c, "c"), (l, "l"), (r, "r")):
            assert np.array_equal(got, z[f"{name}_{k}"]), (name, k)


def test_threadpool_shape_matches_serial():
    z = load_golden("f5_multiband.npz")
    bands = chain([0, 300, 3000], 48000, 4096, 64)
    x = z["c2_3band_4096_48k_x"][:9000].astype(np.float64)
    a = orc.extract_multi_band_threadpool(x[:, 0], x[:, 1], bands)
    b = orc.extract_multi_band(x[:, 0], x[:, 1], bands)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_f6_degenerate_inputs():
    z = load_golden("f6_degenerate.npz")
    for tag in ("silence", "l_eq_r", "r_zero", "l_zero", "tiny", "antiphase"):
        bands = chain([0, 300, 3000], 48000, 1024, 32)
        x = z[f"{tag}_x"].astype(np.float64)
        c, l, r = orc.extract_multi_band(x[:, 0], x[:, 1], bands)
        for got, k in ((c, "c"), (l, "l"), (r, "r")):
            assert np.array_equal(got, z[f"{tag}_{k}"]), (tag, k)
            assert np.all(np.isfinite(got))
    assert not z["silence_c"].any() and not z["silence_l"].any()
    assert not z["r_zero_c"].any()


def test_f7_main_postprocessing():
    z = load_golden("f7_main.npz")
    meta = json.load(open(os.path.join(GOLDEN, "f7_main.json")))
    x = z["x"].astype(np.float64)
    bands = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 48000)
    for mode in ("AB", "split", "stereo_sum", "bogus"):
        sig_l, sig_r = x[:, 0], x[:, 1]
        c, l, r = orc.extract_multi_band(sig_l, sig_r, bands)
        orc.normalise_lcr(c, l, r, orc.input_peak(x))
        lay = orc.export_layout(mode, c, l, r, sig_l, sig_r)
        names = orc.output_names("eyes", mode, bands, 0.75)
        assert sorted(os.path.join("out", v) for v in names.values()) == meta[mode]["files"]
        for k, arr in lay.items():
            want = z[f"{mode}:{os.path.join('out', names[k])}"]
            assert arr.dtype == want.dtype
            assert np.array_equal(arr, want), (mode, k)
    # mono input is duplicated to both channels (main.py:47-48)
    mono = z["mono"].astype(np.float64)
    wave = np.column_stack([mono, mono])
    bands44 = orc.plan_bands([0, 30, 120, 480, 1920, 7680], 0.75, orc.win_blackman_harris, 44100)
    c, l, r = orc.extract_multi_band(wave[:, 0], wave[:, 1], bands44)
    orc.normalise_lcr(c, l, r, orc.input_peak(wave))
    lay = orc.export_layout("stereo_sum", c, l, r)
    names = orc.output_names("eyes", "stereo_sum", bands44, 0.75)
    assert np.array_equal(lay["Sum"], z[f"mono_stereo_sum:{os.path.join('out', names['Sum'])}"])


EDGE_BANDS = {   # as tests/golden/make_golden.py: tag -> (N, f_low, f_high, mode, width_low, width_high), sr = 48000
    "swapped_rc": (1024, 3000.0, 300.0, "raised_cosine", 75.0, 750.0),
    "swapped_hz": (1024, 3000.0, 300.0, "hard_zero", 50.0, 50.0),
    "above_nyquist_rc": (1024, 1920.0, 30000.0, "raised_cosine", 480.0, 7500.0),
    "above_nyquist_hz": (1024, 1920.0, 30000.0, "hard_zero", 50.0, 50.0),
    "both_above_rc": (512, 30000.0, 40000.0, "raised_cosine", 100.0, 100.0),
    "fade_wider_than_band": (2048, 100.0, 200.0, "raised_cosine", 5000.0, 40000.0),
    "zero_width": (1024, 300.0, 3000.0, "raised_cosine", 0.0, 0.0),
}


def test_f8_band_edge_corner_cases():
    """Swapped edges (center_extraction.py:342-343, :290-291), edges above Nyquist (clamp :293, no fade-out :319),
    fades clipped at bin 0 / n_bins, zero fade widths: gain vectors and band outputs bit for bit."""
    z = load_golden("f8_edges.npz")
    x = z["x"].astype(np.float64)
    for tag, (n, lo, hi, mode, wl, wh) in EDGE_BANDS.items():
        band = orc.Band(n, 0.75, lo, hi, 48000, mode, wl, wh)
        assert np.array_equal(orc.band_gain(band), z[f"{tag}_gain"]), tag
        for got, k in zip(orc.band_process(x[:, 0], x[:, 1], band), "clr"):
            assert np.array_equal(got, z[f"{tag}_{k}"]), (tag, k)
    assert not z["both_above_rc_gain"].any() and not z["both_above_rc_c"].any()
    assert z["swapped_rc_gain"].any() and z["above_nyquist_rc_gain"][-1] == 1.0


def test_f8_nonfinite_and_out_of_range_samples():
    """A NaN / Inf sample poisons exactly the frames that contain it (all three outputs, every band); float64 samples
    beyond the float32 range stay finite inside the reference's float64 transforms and overflow only in its final
    float32 cast.  Same placement, same finite values."""
    z = load_golden("f8_edges.npz")
    bands = chain([0, 3000], 48000, 1024, 32)
    assert [b.block_size for b in bands] == json.load(open(os.path.join(GOLDEN, "f8_edges.json")))["sizes"]
    for tag in ("nonfinite", "huge"):
        x = z[f"{tag}_x"]
        with np.errstate(all="ignore"):
            out = orc.extract_multi_band(x[:, 0], x[:, 1], bands)
        for got, k in zip(out, "clr"):
            ref = z[f"{tag}_{k}"]
            assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref)), (tag, k)
            ok = np.isfinite(ref)
            assert np.array_equal(got[ok], ref[ok]), (tag, k)
    # the NaN at sample 2000 (band sizes 1024 / 512): every frame that covers it, nothing else
    bad = ~np.isfinite(z["nonfinite_c"])
    assert bad[2000] and bad[2000 - 700] and not bad[2000 - 1024] and not bad[2000 + 1024] and bad[6500]
    assert np.array_equal(bad, ~np.isfinite(z["nonfinite_l"])) and np.array_equal(bad, ~np.isfinite(z["nonfinite_r"]))
