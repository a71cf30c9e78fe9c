#!/usr/bin/env python3
"""
Generate the golden fixtures under tests/golden/ by IMPORTING the unmodified
reference (willleskowitz/upmix, /root/reference/python-prototype) in the build
container.  The reference has no tests or fixtures of its own (SURVEY.md
section 4), so every pin is produced here.  Only data (inputs + expected
outputs) is written; no reference source travels.

Run:  python tests/golden/make_golden.py           (needs /root/reference)

`soundfile` is not installed in the image and is only used by the reference's
two main() functions; a stub module stands in for it.  For fixture F7 the stub
implements read()/write() in memory so that main.py:main() itself can be run
for each export mode (the mode literal inside main() is swapped through
``code.replace(co_consts=...)``; the function body is executed unmodified).
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference/python-prototype"
HERE = os.path.dirname(os.path.abspath(__file__))

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")
sf_stub = types.ModuleType("soundfile")
sys.modules["soundfile"] = sf_stub
sys.path.insert(0, REF)
import center_extraction as ce  # noqa: E402  (the reference)
import main as ref_main  # noqa: E402


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out, buf.getvalue()


def synth(total, seed):
    rng = np.random.default_rng(seed)
    m = rng.standard_normal(total)
    s = rng.standard_normal(total)
    x = np.empty((total, 2), dtype=np.float32)
    x[:, 0] = 0.1 * (m + 0.5 * s)
    x[:, 1] = 0.1 * (m - 0.5 * s)
    return x


WINDOWS = {
    "blackman_harris": ce.make_blackman_harris,
    "sqrt_hann": ce.make_sqrt_hann,
    "hann": ce.make_hann,
    "blackman": ce.make_blackman,
    "hamming": ce.make_hamming,
    "rect": ce.make_rect,
}


def build_chain(edges, overlap, window, sr, mode, max_block, tf):
    """chain_bands' wiring with the two block-size knobs exposed, built from the reference's own pieces."""
    edges = list(edges)
    if edges[-1] < sr / 2.0:
        edges = edges + [sr / 2.0]
    out, prev = [], 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        n = ce.compute_block_size_for_low_freq(lo, sr, max_block_size=max_block, threshold_factor=tf)
        w_hi = ce.hp_freq_to_crossover_width(hi)
        out.append(ce.MultiBandExtractorAccu(n, overlap, window, lo, hi, sr, mode, prev, w_hi))
        prev = w_hi
    return out


def gain_of(bex):
    """Run the reference's _band_limit on all-ones spectra to read the per-bin gain."""
    nb = bex.block_size // 2 + 1
    a = np.ones(nb, dtype=np.complex128)
    b = np.ones(nb, dtype=np.complex128)
    bex._band_limit(a, b)
    assert np.array_equal(a, b) and np.all(a.imag == 0)
    return a.real.copy()


def f0_plan():
    rec = {"block_size": [], "freq_to_bin": [], "chain": []}
    for f_low in (0, 20, 30, 120, 300, 480, 1920, 3000, 7680, 20000):
        for sr in (44100, 48000, 96000):
            for mx in (4096, 8192, 65536):
                for tf in (32, 64):
                    rec["block_size"].append(
                        [f_low, sr, mx, tf, int(ce.compute_block_size_for_low_freq(f_low, sr, mx, tf))])
    for f, sr, n in ((58.59375, 48000, 2048), (82.03125, 48000, 2048), (35.15625, 48000, 2048),
                     (30, 48000, 8192), (120, 48000, 8192), (480, 48000, 8192), (1920, 48000, 4096),
                     (7680, 48000, 1024), (7680, 48000, 256), (24000, 48000, 256), (37.5, 48000, 8192),
                     (7.5, 48000, 8192), (300, 48000, 4096), (3000, 48000, 1024), (750, 48000, 4096),
                     (22050, 44100, 512), (1000, 44100, 1024), (11.71875 * 2.5, 48000, 4096)):
        rec["freq_to_bin"].append([f, sr, n, int(ce.freq_to_bin(f, sr, n))])
    for x in (0, 1, 2, 3, 5, 1023, 1024, 1025, 65535, 65536):
        rec.setdefault("next_pow2", []).append([x, int(ce.next_power_of_2(x))])
    for edges, sr in (([0, 30, 120, 480, 1920, 7680], 48000), ([0, 30, 120, 480, 1920, 7680], 96000),
                      ([0, 300, 3000], 48000), ([0, 40, 200, 2000], 44100), ([0, 24000], 48000)):
        ext, log = quiet(ce.chain_bands, edges, 0.75, ce.make_rect, sr)
        rec["chain"].append({
            "edges": edges, "sr": sr, "log": log,
            "bands": [[b.block_size, b.hop_size, b.f_low, b.f_high, b.xover_width_low_hz,
                       b.xover_width_high_hz, b.xover_mode] for b in ext]})
    with open(os.path.join(HERE, "f0_plan.json"), "w") as fh:
        json.dump(rec, fh, indent=0)


def f1_windows():
    out = {}
    for name, fn in WINDOWS.items():
        for n in (256, 2048):
            wa = fn(n)
            out[f"{name}_{n}_wa"] = wa
            for ov in (0.5, 0.75):
                out[f"{name}_{n}_ws_{ov}"] = ce.design_wola_synthesis_window(wa, ov)
    wa = ce.make_blackman_harris(8192)
    out["blackman_harris_8192_wa"] = wa
    out["blackman_harris_8192_ws_0.75"] = ce.design_wola_synthesis_window(wa, 0.75)
    out["blackman_harris_2048_ws_0.875"] = ce.design_wola_synthesis_window(ce.make_blackman_harris(2048), 0.875)
    out["hann_2048_ws_0.6"] = ce.design_wola_synthesis_window(ce.make_hann(2048), 0.6)
    np.savez_compressed(os.path.join(HERE, "f1_windows.npz"), **out)


PLANS = {
    # name: (edges, sr, max_block, threshold_factor, mode)
    "c3_6band_8192_48k": ([0, 30, 120, 480, 1920, 7680], 48000, 8192, 32, "raised_cosine"),
    "c4_6band_8192_96k": ([0, 30, 120, 480, 1920, 7680], 96000, 8192, 32, "raised_cosine"),
    "c2_3band_4096_48k": ([0, 300, 3000], 48000, 4096, 64, "raised_cosine"),
    "hard_300_3000_1024": ([300, 3000], 48000, 1024, 32, "hard_zero"),
    "unknown_mode_1024": ([300, 3000], 48000, 1024, 32, "no_such_mode"),
}


def f2_gains():
    out = {}
    for name, (edges, sr, mx, tf, mode) in PLANS.items():
        # the [300,3000] plans get an explicit upper edge = 3000 (no Nyquist append wanted)
        if name.startswith(("hard", "unknown")):
            ext = [ce.MultiBandExtractorAccu(1024, 0.75, ce.make_rect, 300.0, 3000.0, sr, mode, 75.0, 750.0)]
        else:
            ext = build_chain(edges, 0.75, ce.make_rect, sr, mode, mx, tf)
        for i, b in enumerate(ext):
            out[f"{name}_b{i}_N{b.block_size}"] = gain_of(b)
    np.savez_compressed(os.path.join(HERE, "f2_gains.npz"), **out)


def f3_frames():
    out = {}
    for n, lo, hi, wl, wh, seed in ((256, 7680.0, 24000.0, 480.0, 6000.0, 30),
                                    (2048, 0.0, 24000.0, 0.0, 6000.0, 31),
                                    (8192, 120.0, 480.0, 30.0, 120.0, 32)):
        bex = ce.MultiBandExtractorAccu(n, 0.75, ce.make_blackman_harris, lo, hi, 48000, "raised_cosine", wl, wh)
        x = synth(n, seed)
        bl, br = x[:, 0].astype(np.float64), x[:, 1].astype(np.float64)
        sl = ce.forward_stft(bl, bex.analysis_window)
        sr_ = ce.forward_stft(br, bex.analysis_window)
        bex._band_limit(sl, sr_)
        # the mask block, through the reference's own process_stereo_chunk on a fresh extractor:
        c, l, r = bex.process_stereo_chunk(bl, br)
        fl = bex.flush_final()
        rec_c = np.concatenate([c, fl[0]])[:n]
        rec_l = np.concatenate([l, fl[1]])[:n]
        rec_r = np.concatenate([r, fl[2]])[:n]
        out[f"N{n}_x"] = x
        out[f"N{n}_params"] = np.array([lo, hi, wl, wh])
        out[f"N{n}_specL_bl"] = sl.astype(np.complex64)
        out[f"N{n}_specR_bl"] = sr_.astype(np.complex64)
        # first-frame accumulators = rec (accum starts at zero): hop emitted + flushed rest, shifted
        out[f"N{n}_rec_c"] = rec_c
        out[f"N{n}_rec_l"] = rec_l
        out[f"N{n}_rec_r"] = rec_r
    np.savez_compressed(os.path.join(HERE, "f3_frames.npz"), **out)


def run_band(bex, x):
    c, l, r = bex.process_all_blocks(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64))
    return c, l, r


def f4_oneband():
    out = {}
    for tag, total, seed in (("T12345", 12345, 40), ("T1000", 1000, 41), ("T512", 512, 42), ("T2048", 2048, 43),
                             ("T1", 1, 44)):
        bex = ce.MultiBandExtractorAccu(2048, 0.75, ce.make_blackman_harris, 0.0, 24000.0, 48000,
                                        "raised_cosine", 0.0, 6000.0)
        x = synth(total, seed)
        c, l, r = run_band(bex, x)
        out[f"{tag}_x"] = x
        out[f"{tag}_c"], out[f"{tag}_l"], out[f"{tag}_r"] = c, l, r
    # other overlaps / windows on one band
    for tag, ov, wname, n in (("ov50_sqrt_hann", 0.5, "sqrt_hann", 1024), ("ov875_hann", 0.875, "hann", 1024),
                              ("ov60_hamming", 0.6, "hamming", 512)):
        bex = ce.MultiBandExtractorAccu(n, ov, WINDOWS[wname], 200.0, 8000.0, 44100, "raised_cosine", 50.0, 2000.0)
        x = synth(6000, 45)
        c, l, r = run_band(bex, x)
        out[f"{tag}_x"] = x
        out[f"{tag}_c"], out[f"{tag}_l"], out[f"{tag}_r"] = c, l, r
    np.savez_compressed(os.path.join(HERE, "f4_oneband.npz"), **out)


def f5_multiband():
    out = {}
    total = 32768
    for name in ("c3_6band_8192_48k", "c2_3band_4096_48k", "c4_6band_8192_96k"):
        edges, sr, mx, tf, mode = PLANS[name]
        ext = build_chain(edges, 0.75, ce.make_blackman_harris, sr, mode, mx, tf)
        seed = {"c3_6band_8192_48k": 2, "c2_3band_4096_48k": 1, "c4_6band_8192_96k": 3}[name]
        t = total if "c4" not in name else 20000
        x = synth(t, seed)
        (c, l, r), _ = quiet(ce.extract_center_left_right_multi_band_in_memory,
                             x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), sr, ext)
        out[f"{name}_x"] = x
        out[f"{name}_c"], out[f"{name}_l"], out[f"{name}_r"] = c, l, r
    # the reference's own default plan through its own chain_bands (max STFT 65536)
    ext, _ = quiet(ce.chain_bands, [0, 30, 120, 480, 1920, 7680], 0.75, ce.make_blackman_harris, 48000)
    x = synth(20000, 5)
    c, l, r = ce.extract_center_left_right_multi_band_in_memory(
        x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), 48000, ext)
    out["default_65536_x"] = x
    out["default_65536_c"], out["default_65536_l"], out["default_65536_r"] = c, l, r
    np.savez_compressed(os.path.join(HERE, "f5_multiband.npz"), **out)


def f6_degenerate():
    out = {}
    total = 6000
    base = synth(total, 60)
    cases = {
        "silence": np.zeros((total, 2), np.float32),
        "l_eq_r": np.stack([base[:, 0], base[:, 0]], axis=1),
        "r_zero": np.stack([base[:, 0], np.zeros(total, np.float32)], axis=1),
        "l_zero": np.stack([np.zeros(total, np.float32), base[:, 1]], axis=1),
        "tiny": (base * np.float32(1e-7)),
        "antiphase": np.stack([base[:, 0], -base[:, 0]], axis=1),
    }
    for tag, x in cases.items():
        ext = build_chain([0, 300, 3000], 0.75, ce.make_blackman_harris, 48000, "raised_cosine", 1024, 32)
        (c, l, r), _ = quiet(ce.extract_center_left_right_multi_band_in_memory,
                             x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), 48000, ext)
        out[f"{tag}_x"] = x.astype(np.float32)
        out[f"{tag}_c"], out[f"{tag}_l"], out[f"{tag}_r"] = c, l, r
    np.savez_compressed(os.path.join(HERE, "f6_degenerate.npz"), **out)


EDGE_BANDS = {   # tag: (N, f_low, f_high, mode, width_low, width_high)   sr = 48000
    "swapped_rc": (1024, 3000.0, 300.0, "raised_cosine", 75.0, 750.0),       # f_low > f_high: bins swap (:342-343, :290-291)
    "swapped_hz": (1024, 3000.0, 300.0, "hard_zero", 50.0, 50.0),
    "above_nyquist_rc": (1024, 1920.0, 30000.0, "raised_cosine", 480.0, 7500.0),   # bin_high clamps (:293), no fade-out
    "above_nyquist_hz": (1024, 1920.0, 30000.0, "hard_zero", 50.0, 50.0),
    "both_above_rc": (512, 30000.0, 40000.0, "raised_cosine", 100.0, 100.0),      # bin_low > clamped bin_high: silence
    "fade_wider_than_band": (2048, 100.0, 200.0, "raised_cosine", 5000.0, 40000.0),  # fades clipped at 0 and n_bins
    "zero_width": (1024, 300.0, 3000.0, "raised_cosine", 0.0, 0.0),
}


def f8_edges():
    """Band-edge corner cases of _band_limit and non-finite / out-of-range samples (edge cases)."""
    out, meta = {}, {}
    x = synth(5000, 80)
    out["x"] = x
    for tag, (n, lo, hi, mode, wl, wh) in EDGE_BANDS.items():
        bex = ce.MultiBandExtractorAccu(n, 0.75, ce.make_blackman_harris, lo, hi, 48000, mode, wl, wh)
        out[f"{tag}_gain"] = gain_of(bex)
        (c, l, r) = bex.process_all_blocks(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64))
        out[f"{tag}_c"], out[f"{tag}_l"], out[f"{tag}_r"] = c, l, r
    # non-finite samples: one NaN in L, one +Inf in R, far apart; 2-band plan [1024, 512]
    y = synth(9000, 81).astype(np.float64)
    y[2000, 0] = np.nan
    y[6500, 1] = np.inf
    ext = build_chain([0, 3000], 0.75, ce.make_blackman_harris, 48000, "raised_cosine", 1024, 32)
    with np.errstate(all="ignore"):
        (c, l, r), _ = quiet(ce.extract_center_left_right_multi_band_in_memory, y[:, 0], y[:, 1], 48000, ext)
    out["nonfinite_x"] = y
    out["nonfinite_c"], out["nonfinite_l"], out["nonfinite_r"] = c, l, r
    # float64 samples beyond the float32 range (the reference computes in float64 and casts its result to float32)
    z = synth(9000, 82).astype(np.float64)
    z[4000:4003, 0] = [1e39, -2e39, 5e38]
    ext = build_chain([0, 3000], 0.75, ce.make_blackman_harris, 48000, "raised_cosine", 1024, 32)
    with np.errstate(all="ignore"):
        (c, l, r), _ = quiet(ce.extract_center_left_right_multi_band_in_memory, z[:, 0], z[:, 1], 48000, ext)
    out["huge_x"] = z
    out["huge_c"], out["huge_l"], out["huge_r"] = c, l, r
    meta["sizes"] = [int(b.block_size) for b in ext]
    meta["numpy"] = np.__version__
    np.savez_compressed(os.path.join(HERE, "f8_edges.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "f8_edges.json"), "w"), indent=1)


def f7_main():
    """Run main.py:main() itself for each export mode with an in-memory soundfile stub."""
    out = {}
    meta = {}
    total = 9000
    x = synth(total, 70)
    x[100, 0] = 0.9  # a defined input peak
    mono = x[:, 0].copy()

    def run(mode, wave, sr):
        written = {}
        sf_stub.read = lambda path: (wave.astype(np.float64), sr)
        sf_stub.write = lambda path, data, rate: written.__setitem__(path, (np.array(data), rate))
        code = ref_main.main.__code__
        # identical literals share one co_consts slot, so the assignment and the
        # `== "stereo_sum"` test are swapped together; an unknown mode therefore
        # needs a value that compares unequal even to itself.
        class _NoMatch(str):
            __hash__ = str.__hash__

            def __eq__(self, other):
                return False
        val = _NoMatch(mode) if mode == "bogus" else mode
        consts = tuple(val if (isinstance(c, str) and c == "stereo_sum") else c for c in code.co_consts)
        fn = types.FunctionType(code.replace(co_consts=consts), ref_main.main.__globals__)
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, "in"))
            open(os.path.join(tmp, "in", "eyes.wav"), "wb").close()
            os.chdir(tmp)
            try:
                _, log = quiet(fn)
            finally:
                os.chdir(cwd)
        return written, log

    for mode in ("AB", "split", "stereo_sum", "bogus"):
        written, log = run(mode, x, 48000)
        meta[mode] = {"files": sorted(written), "log_tail": [ln for ln in log.splitlines()
                                                             if not ln.startswith("[Band")]}
        for path, (data, rate) in written.items():
            out[f"{mode}:{path}"] = data
            assert rate == 48000
    written, log = run("stereo_sum", mono, 44100)
    meta["mono_stereo_sum"] = {"files": sorted(written)}
    for path, (data, rate) in written.items():
        out[f"mono_stereo_sum:{path}"] = data
    out["x"] = x
    out["mono"] = mono
    # `final_x *= scale_factor` (main.py:95-97) multiplies in float64 under NumPy >= 2 and in float32 under NumPy 1.x
    # (NEP 50): the fixture pins the behaviour of the NumPy that generated it
    meta["numpy"] = np.__version__
    np.savez_compressed(os.path.join(HERE, "f7_main.npz"), **out)
    with open(os.path.join(HERE, "f7_main.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


if __name__ == "__main__":
    only = sys.argv[1:]   # e.g. `make_golden.py f8_edges`: regenerate just that fixture
    for fn in (f0_plan, f1_windows, f2_gains, f3_frames, f4_oneband, f5_multiband, f6_degenerate, f7_main, f8_edges):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("wrote", fn.__name__)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
