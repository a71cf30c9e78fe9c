"""
CPU oracle for the multi-band STFT centre-extraction hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the *checker*; the shipped path (``upmix_amd``) never
falls back to it and fails loudly when the HIP library is missing.

It is a NumPy restatement (float64 FFT via ``numpy.fft``, float32 overlap-add)
of the reference algorithm in ``python-prototype/center_extraction.py`` and of
the caller-side arithmetic in ``python-prototype/main.py``.  Every function
cites the reference lines it follows.  Parity is PINNED: ``tests/golden/*.npz``
were produced by importing the unmodified reference in the build container
(``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks
this file against them bit-for-bit (max |diff| == 0).

Two formulations of the per-band loop are provided:

* ``band_process_streaming`` - frame-at-a-time with shifting accumulators, the
  same shape as the reference's ``process_all_blocks`` (used for the CPU
  baseline timing and to validate the closed form);
* ``band_process`` - batched closed form: for every frame ``j`` with
  ``j*hop < T``: ``out[j*hop : j*hop+N] += w_S * irfft(mask(g * rfft(w_A * x_j)))``
  added in increasing ``j`` in float32.  Bit-identical
  to the streaming form and ~10x faster, so tests finish in seconds.
"""

from __future__ import annotations

import math
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Callable, List, Sequence, Tuple

import numpy as np

EPS = 1e-12  # center_extraction.py:36


# --------------------------------------------------------------------------
# Windows (center_extraction.py:42-75)
# --------------------------------------------------------------------------
def win_blackman_harris(n: int) -> np.ndarray:
    """4-term symmetric Blackman-Harris, N-1 denominator (center_extraction.py:42-53)."""
    k = np.arange(n)
    c = (0.35875, 0.48829, 0.14128, 0.01168)
    d = n - 1
    w = c[0] - c[1] * np.cos(2 * np.pi * k / d) + c[2] * np.cos(4 * np.pi * k / d) \
        - c[3] * np.cos(6 * np.pi * k / d)
    return w.astype(np.float32)


def win_sqrt_hann(n: int) -> np.ndarray:
    """center_extraction.py:56-59"""
    return np.sqrt(np.hanning(n)).astype(np.float32)


def win_hann(n: int) -> np.ndarray:
    """center_extraction.py:61-63"""
    return np.hanning(n).astype(np.float32)


def win_blackman(n: int) -> np.ndarray:
    """center_extraction.py:65-67"""
    return np.blackman(n).astype(np.float32)


def win_hamming(n: int) -> np.ndarray:
    """center_extraction.py:69-71"""
    return np.hamming(n).astype(np.float32)


def win_rect(n: int) -> np.ndarray:
    """center_extraction.py:73-75"""
    return np.ones(n, dtype=np.float32)


WINDOWS = {
    "blackman_harris": win_blackman_harris,
    "sqrt_hann": win_sqrt_hann,
    "hann": win_hann,
    "blackman": win_blackman,
    "hamming": win_hamming,
    "rect": win_rect,
}


def hop_of(block_size: int, overlap: float) -> int:
    """hop = int(N * (1 - overlap)) (center_extraction.py:90 and :252)."""
    return int(block_size * (1.0 - overlap))


def wola_synthesis_window(w_a: np.ndarray, overlap: float) -> np.ndarray:
    """
    w_S[n] = w_A[n] / (sum_{k<K} w_A[(n + k*hop) mod L]^2 + EPS)   (center_extraction.py:80-105)

    The reference accumulates the sum in a Python loop starting from the float
    ``0.0`` and adding ``np.float32`` squares, which under NumPy >= 2 (NEP 50)
    is a float32 accumulation in k order; restated here as float32 vector ops
    (checked bit-exact against the reference by the golden fixture F1).
    """
    n = len(w_a)
    hop = int(n * (1.0 - overlap))
    if hop < 1:
        raise ValueError("Overlap too large; resulting hop size < 1.")
    k_frames = int(round(1.0 / (1.0 - overlap)))
    w = np.asarray(w_a)
    # NumPy *scalar* ``float32 ** 2`` goes through powf(), which is not always
    # equal to the correctly rounded x*x that the array power loop produces
    # (4 of 2048 Blackman-Harris taps differ by 1 ulp), so square per scalar.
    sq = np.array([v ** 2 for v in w], dtype=w.dtype)
    pos = np.arange(n)
    acc = np.zeros(n, dtype=w.dtype)
    for k in range(k_frames):
        acc = acc + sq[(pos + k * hop) % n]
    return (w / (acc + np.asarray(EPS, dtype=w.dtype))).astype(w.dtype)


# --------------------------------------------------------------------------
# Plan arithmetic (center_extraction.py:142-212)
# --------------------------------------------------------------------------
def freq_to_bin(freq_hz: float, sr: float, fft_size: int) -> int:
    """Python round() = half-to-even, no clamp (center_extraction.py:154)."""
    return int(round(freq_hz / (sr / float(fft_size))))


def next_pow2(x: int) -> int:
    """center_extraction.py:156-171"""
    p = 1
    while p < x:
        p <<= 1
    return p


def block_size_for_low_freq(f_low: float, sr: float, max_block_size: int = 2 ** 16,
                            threshold_factor: float = 32) -> int:
    """N = min(nextpow2(ceil(sr*tf/f_low)), max); max if f_low <= 0 (center_extraction.py:173-197)."""
    if f_low <= 0.0:
        return max_block_size
    need = (sr * threshold_factor) / f_low
    return min(next_pow2(int(np.ceil(need))), max_block_size)


def crossover_width(hp_freq: float, fraction: float = 0.25) -> float:
    """center_extraction.py:200-212 (fraction hard-wired to 0.25 there)."""
    return hp_freq * fraction


# --------------------------------------------------------------------------
# Band description + band limiter gain (center_extraction.py:273-351)
# --------------------------------------------------------------------------
@dataclass
class Band:
    """Parameters the reference keeps on MultiBandExtractorAccu (center_extraction.py:240-271)."""
    block_size: int
    overlap: float
    f_low: float
    f_high: float
    sr: float
    xover_mode: str = "hard_zero"
    xover_width_low_hz: float = 50.0
    xover_width_high_hz: float = 50.0
    window: Callable[[int], np.ndarray] = win_blackman_harris
    hop_size: int = field(init=False)
    analysis_window: np.ndarray = field(init=False, repr=False)
    synthesis_window: np.ndarray = field(init=False, repr=False)

    def __post_init__(self):
        self.hop_size = int(self.block_size * (1 - self.overlap))
        if self.hop_size < 1:
            raise ValueError("Overlap too large; hop size < 1 is not allowed.")
        self.analysis_window = self.window(self.block_size)
        self.synthesis_window = wola_synthesis_window(self.analysis_window, self.overlap)


def band_gain(band: Band) -> np.ndarray:
    """
    Real per-bin gain g[k] (float64) equivalent to ``_band_limit`` applied to a
    spectrum (center_extraction.py:334-351 dispatching to :273-280 / :282-332).
    """
    n_bins = band.block_size // 2 + 1
    fft_size = (n_bins - 1) * 2
    lo = freq_to_bin(band.f_low, band.sr, fft_size)
    hi = freq_to_bin(band.f_high, band.sr, fft_size)
    if lo > hi:
        lo, hi = hi, lo
    g = np.ones(n_bins, dtype=np.float64)

    if band.xover_mode != "raised_cosine":
        # hard zero, also the fallback for unknown modes (center_extraction.py:345-351)
        # NumPy slice semantics (negative indices wrap) are kept on purpose.
        g[:lo] = 0.0
        g[hi + 1:] = 0.0
        return g

    lo = max(lo, 0)
    hi = min(hi, n_bins - 1)
    if lo > hi:
        g[:] = 0.0
        return g
    fade_lo = freq_to_bin(band.xover_width_low_hz, band.sr, fft_size)
    fade_hi = freq_to_bin(band.xover_width_high_hz, band.sr, fft_size)

    if band.f_low > 0:                                   # :304-315
        start = max(0, lo - fade_lo)
        g[:start] = 0.0
        span = lo - start
        for i in range(span):
            g[start + i] *= 0.5 * (1.0 - np.cos(np.pi * ((i + 0.5) / span)))
    if band.f_high < band.sr * 0.5:                      # :318-332
        a = hi + 1
        if a < n_bins:
            b = min(a + fade_hi, n_bins)
            span = b - a
            for i in range(span):
                g[a + i] *= 0.5 * (1.0 + np.cos(np.pi * ((i + 0.5) / span)))
            if b < n_bins:
                g[b:] = 0.0
    return g


# --------------------------------------------------------------------------
# One frame: rfft -> band limit -> mask -> irfft (center_extraction.py:353-389)
# --------------------------------------------------------------------------
def frames_to_recs(blk_l: np.ndarray, blk_r: np.ndarray, band: Band, g: np.ndarray
                   ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """
    blk_*: [..., N] real blocks.  Returns float32 (rec_c, rec_l, rec_r), each
    [..., N], already multiplied by the synthesis window.
    forward_stft :110-122, mask :373-384, inverse_stft :124-137.
    """
    w_a = band.analysis_window
    w_s = band.synthesis_window
    spec_l = np.fft.rfft(blk_l * w_a, axis=-1) * g
    spec_r = np.fft.rfft(blk_r * w_a, axis=-1) * g

    cross_mag = np.abs(spec_l * np.conjugate(spec_r))
    mag_l = np.abs(spec_l)
    mag_r = np.abs(spec_r)
    coherence = cross_mag / ((mag_l * mag_r) + EPS)
    balance = (mag_l - mag_r) / (mag_l + mag_r + EPS)
    center_factor = coherence * (1.0 - np.abs(balance))

    spec_c = 0.5 * center_factor * (spec_l + spec_r)
    spec_ls = spec_l - spec_c
    spec_rs = spec_r - spec_c

    def back(spec):
        rec = np.fft.irfft(spec, axis=-1).astype(np.float32)
        rec *= w_s
        return rec

    return back(spec_c), back(spec_ls), back(spec_rs)


def frame_spectra(blk_l: np.ndarray, blk_r: np.ndarray, band: Band, g: np.ndarray):
    """Single-frame spectra (spec_center, spec_left, spec_right) for fixture F3."""
    w_a = band.analysis_window
    spec_l = np.fft.rfft(blk_l * w_a) * g
    spec_r = np.fft.rfft(blk_r * w_a) * g
    mag_l, mag_r = np.abs(spec_l), np.abs(spec_r)
    coh = np.abs(spec_l * np.conjugate(spec_r)) / ((mag_l * mag_r) + EPS)
    bal = (mag_l - mag_r) / (mag_l + mag_r + EPS)
    spec_c = 0.5 * (coh * (1.0 - np.abs(bal))) * (spec_l + spec_r)
    return spec_c, spec_l - spec_c, spec_r - spec_c


# --------------------------------------------------------------------------
# Whole-signal per-band processing (center_extraction.py:426-472)
# --------------------------------------------------------------------------
def band_process_streaming(sig_l: np.ndarray, sig_r: np.ndarray, band: Band):
    """
    Reference-shaped loop: pad, frame at idx = 0, hop, 2*hop, ... < len(padded),
    float32 accumulators, emit first hop, shift, zero tail, final flush, trim.
    (center_extraction.py:426-472 driving :353-409 and :411-424)
    """
    n = band.block_size
    hop = band.hop_size
    total = len(sig_l)
    tail = n - hop
    num_hops = math.ceil((total - tail) / hop)
    padded_len = num_hops * hop + tail
    extra = max(0, padded_len - total)
    pl = np.pad(sig_l, (0, extra), mode="constant")
    pr = np.pad(sig_r, (0, extra), mode="constant")
    g = band_gain(band)

    acc = [np.zeros(n, dtype=np.float32) for _ in range(3)]
    pieces = ([], [], [])
    pos = 0
    while pos < len(pl):
        bl = pl[pos:pos + n]
        br = pr[pos:pos + n]
        if len(bl) < n:
            bl = np.pad(bl, (0, n - len(bl)), mode="constant")
            br = np.pad(br, (0, n - len(br)), mode="constant")
        recs = frames_to_recs(bl, br, band, g)
        for a, rec, out in zip(acc, recs, pieces):
            a += rec
            out.append(a[:hop].copy())
            a[:-hop] = a[hop:]
            a[-hop:] = 0
        pos += hop
    for a, out in zip(acc, pieces):
        out.append(a.copy())
        a[:] = 0
    return tuple(np.concatenate(p)[:total] for p in pieces)


def band_process(sig_l: np.ndarray, sig_r: np.ndarray, band: Band, batch: int = 256,
                 own_len: int = None, out_len: int = None):
    """
    Closed form of the same computation: frames j with
    j*hop < T, input zero-extended on the right, contributions added to the
    float32 output in increasing j.  Returns (c, l, r) float32[T].

    ``own_len`` / ``out_len`` (shard tests only): compute just the frames with
    j*hop < own_len (they may read input past own_len) and return out_len samples.
    """
    n = band.block_size
    hop = band.hop_size
    total = len(sig_l)
    n_frames = ((total if own_len is None else own_len) + hop - 1) // hop
    g = band_gain(band)
    ext = (n_frames - 1) * hop + n if n_frames > 0 else 0
    xl = np.zeros(ext, dtype=np.float64)
    xr = np.zeros(ext, dtype=np.float64)
    m = min(total, ext)
    xl[:m] = sig_l[:m]
    xr[:m] = sig_r[:m]
    outs = [np.zeros(ext, dtype=np.float32) for _ in range(3)]
    starts = np.arange(n_frames) * hop
    col = np.arange(n)
    for b0 in range(0, n_frames, batch):
        idx = starts[b0:b0 + batch, None] + col[None, :]
        recs = frames_to_recs(xl[idx], xr[idx], band, g)
        for j in range(idx.shape[0]):
            s = int(starts[b0 + j])
            for o, rec in zip(outs, recs):
                o[s:s + n] += rec[j]
    want = total if out_len is None else out_len
    res = []
    for o in outs:
        r = np.zeros(want, dtype=np.float32)
        r[:min(want, ext)] = o[:min(want, ext)]
        res.append(r)
    return tuple(res)


# --------------------------------------------------------------------------
# Multi-band drivers (center_extraction.py:477-513, :518-580)
# --------------------------------------------------------------------------
def plan_bands(band_edges: Sequence[float], overlap: float, window: Callable[[int], np.ndarray],
               sr: float, xover_mode: str = "raised_cosine", *, max_block_size: int = 2 ** 16,
               threshold_factor: float = 32, xo_fraction: float = 0.25) -> List[Band]:
    """
    Band planner (center_extraction.py:518-580).  ``max_block_size`` /
    ``threshold_factor`` are the two knobs of compute_block_size_for_low_freq
    (:173) that chain_bands leaves at their defaults (:555); ``xo_fraction`` is
    the 0.25 of :212.
    """
    edges = list(band_edges)
    if edges[-1] < (sr / 2.0):
        edges = edges + [sr / 2.0]
    bands: List[Band] = []
    prev_high_width = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        n = block_size_for_low_freq(lo, sr, max_block_size, threshold_factor)
        width_hi = crossover_width(hi, xo_fraction)
        bands.append(Band(block_size=n, overlap=overlap, f_low=lo, f_high=hi, sr=sr,
                          xover_mode=xover_mode, xover_width_low_hz=prev_high_width,
                          xover_width_high_hz=width_hi, window=window))
        prev_high_width = width_hi
    return bands


def extract_multi_band(sig_l: np.ndarray, sig_r: np.ndarray, bands: Sequence[Band],
                       per_band=band_process):
    """Sum of per-band outputs in list order, float32 (center_extraction.py:503-513).  Returns (C, L, R)."""
    total = len(sig_l)
    fin = [np.zeros(total, dtype=np.float32) for _ in range(3)]
    for band in bands:
        res = per_band(sig_l, sig_r, band)
        for f, r in zip(fin, res):
            f += r
    return tuple(fin)


def extract_multi_band_threadpool(sig_l: np.ndarray, sig_r: np.ndarray, bands: Sequence[Band]):
    """
    The reference's scheduling shape: ThreadPoolExecutor() with default workers,
    one task per band running the sequential frame loop, then the serial float32
    band sum (center_extraction.py:498-513).  Used for bench.py's cpu_baseline.
    """
    with ThreadPoolExecutor() as pool:
        futs = [pool.submit(band_process_streaming, sig_l, sig_r, b) for b in bands]
        results = [f.result() for f in futs]
    total = len(sig_l)
    fin = [np.zeros(total, dtype=np.float32) for _ in range(3)]
    for res in results:
        for f, r in zip(fin, res):
            f += r
    return tuple(fin)


# --------------------------------------------------------------------------
# Caller-side arithmetic of main.py (main.py:47-55, 85-97, 102-157)
# --------------------------------------------------------------------------
def input_peak(wave: np.ndarray) -> float:
    """main.py:53-55"""
    peak = np.max(np.abs(wave))
    return 1e-9 if peak <= 0.0 else peak


def normalise_lcr(center, left, right, peak_in):
    """One global scale so Ls/C/Rs do not exceed the input peak (main.py:85-97).  In place; returns scale."""
    overall = max(np.max(np.abs(left)), np.max(np.abs(center)), np.max(np.abs(right)), 1e-9)
    scale = peak_in / overall
    left *= scale
    center *= scale
    right *= scale
    return scale


def export_layout(mode: str, center, left, right, sig_l=None, sig_r=None):
    """
    Channel layouts per export mode (main.py:110-157).  Returns a dict
    name -> [T,2] array: 'AB' -> {'AB'}, 'split' -> {'Ls','C','Rs'},
    'stereo_sum' -> {'Sum'}; unknown mode -> {} (main.py:159-160).
    """
    if mode == "AB":
        up = left + center + right
        orig = sig_l + sig_r
        m = min(len(up), len(orig))
        return {"AB": np.column_stack([up[:m], orig[:m]])}
    if mode == "split":
        return {"Ls": np.column_stack([left, np.zeros_like(left)]),
                "C": np.column_stack([center, center]),
                "Rs": np.column_stack([np.zeros_like(right), right])}
    if mode == "stereo_sum":
        lch = left + 0.5 * center
        rch = right + 0.5 * center
        m = min(len(lch), len(rch))
        return {"Sum": np.column_stack([lch[:m], rch[:m]])}
    return {}


def band_info_string(bands: Sequence[Band]) -> str:
    """main.py:102-106"""
    return "_".join(f"b{b.block_size}({int(b.f_low)}-{int(b.f_high)})" for b in bands)


def output_names(base: str, mode: str, bands: Sequence[Band], overlap: float):
    """File-name scheme of main.py:117, :131-139, :151."""
    info = band_info_string(bands)
    if mode == "AB":
        return {"AB": f"{base}_AB_{info}_ov{overlap:.2f}.wav"}
    if mode == "split":
        return {k: f"{base}_{k}_{info}.wav" for k in ("Ls", "C", "Rs")}
    if mode == "stereo_sum":
        return {"Sum": f"{base}_Sum_{info}_ov{overlap:.2f}.wav"}
    return {}


# --------------------------------------------------------------------------
# Synthetic input of SURVEY.md section 8(d)
# --------------------------------------------------------------------------
def synthetic_stereo(total: int, seed) -> np.ndarray:
    """Correlated Gaussian stereo, float32 [T,2]: L = 0.1(m + 0.5 s), R = 0.1(m - 0.5 s)."""
    rng = np.random.default_rng(seed)
    m = rng.standard_normal(total)
    s = rng.standard_normal(total)
    out = np.empty((total, 2), dtype=np.float32)
    out[:, 0] = 0.1 * (m + 0.5 * s)
    out[:, 1] = 0.1 * (m - 0.5 * s)
    return out
