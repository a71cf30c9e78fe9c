"""
Host-side band planning for the MI355X upmix path: windows, WOLA synthesis
window, plan arithmetic and the band-limit gain vector.  These are O(N) setup
computations that the reference also does once per band on the host
(center_extraction.py:42-105, :142-212, :273-351); their results cross the C ABI
as float arrays.  Expressions follow the reference exactly where rounding
matters (half-to-even bin rounding, int() truncation of the hop, float32
accumulation of the window power sum).
"""
from __future__ import annotations

import numpy as np

EPS = 1e-12   # center_extraction.py:36


# ---- windows: Callable[[int], float32[N]]  (center_extraction.py:42-75) ----
def make_blackman_harris(N: int) -> np.ndarray:
    """4-term Blackman-Harris, symmetric (denominator N-1).  center_extraction.py:42-53"""
    ph = 2.0 * np.pi * np.arange(N) / (N - 1)
    w = 0.35875 - 0.48829 * np.cos(ph) + 0.14128 * np.cos(2 * ph) - 0.01168 * np.cos(3 * ph)
    return w.astype(np.float32)


def make_sqrt_hann(N: int) -> np.ndarray:
    """center_extraction.py:56-59"""
    return np.sqrt(np.hanning(N)).astype(np.float32)


def make_hann(N: int) -> np.ndarray:
    """center_extraction.py:61-63"""
    return np.hanning(N).astype(np.float32)


def make_blackman(N: int) -> np.ndarray:
    """center_extraction.py:65-67"""
    return np.blackman(N).astype(np.float32)


def make_hamming(N: int) -> np.ndarray:
    """center_extraction.py:69-71"""
    return np.hamming(N).astype(np.float32)


def make_rect(N: int) -> np.ndarray:
    """center_extraction.py:73-75"""
    return np.ones(N, dtype=np.float32)


WINDOW_FUNCS = {
    "blackman_harris": make_blackman_harris,
    "sqrt_hann": make_sqrt_hann,
    "hann": make_hann,
    "blackman": make_blackman,
    "hamming": make_hamming,
    "rect": make_rect,
}


def design_wola_synthesis_window(analysis_window: np.ndarray, overlap: float) -> np.ndarray:
    """
    w_S(n) = w_A(n) / (sum_k w_A(n + k*hop mod L)^2 + EPS), k < K = round(1/(1-overlap)),
    hop = int(L*(1-overlap)).  center_extraction.py:80-105.  ValueError if hop < 1 (:91-92).
    Accumulated in the window's dtype (float32) in k order like the reference
    under NumPy >= 2; squares use scalar ``**`` as the reference does.
    """
    L = len(analysis_window)
    hop = int(L * (1.0 - overlap))
    if hop < 1:
        raise ValueError("Overlap too large; resulting hop size < 1.")
    K = int(round(1.0 / (1.0 - overlap)))
    w = np.asarray(analysis_window)
    power = np.array([v ** 2 for v in w], dtype=w.dtype)
    n = np.arange(L)
    total = np.zeros(L, dtype=w.dtype)
    for k in range(K):
        total = total + power[(n + k * hop) % L]
    return (w / (total + np.asarray(EPS, dtype=w.dtype))).astype(w.dtype)


# ---- plan arithmetic (center_extraction.py:142-212) ------------------------
def freq_to_bin(freq_hz: float, sr: float, fft_size: int) -> int:
    """Nearest rFFT bin, Python round() (half to even), unclamped.  center_extraction.py:142-154"""
    return int(round(freq_hz / (sr / float(fft_size))))


def next_power_of_2(x: int) -> int:
    """center_extraction.py:156-171"""
    if x < 1:
        return 1
    return 1 << (int(x) - 1).bit_length()


def compute_block_size_for_low_freq(f_low: float, sr: float, max_block_size: int = 2 ** 16,
                                    threshold_factor: float = 32) -> int:
    """nextpow2(ceil(sr*threshold_factor/f_low)) clamped to max_block_size.  center_extraction.py:173-197"""
    if f_low <= 0.0:
        return max_block_size
    threshold = (sr * threshold_factor) / f_low
    return min(next_power_of_2(int(np.ceil(threshold))), max_block_size)


def hp_freq_to_crossover_width(hp_freq: float, fraction: float = 0.25) -> float:
    """Fade width = 25 % of the crossover frequency.  center_extraction.py:200-212"""
    return hp_freq * fraction


# ---- band limiter as a gain vector (center_extraction.py:273-351) ----------
def band_limit_gain(block_size: int, sr: float, f_low: float, f_high: float, xover_mode: str,
                    xover_width_low_hz: float, xover_width_high_hz: float) -> np.ndarray:
    """
    float64[N/2+1] factor that ``_band_limit`` multiplies both spectra by:
    hard zero outside [bin_low, bin_high] (:273-280; also any unknown mode,
    :349-351), or raised-cosine fades that lie OUTSIDE the pass band (:282-332).
    """
    n_bins = block_size // 2 + 1
    fft_size = (n_bins - 1) * 2
    b_lo = freq_to_bin(f_low, sr, fft_size)
    b_hi = freq_to_bin(f_high, sr, fft_size)
    if b_lo > b_hi:
        b_lo, b_hi = b_hi, b_lo
    gain = np.ones(n_bins, dtype=np.float64)
    if xover_mode != "raised_cosine":
        gain[:b_lo] = 0.0
        gain[b_hi + 1:] = 0.0
        return gain
    b_lo = max(b_lo, 0)
    b_hi = min(b_hi, n_bins - 1)
    if b_lo > b_hi:
        gain[:] = 0.0
        return gain
    if f_low > 0:
        first = max(0, b_lo - freq_to_bin(xover_width_low_hz, sr, fft_size))
        gain[:first] = 0.0
        count = b_lo - first
        if count > 0:
            ramp = (np.arange(count) + 0.5) / count
            gain[first:b_lo] *= np.array([0.5 * (1.0 - np.cos(np.pi * x)) for x in ramp])
    if f_high < sr * 0.5 and b_hi + 1 < n_bins:
        start = b_hi + 1
        stop = min(start + freq_to_bin(xover_width_high_hz, sr, fft_size), n_bins)
        count = stop - start
        if count > 0:
            ramp = (np.arange(count) + 0.5) / count
            gain[start:stop] *= np.array([0.5 * (1.0 + np.cos(np.pi * x)) for x in ramp])
        if stop < n_bins:
            gain[stop:] = 0.0
    return gain
