"""
upmix_amd - MI355X-native (gfx950) implementation of the multi-band STFT
centre-extraction hot path of willleskowitz/upmix, behind the reference's own
Python call surface.  Hand-written HIP kernels through a C ABI
(include/upmix_hip.h, libupmix_hip.so); no PyTorch, no CPU fallback.
"""
from .plan import (EPS, WINDOW_FUNCS, band_limit_gain, compute_block_size_for_low_freq,
                   design_wola_synthesis_window, freq_to_bin, hp_freq_to_crossover_width, make_blackman,
                   make_blackman_harris, make_hamming, make_hann, make_rect, make_sqrt_hann, next_power_of_2)
from .extractor import (DevicePlan, MultiBandExtractorAccu, chain_bands,
                        extract_center_left_right_multi_band_in_memory, process_tracks)

__all__ = [
    "EPS", "WINDOW_FUNCS", "band_limit_gain", "compute_block_size_for_low_freq", "design_wola_synthesis_window",
    "freq_to_bin", "hp_freq_to_crossover_width", "make_blackman", "make_blackman_harris", "make_hamming",
    "make_hann", "make_rect", "make_sqrt_hann", "next_power_of_2", "DevicePlan", "MultiBandExtractorAccu",
    "chain_bands", "extract_center_left_right_multi_band_in_memory", "process_tracks",
]
