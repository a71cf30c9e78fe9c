"""
Caller-side arithmetic of the reference driver (python-prototype/main.py):
input peak (:53-55), one global scale for Ls/C/Rs (:85-97), the three export
layouts (:110-157) and the output file names (:102-108, :117, :131-139, :151).
Pure NumPy on the planes the GPU returns; pinned by fixture F7.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np

EXPORT_MODES = ("AB", "split", "stereo_sum")


def input_peak(wave: np.ndarray) -> float:
    """max |wave| over both channels, 1e-9 for a silent file.  main.py:53-55"""
    peak = np.max(np.abs(wave))
    if peak <= 0.0:
        peak = 1e-9
    return peak


def scale_to_input_peak(final_center: np.ndarray, final_left: np.ndarray, final_right: np.ndarray, peak_in: float):
    """In-place scale so max(|Ls|,|C|,|Rs|) == peak_in; returns (scale_factor, overall_peak).  main.py:85-97"""
    overall_peak = max(np.max(np.abs(final_left)), np.max(np.abs(final_center)), np.max(np.abs(final_right)), 1e-9)
    scale_factor = peak_in / overall_peak
    final_left *= scale_factor
    final_center *= scale_factor
    final_right *= scale_factor
    return scale_factor, overall_peak


def band_info_str(band_extractors: Sequence) -> str:
    """'b{N}({lo}-{hi})' joined by '_'.  main.py:102-106"""
    return "_".join(f"b{b.block_size}({int(b.f_low)}-{int(b.f_high)})" for b in band_extractors)


def export_arrays(export_mode: str, final_center, final_left, final_right, L=None, R=None) -> Dict[str, np.ndarray]:
    """
    Channel layouts: AB = [Ls+C+Rs, L+R]; split = Ls:[Ls,0], C:[C,C], Rs:[0,Rs];
    stereo_sum = [Ls + C/2, Rs + C/2]; anything else -> {} (main.py:110-160).
    """
    if export_mode == "AB":
        upmix_sum = final_left + final_center + final_right
        orig_sum = L + R
        n = min(len(upmix_sum), len(orig_sum))
        return {"AB": np.column_stack([upmix_sum[:n], orig_sum[:n]])}
    if export_mode == "split":
        return {
            "Ls": np.column_stack([final_left, np.zeros_like(final_left)]),
            "C": np.column_stack([final_center, final_center]),
            "Rs": np.column_stack([np.zeros_like(final_right), final_right]),
        }
    if export_mode == "stereo_sum":
        left_ch = final_left + 0.5 * final_center
        right_ch = final_right + 0.5 * final_center
        n = min(len(left_ch), len(right_ch))
        return {"Sum": np.column_stack([left_ch[:n], right_ch[:n]])}
    return {}


def export_file_names(base_in_name: str, export_mode: str, band_extractors: Sequence, overlap: float) -> Dict[str, str]:
    """main.py:117 (AB), :131-139 (split), :151 (stereo_sum)."""
    info = band_info_str(band_extractors)
    if export_mode == "AB":
        return {"AB": f"{base_in_name}_AB_{info}_ov{overlap:.2f}.wav"}
    if export_mode == "split":
        return {k: f"{base_in_name}_{k}_{info}.wav" for k in ("Ls", "C", "Rs")}
    if export_mode == "stereo_sum":
        return {"Sum": f"{base_in_name}_Sum_{info}_ov{overlap:.2f}.wav"}
    return {}
