#!/usr/bin/env python3
"""
WAV in -> Ls/C/Rs WAV out on an MI355X: the flow of python-prototype/main.py
(load :43, mono->stereo :47-48, peak :53-55, chain_bands :67-73, extract :78-80,
scale :85-97, export :110-160) with the constants the reference asks the user to
edit exposed as arguments.  Defaults reproduce main.py (eyes.wav, stereo_sum,
edges 0/30/120/480/1920/7680, overlap 0.75, Blackman-Harris, raised cosine, STFT up to 65536).
By default the WAV codec, the peak normalisation and the export layouts run on the GPU too
(upx_wav_pipeline); --host-export keeps them in NumPy.

    python -m upmix_amd.cli eyes.wav --export-mode split
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import export, wav
from .extractor import chain_bands, extract_center_left_right_multi_band_in_memory
from .plan import WINDOW_FUNCS

_WRITE_NOTES = {
    "AB": "[AB] Wrote 2-ch => {path}\n  Left  = (Ls + C + Rs)\n  Right = (L + R)\n",
    "Ls": "[Split] Wrote => {path} (Left=Ls, Right=0)",
    "C": "[Split] Wrote => {path} (Left=C, Right=C)",
    "Rs": "[Split] Wrote => {path} (Left=0, Right=Rs)",
    "Sum": "[StereoSum] Wrote 2-ch => {path}\n  Left  = (Ls + C/2)\n  Right = (Rs + C/2)\n",
}


def run(in_filename: str = "eyes.wav", export_mode: str = "stereo_sum", in_dir: str = "in", out_dir: str = "out",
        band_edges=(0, 30, 120, 480, 1920, 7680), overlap: float = 0.75, window: str = "blackman_harris",
        xover_mode: str = "raised_cosine", max_stft: int = 65536, threshold_factor: float = 32,
        xo_fraction: float = 0.25, device: int = 0, subtype: str = "PCM_16", host_export: bool = False,
        reader=wav.read, writer=wav.write):
    """One file through the path; returns {name: path} of the files written."""
    os.makedirs(out_dir, exist_ok=True)
    in_path = os.path.join(in_dir, in_filename)
    if not os.path.isfile(in_path):
        raise FileNotFoundError(f"File not found: {in_path}")
    base_in_name = os.path.splitext(in_filename)[0]
    custom_io = reader is not wav.read or writer is not wav.write
    if not host_export and not custom_io and export_mode in export.EXPORT_MODES:
        try:
            meta = wav.info(in_path)
            kind = wav.device_kind(meta, in_path)
        except ValueError:
            meta = None
        if meta is not None and meta["channels"] in (1, 2) and meta["n_frames"] > 0:
          try:
            return _run_device_codec(meta, kind, int(meta["channels"]), int(meta["rate"]), int(meta["n_frames"]), in_path,
                                     base_in_name, export_mode, out_dir, band_edges, overlap, window, xover_mode, max_stft,
                                     threshold_factor, xo_fraction, device, subtype)
          except _HostFlow:
            pass       # a file the device codec cannot take in chunks: the NumPy flow below
    wave, sr = reader(in_path)
    print(f"Loaded '{in_path}', sr={sr}, shape={wave.shape}")
    if wave.ndim == 1:
        wave = np.column_stack([wave, wave])
    L = wave[:, 0]
    R = wave[:, 1]
    peak_in = export.input_peak(wave)

    band_extractors = chain_bands(list(band_edges), overlap, WINDOW_FUNCS[window], sr, xover_mode,
                                  max_block_size=max_stft, threshold_factor=threshold_factor,
                                  xo_fraction=xo_fraction, device=device)
    final_center, final_left, final_right = extract_center_left_right_multi_band_in_memory(
        L, R, sr, band_extractors, device=device)

    scale_factor, overall_peak = export.scale_to_input_peak(final_center, final_left, final_right, peak_in)
    print(f"Original peak = {peak_in:.4f}, L/C/R peak = {overall_peak:.4f}")
    print(f"Applying scale_factor = {scale_factor:.4f}")

    arrays = export.export_arrays(export_mode, final_center, final_left, final_right, L, R)
    names = export.export_file_names(base_in_name, export_mode, band_extractors, overlap)
    written = {}
    if not arrays:
        print(f"Unknown export_mode '{export_mode}' -- no files written.")
    for key in ("AB", "Ls", "C", "Rs", "Sum"):
        if key in arrays:
            path = os.path.join(out_dir, names[key])
            if writer is wav.write:
                writer(path, arrays[key], sr, subtype)
            else:
                writer(path, arrays[key], sr)
            print(_WRITE_NOTES[key].format(path=path))
            written[key] = path
    print("Done.")
    return written


_SUBTYPE_KIND = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32, "FLOAT": 1032}
LAUNCH_FRAMES = 1 << 29     # a launch indexes at most 2^29 - 1 frames; longer files go through the device in chunks


class _HostFlow(Exception):
    """The device codec cannot take this file: run() continues with the host flow."""


def codec_can_take(n_frames: int, band_extractors, env=os.environ) -> bool:
    """
    Whether upx_wav_pipeline / upx_wav_shard_open accepts a file of n_frames: files of 2^29 frames or more run in chunks
    on the plan's shard grid, which needs hops that share one (every hop divides the largest; overlaps that are not
    powers of two do not) and the chunk schedule switched on (it always is, unless a process that opted into the tuning
    knobs with UPX_TUNING=1 set UPX_WAV_CHUNK=0).  Otherwise the library answers UPX_ERR_INVALID, and the callers take the
    host flow instead of aborting the run.
    """
    if n_frames < LAUNCH_FRAMES:
        return True
    hops = [int(b.hop_size) for b in band_extractors]
    chunks_off = str(env.get("UPX_TUNING", "")).strip() == "1" and str(env.get("UPX_WAV_CHUNK", "")).strip() == "0"
    return all(max(hops) % h == 0 for h in hops) and not chunks_off


def _print_plan(band_extractors) -> None:
    """The per-band lines chain_bands prints (center_extraction.py:562-566), for a plan that was built quietly."""
    for i, b in enumerate(band_extractors):
        print(f"[Band {i+1}] f_low={b.f_low:.1f} Hz, f_high={b.f_high:.1f} Hz, block_size={b.block_size}, "
              f"xover_low={b.xover_width_low_hz:.1f} Hz, xover_high={b.xover_width_high_hz:.1f} Hz")


def _run_device_codec(meta, kind, channels, sr, n_frames, in_path, base_in_name, export_mode, out_dir, band_edges,
                      overlap, window, xover_mode, max_stft, threshold_factor, xo_fraction, device, subtype):
    """Same flow with decode, peak scale, export layout and quantisation on the GPU (upx_wav_pipeline); the file's sample
    bytes are read, undecoded, straight into page-locked memory, so the chunks' uploads run beside their kernels."""
    from .extractor import DevicePlan
    band_extractors = chain_bands(list(band_edges), overlap, WINDOW_FUNCS[window], sr, xover_mode,
                                  max_block_size=max_stft, threshold_factor=threshold_factor,
                                  xo_fraction=xo_fraction, device=device, verbose=False)
    if not codec_can_take(n_frames, band_extractors):
        raise _HostFlow()
    print(f"Loaded '{in_path}', sr={sr}, shape={(n_frames, channels) if channels > 1 else (n_frames,)}")
    _print_plan(band_extractors)
    plan = DevicePlan(band_extractors, device)
    try:
        block = meta["bits"] // 8 * channels
        raw = wav.read_raw_range(in_path, 0, n_frames, meta, out=plan.host_empty(n_frames * block))
        payloads, stats = plan.wav_pipeline(raw, kind, channels, n_frames, export_mode, _SUBTYPE_KIND[subtype])
    finally:
        plan.close()
    print(f"Original peak = {stats['peak_in']:.4f}, L/C/R peak = {stats['overall_peak']:.4f}")
    print(f"Applying scale_factor = {stats['scale_factor']:.4f}")
    names = export.export_file_names(base_in_name, export_mode, band_extractors, overlap)
    written = {}
    for key in ("AB", "Ls", "C", "Rs", "Sum"):
        if key in payloads:
            path = os.path.join(out_dir, names[key])
            wav.write_raw(path, payloads[key], sr, _SUBTYPE_KIND[subtype], 2)
            print(_WRITE_NOTES[key].format(path=path))
            written[key] = path
    print("Done.")
    return written


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="upmix_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("in_filename", nargs="?", default="eyes.wav", help="WAV name inside --in-dir")
    ap.add_argument("--export-mode", default="stereo_sum", help="AB | split | stereo_sum")
    ap.add_argument("--in-dir", default="in")
    ap.add_argument("--out-dir", default="out")
    ap.add_argument("--band-edges", default="0,30,120,480,1920,7680", help="comma-separated Hz")
    ap.add_argument("--overlap", type=float, default=0.75)
    ap.add_argument("--window", default="blackman_harris", choices=sorted(WINDOW_FUNCS))
    ap.add_argument("--xover-mode", default="raised_cosine")
    ap.add_argument("--max-stft", type=int, default=65536, help="max STFT size (reference: 65536; BASELINE configs use 8192)")
    ap.add_argument("--host-export", action="store_true", help="decode / scale / export with NumPy on the host instead of on the GPU")
    ap.add_argument("--threshold-factor", type=float, default=32)
    ap.add_argument("--xo-fraction", type=float, default=0.25)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--subtype", default="PCM_16", choices=["PCM_16", "PCM_24", "PCM_32", "FLOAT"])
    a = ap.parse_args(argv)
    run(a.in_filename, a.export_mode, a.in_dir, a.out_dir, [float(v) for v in a.band_edges.split(",")], a.overlap,
        a.window, a.xover_mode, a.max_stft, a.threshold_factor, a.xo_fraction, a.device, a.subtype, a.host_export)
    return 0


if __name__ == "__main__":
    sys.exit(main())
