#!/usr/bin/env python3
"""
A batch of independent tracks over the GPUs of a node (BASELINE configs[4], SURVEY.md 8(e): "replicas only").

The reference handles one file per ``python main.py`` run (main.py:36-80: load, chain_bands, extract, scale,
export); a batch is that flow once per file.  Here every rank (one process per GPU) takes the tracks
``rank, rank + world, rank + 2 world, ...`` and pushes them through ONE band plan on its GPU with
``upx_process_tracks`` (uploads, kernels and downloads of consecutive tracks overlap).  There is no data-path
communication between ranks: no collective, no seam.

    python -m upmix_amd.batch a.wav b.wav c.wav --export-mode stereo_sum
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        -m upmix_amd.batch in/*.wav

RANK / WORLD_SIZE / LOCAL_RANK come from the launcher's environment (absent: one rank).  Tracks of one call
share sample rate and band plan; files with another sample rate get their own plan (grouped per rate).
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import export, wav
from .plan import WINDOW_FUNCS


def assign_tracks(n_tracks: int, rank: int, world: int) -> List[int]:
    """Indices of the tracks rank `rank` of `world` processes: round-robin, so unequal lengths spread evenly."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of {world}")
    return list(range(rank, n_tracks, world))


def process_tracks_rank(tracks: Sequence[np.ndarray], band_extractors, rank: int, world: int, *,
                        device: Optional[int] = None,
                        engine: Optional[Callable[[List[np.ndarray]], List[tuple]]] = None) -> Dict[int, tuple]:
    """
    This rank's share of a batch: {track index: (center, left, right)}.  `tracks` may hold None for tracks the
    caller did not load (only the indices of assign_tracks are touched).  `engine(list_of_stereo)` defaults to the
    HIP library (extractor.process_tracks on `device`, default LOCAL_RANK / rank); the CPU tests plug the oracle in.
    """
    mine = assign_tracks(len(tracks), rank, world)
    if engine is None:
        from .extractor import process_tracks
        dev = int(os.environ.get("LOCAL_RANK", rank)) if device is None else int(device)
        engine = lambda xs: process_tracks(xs, band_extractors, device=dev)   # noqa: E731
    results = engine([tracks[i] for i in mine]) if mine else []
    return dict(zip(mine, results))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="upmix_amd.batch", description=__doc__.split("\n\n")[0])
    ap.add_argument("files", nargs="+", help="WAV files (paths, or names inside --in-dir)")
    ap.add_argument("--export-mode", default="stereo_sum", help="AB | split | stereo_sum")
    ap.add_argument("--in-dir", default="")
    ap.add_argument("--out-dir", default="out")
    ap.add_argument("--band-edges", default="0,30,120,480,1920,7680", help="comma-separated Hz")
    ap.add_argument("--overlap", type=float, default=0.75)
    ap.add_argument("--window", default="blackman_harris", choices=sorted(WINDOW_FUNCS))
    ap.add_argument("--xover-mode", default="raised_cosine")
    ap.add_argument("--max-stft", type=int, default=65536)
    ap.add_argument("--threshold-factor", type=float, default=32)
    ap.add_argument("--xo-fraction", type=float, default=0.25)
    ap.add_argument("--subtype", default="PCM_16", choices=["PCM_16", "PCM_24", "PCM_32", "FLOAT"])
    a = ap.parse_args(argv)

    from .extractor import chain_bands, process_tracks
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = int(os.environ.get("LOCAL_RANK", "0"))
    paths = [os.path.join(a.in_dir, f) if a.in_dir else f for f in a.files]
    for p in paths:
        if not os.path.isfile(p):
            raise FileNotFoundError(f"File not found: {p}")   # main.py:40-41
    mine = assign_tracks(len(paths), rank, world)
    os.makedirs(a.out_dir, exist_ok=True)
    edges = [float(v) for v in a.band_edges.split(",")]

    loaded = {}
    for i in mine:
        wave, sr = wav.read(paths[i])
        if wave.ndim == 1:
            wave = np.column_stack([wave, wave])           # main.py:47-48
        loaded[i] = (wave, sr)
        print(f"[rank {rank}] Loaded '{paths[i]}', sr={sr}, shape={wave.shape}")
    for sr in sorted({sr for _, sr in loaded.values()}):
        group = [i for i in mine if loaded[i][1] == sr]
        bands = chain_bands(edges, a.overlap, WINDOW_FUNCS[a.window], sr, a.xover_mode, max_block_size=a.max_stft,
                            threshold_factor=a.threshold_factor, xo_fraction=a.xo_fraction, device=device,
                            verbose=rank == 0)
        results = process_tracks([loaded[i][0] for i in group], bands, device=device)
        for i, (c, l, r) in zip(group, results):
            wave = loaded[i][0]
            peak_in = export.input_peak(wave)
            scale_factor, overall_peak = export.scale_to_input_peak(c, l, r, peak_in)
            print(f"[rank {rank}] {os.path.basename(paths[i])}: Original peak = {peak_in:.4f}, "
                  f"L/C/R peak = {overall_peak:.4f}, scale_factor = {scale_factor:.4f}")
            arrays = export.export_arrays(a.export_mode, c, l, r, wave[:, 0], wave[:, 1])
            if not arrays:
                print(f"Unknown export_mode '{a.export_mode}' -- no files written.")   # main.py:159-160
            names = export.export_file_names(os.path.splitext(os.path.basename(paths[i]))[0], a.export_mode, bands,
                                             a.overlap)
            for key, arr in arrays.items():
                path = os.path.join(a.out_dir, names[key])
                wav.write(path, arr, sr, a.subtype)
                print(f"[rank {rank}] Wrote => {path}")
    print(f"[rank {rank}] Done ({len(mine)} of {len(paths)} tracks).")
    return 0


if __name__ == "__main__":
    sys.exit(main())
