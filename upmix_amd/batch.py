#!/usr/bin/env python3
"""
A batch of independent tracks over the GPUs of a node (BASELINE configs[4], SURVEY.md 8(e): "replicas only").

The reference handles one file per ``python main.py`` run (main.py:36-80: load, chain_bands, extract, scale,
export); a batch is that flow once per file.  Here every rank (one process per GPU) takes the tracks
``rank, rank + world, rank + 2 world, ...`` and pushes them through ONE band plan on its GPU.  The file entry
(``main``) runs each track through the device codec (``upx_wav_pipeline``: raw samples up, decode, all bands, peak
scale, export layout and quantisation on the GPU, final 2-channel data down) while a reader thread fetches the next
file into page-locked memory and a writer thread stores the previous result; ``process_tracks_rank`` is the
in-memory form (``upx_process_tracks``: uploads, kernels and downloads of consecutive tracks overlap).  There is no
data-path communication between ranks: no collective, no seam, no process group.

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

from . import cli, export, wav
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


_SUBTYPE_KIND = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32, "FLOAT": 1032}


def _host_flow(plan, path, meta, bands, a, out_dir, rank):
    """main.py:43-157 with NumPy around the kernels: --host-export, more than two channels, formats the device codec
    does not read."""
    wave, sr = wav.read(path)
    if wave.ndim == 1:
        wave = np.column_stack([wave, wave])               # main.py:47-48
    c, l, r = plan.process(np.ascontiguousarray(wave[:, :2], dtype=np.float32))   # main.py:49-50: columns 0 and 1
    peak_in = export.input_peak(wave)                      # over every channel, like main.py:53
    scale_factor, overall_peak = export.scale_to_input_peak(c, l, r, peak_in)
    arrays = export.export_arrays(a.export_mode, c, l, r, wave[:, 0], wave[:, 1])
    names = export.export_file_names(os.path.splitext(os.path.basename(path))[0], a.export_mode, bands, a.overlap)
    jobs = [(os.path.join(out_dir, names[k]), arr) for k, arr in arrays.items()]
    return (peak_in, overall_peak, scale_factor), [lambda p=p, arr=arr: wav.write(p, arr, sr, a.subtype) for p, arr in jobs], \
        [p for p, _ in jobs]


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
    ap.add_argument("--host-export", action="store_true", help="decode / scale / export with NumPy on the host")
    a = ap.parse_args(argv)

    from concurrent.futures import ThreadPoolExecutor
    from .extractor import DevicePlan, chain_bands
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
    metas = {i: wav.info(paths[i]) for i in mine}
    known_mode = a.export_mode in export.EXPORT_MODES
    if not known_mode:
        print(f"Unknown export_mode '{a.export_mode}' -- no files written.")   # main.py:159-160

    with ThreadPoolExecutor(max_workers=1) as reader, ThreadPoolExecutor(max_workers=1) as writer:
        pending = []
        for sr in sorted({m["rate"] for m in metas.values()}):
            group = [i for i in mine if metas[i]["rate"] == sr]
            bands = chain_bands(edges, a.overlap, WINDOW_FUNCS[a.window], sr, a.xover_mode, max_block_size=a.max_stft,
                                threshold_factor=a.threshold_factor, xo_fraction=a.xo_fraction, device=device,
                                verbose=rank == 0)
            plan = DevicePlan(bands, device)
            ahead = None
            try:
                def on_device(i):
                    m = metas[i]
                    try:
                        wav.device_kind(m, paths[i])
                    except ValueError:
                        return False
                    # (files of 2^29 frames or more need a plan that can chunk: cli.codec_can_take; else the host flow)
                    return (known_mode and not a.host_export and m["channels"] in (1, 2) and m["n_frames"] > 0 and
                            cli.codec_can_take(m["n_frames"], bands))

                def fetch(i):   # reader thread: the file's sample bytes, undecoded, into page-locked memory
                    m = metas[i]
                    block = m["bits"] // 8 * m["channels"]
                    return wav.read_raw_range(paths[i], 0, m["n_frames"], m, out=plan.host_empty(m["n_frames"] * block))

                ahead = reader.submit(fetch, group[0]) if group and on_device(group[0]) else None
                for pos, i in enumerate(group):
                    m = metas[i]
                    shape = (m["n_frames"], m["channels"]) if m["channels"] > 1 else (m["n_frames"],)
                    print(f"[rank {rank}] Loaded '{paths[i]}', sr={sr}, shape={shape}")
                    if on_device(i):
                        raw = ahead.result()
                        nxt = group[pos + 1] if pos + 1 < len(group) else None
                        ahead = reader.submit(fetch, nxt) if nxt is not None and on_device(nxt) else None
                        payloads, st = plan.wav_pipeline(raw, wav.device_kind(m), m["channels"], m["n_frames"],
                                                         a.export_mode, _SUBTYPE_KIND[a.subtype])
                        stats = (st["peak_in"], st["overall_peak"], st["scale_factor"])
                        names = export.export_file_names(os.path.splitext(os.path.basename(paths[i]))[0], a.export_mode,
                                                         bands, a.overlap)
                        outs = [os.path.join(a.out_dir, names[k]) for k in payloads]
                        jobs = [lambda p=p, b=b, sr=sr: wav.write_raw(p, b, sr, _SUBTYPE_KIND[a.subtype], 2)
                                for p, b in zip(outs, payloads.values())]
                    else:
                        if pos + 1 < len(group) and on_device(group[pos + 1]) and ahead is None:
                            ahead = reader.submit(fetch, group[pos + 1])
                        stats, jobs, outs = _host_flow(plan, paths[i], m, bands, a, a.out_dir, rank) if known_mode else \
                            ((float("nan"),) * 3, [], [])
                    if known_mode:
                        print(f"[rank {rank}] {os.path.basename(paths[i])}: Original peak = {stats[0]:.4f}, "
                              f"L/C/R peak = {stats[1]:.4f}, scale_factor = {stats[2]:.4f}")
                    for job, out in zip(jobs, outs):   # writer thread: the previous track's files while the next one runs
                        pending.append((writer.submit(job), out))
                for fut, out in pending:
                    fut.result()
                    print(f"[rank {rank}] Wrote => {out}")
                pending.clear()
            finally:
                # nothing may still be using the plan when it is destroyed (round-3 advisor finding: on the error path the
                # reader thread could be inside fetch() -> plan.host_empty -> upx_host_alloc(plan)): wait for the read
                # ahead and for the writer jobs, whatever they end with, before the plan goes
                for fut in ([ahead] if ahead is not None else []) + [f for f, _ in pending]:
                    try:
                        fut.result()
                    except Exception:   # noqa: BLE001 - the exception that brought us here is the one that propagates
                        pass
                pending.clear()
                plan.close()
    print(f"[rank {rank}] Done ({len(mine)} of {len(paths)} tracks).")
    return 0


if __name__ == "__main__":
    sys.exit(main())
