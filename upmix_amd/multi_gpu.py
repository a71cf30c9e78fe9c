#!/usr/bin/env python3
"""
One long WAV across the GPUs of a node: one process per GPU, time shards, one RCCL all-reduce for the
overlap-add seam (BASELINE configs[3]).  Launch with any one-process-per-GPU launcher that sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, e.g.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m upmix_amd.multi_gpu eyes.wav --export-mode stereo_sum

Every rank reads ONLY its own time shard (+ the right halo) from the input file and writes ONLY its own slice of
every output file, at that slice's byte offset (rank 0 writes the headers first); nothing but two scalars per rank
(input peak, output peak: the one global scale of main.py:85-97) and the RCCL unique id crosses the process group
(torch.distributed gloo), and nothing but the overlap-add seam crosses RCCL.  The export arithmetic is main.py's
(:110-157), on each rank's slice.
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Callable, Optional

import numpy as np

from . import export, sharding, wav
from .plan import WINDOW_FUNCS


def run_rank(in_path: str, out_dir: str, export_mode: str, bands, overlap: float, subtype: str, rank: int, world: int,
             dist, engine: Optional[Callable] = None, device: int = 0, log=print):
    """
    One rank's part of the job; returns {key: path} of the files (written by all ranks together).
    `engine(local_stereo, shard, geo) -> (center, left, right)` for the samples the shard owns, seam included;
    default: this rank's GPU (DevicePlan + RCCL seam).  The CPU tests plug the oracle + a gloo seam in.
    `dist` = an initialised torch.distributed module (any backend that can all-reduce CPU tensors), or None if world == 1.
    """
    meta = wav.info(in_path)
    total, sr, channels = meta["n_frames"], meta["rate"], meta["channels"]
    geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
    shard = geo.plan(total, world)[rank]
    local = wav.read_range(in_path, shard.start, shard.t_in, meta)        # own range + right halo, nothing else
    if local.ndim == 1:
        local = np.column_stack([local, local])                            # main.py:47-48
    own = local[:shard.own_len]

    plan = seam = None
    if engine is None:
        from .extractor import DevicePlan
        plan = DevicePlan(bands, device)
        if world > 1:
            seam = sharding.RcclSeam(plan, rank, world, broadcast=lambda b: sharding.broadcast_bytes_gloo(dist, b))
        c, l, r = sharding.process_local_shard(plan, local, shard, geo, world, seam)
    else:
        c, l, r = engine(local, shard, geo)

    # one global scale (main.py:53-55, :85-97): max over the ranks of the input peak and of the output peak
    peaks = np.array([float(np.max(np.abs(own), initial=0.0)),
                      max(float(np.max(np.abs(c), initial=0.0)), float(np.max(np.abs(l), initial=0.0)),
                          float(np.max(np.abs(r), initial=0.0)))], dtype=np.float64)
    if world > 1:
        import torch   # only with a process group: a single rank never loads torch next to libupmix_hip.so
        t = torch.from_numpy(peaks)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    peak_in = float(peaks[0]) if float(peaks[0]) > 0.0 else 1e-9
    overall_peak = max(float(peaks[1]), 1e-9)
    # main.py:90-97: peak_in and the plane maxima are NumPy scalars there, so scale_factor is a float64 NumPy scalar and
    # `final_x *= scale_factor` multiplies in float64 and rounds once to float32 (a Python float would multiply in float32)
    scale_factor = np.float64(peak_in / overall_peak)
    for p in (c, l, r):
        p *= scale_factor
    arrays = export.export_arrays(export_mode, c, l, r, own[:, 0], own[:, 1])
    names = export.export_file_names(os.path.splitext(os.path.basename(in_path))[0], export_mode, bands, overlap)
    written = {}
    if rank == 0:
        log(f"Original peak = {peak_in:.4f}, L/C/R peak = {overall_peak:.4f}")
        log(f"Applying scale_factor = {scale_factor:.4f}")
        os.makedirs(out_dir, exist_ok=True)
        if not arrays:
            log(f"Unknown export_mode '{export_mode}' -- no files written.")     # main.py:159-160
        for key in arrays:
            wav.create(os.path.join(out_dir, names[key]), total, sr, subtype, 2)   # header + full size
    if world > 1:
        dist.barrier()                                                      # headers exist before anyone writes a slice
    for key, arr in arrays.items():
        path = os.path.join(out_dir, names[key])
        code, bits, payload = wav.encode(arr, subtype)
        block = 2 * bits // 8
        data_offset = wav.info(path)["data_offset"]
        wav.write_at(path, data_offset + shard.start * block, payload)
        written[key] = path
    if world > 1:
        dist.barrier()
    if rank == 0:
        for key, path in written.items():
            log(f"Wrote => {path}")
        log("Done.")
    if seam is not None:
        seam.close()
    if plan is not None:
        plan.close()
    return written


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="upmix_amd.multi_gpu")
    ap.add_argument("in_filename")
    ap.add_argument("--export-mode", default="stereo_sum", choices=list(export.EXPORT_MODES))
    ap.add_argument("--in-dir", default="in")
    ap.add_argument("--out-dir", default="out")
    ap.add_argument("--band-edges", default="0,30,120,480,1920,7680")
    ap.add_argument("--overlap", type=float, default=0.75)
    ap.add_argument("--window", default="blackman_harris", choices=sorted(WINDOW_FUNCS))
    ap.add_argument("--xover-mode", default="raised_cosine")
    ap.add_argument("--max-stft", type=int, default=8192)
    ap.add_argument("--subtype", default="PCM_16", choices=["PCM_16", "PCM_24", "PCM_32", "FLOAT"])
    a = ap.parse_args(argv)

    from .extractor import chain_bands
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    in_path = os.path.join(a.in_dir, a.in_filename)
    if not os.path.isfile(in_path):
        raise FileNotFoundError(f"File not found: {in_path}")              # main.py:40-41
    sr = wav.info(in_path)["rate"]
    bands = chain_bands([float(v) for v in a.band_edges.split(",")], a.overlap, WINDOW_FUNCS[a.window], sr,
                        a.xover_mode, max_block_size=a.max_stft, device=local_rank, verbose=rank == 0)
    run_rank(in_path, a.out_dir, a.export_mode, bands, a.overlap, a.subtype, rank, world, dist, device=local_rank,
             log=print if rank == 0 else (lambda *_: None))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    rc = main()
    sys.stdout.flush()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # torch (its own bundled HIP runtime) and libupmix_hip.so (the system one) share this process: skip the
        # interpreter's teardown, where the two runtimes' exit handlers can collide
        os._exit(rc)
    sys.exit(rc)
