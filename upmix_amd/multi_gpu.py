#!/usr/bin/env python3
"""
One long WAV across the GPUs of a node: one process per GPU, time shards, one RCCL all-reduce for the
overlap-add seam (BASELINE configs[3]).  Launch with any one-process-per-GPU launcher that sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, e.g.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m upmix_amd.multi_gpu eyes.wav --export-mode stereo_sum

torch.distributed (gloo) is used for the rendezvous only (RCCL unique id, global peak, gathering the shards on
rank 0); all GPU work and the seam exchange go through libupmix_hip.so.  The export arithmetic is main.py's
(one global scale from the global peak, main.py:85-97).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import export, sharding, wav
from .extractor import DevicePlan, chain_bands
from .plan import WINDOW_FUNCS


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

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group("gloo", rank=rank, world_size=world)

    in_path = os.path.join(a.in_dir, a.in_filename)
    if not os.path.isfile(in_path):
        raise FileNotFoundError(f"File not found: {in_path}")
    wave, sr = wav.read(in_path)                       # every rank decodes; only its shard goes to its GPU
    if wave.ndim == 1:
        wave = np.column_stack([wave, wave])
    bands = chain_bands([float(v) for v in a.band_edges.split(",")], a.overlap, WINDOW_FUNCS[a.window], sr,
                        a.xover_mode, max_block_size=a.max_stft, device=local_rank, verbose=rank == 0)
    plan = DevicePlan(bands, local_rank)
    seam = None
    if world > 1:
        seam = sharding.RcclSeam(plan, rank, world, broadcast=lambda b: sharding.broadcast_bytes_gloo(dist, b))
    shard, (c, l, r) = sharding.process_rank(plan, wave, rank, world, seam)

    # one global scale (main.py:85-97): global max over the ranks
    peak = torch.tensor([max(float(np.max(np.abs(c), initial=0.0)), float(np.max(np.abs(l), initial=0.0)),
                             float(np.max(np.abs(r), initial=0.0)))], dtype=torch.float64)
    dist.all_reduce(peak, op=dist.ReduceOp.MAX)
    peak_in = export.input_peak(wave)
    overall_peak = max(float(peak.item()), 1e-9)
    scale_factor = peak_in / overall_peak
    for p in (c, l, r):
        p *= scale_factor
    arrays = export.export_arrays(a.export_mode, c, l, r, wave[shard.start:shard.start + shard.own_len, 0],
                                  wave[shard.start:shard.start + shard.own_len, 1])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object({k: v for k, v in arrays.items()}, gathered, dst=0)
    if rank == 0:
        print(f"Original peak = {peak_in:.4f}, L/C/R peak = {overall_peak:.4f}")
        print(f"Applying scale_factor = {scale_factor:.4f}")
        os.makedirs(a.out_dir, exist_ok=True)
        names = export.export_file_names(os.path.splitext(a.in_filename)[0], a.export_mode, bands, a.overlap)
        for key, fname in names.items():
            full = np.concatenate([g[key] for g in gathered], axis=0)
            path = os.path.join(a.out_dir, fname)
            wav.write(path, full, sr, a.subtype)
            print(f"Wrote => {path}")
        print("Done.")
    if seam is not None:
        seam.close()
    plan.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
