#!/usr/bin/env python3
"""
One long WAV across the GPUs of a node: one process per GPU, time shards, one RCCL all-reduce for the
overlap-add seam (BASELINE configs[3]).  Launch with any one-process-per-GPU launcher that sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, e.g.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m upmix_amd.multi_gpu eyes.wav --export-mode stereo_sum

Every rank reads ONLY the bytes of its own time shard (+ the right halo) from the input file, straight into
page-locked memory, and writes ONLY its own slice of every output file at that slice's byte offset (rank 0 writes
the headers first).  Between the two, everything runs on the rank's GPU, and since round 4 the three stages STREAM
(upx_wav_shard_open / _feed / _seal, _finish_async / _wait_piece): a few threads read 4 M-frame pieces of the shard,
each piece is fed to the device as soon as it is there (upload, and behind it the decode and the kernels of every
chunk whose input is complete), the overlap-add seam crosses RCCL, the shard's peaks come out of the same stream; the
two scalars cross the process group (upmix_amd.rendezvous: standard-library sockets - no torch in this process); then
scale, export layout (main.py:110-157) and quantisation run piece by piece on the device, each piece of final
2-channel sample data comes down while the next one is exported, and is written to the files the moment it has landed.  `--host-export`, files with more than two channels and sample formats the device codec does
not read take the NumPy flow of round 2 (decode, scale and export on the host, float64 like main.py).
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Optional

import numpy as np

from . import export, sharding, wav
from .plan import WINDOW_FUNCS
from .rendezvous import Rendezvous

_SUBTYPE_KIND = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32, "FLOAT": 1032}
PIECE_FRAMES = 1 << 22      # frames per piece of the streamed file -> GPU -> file flow (25 MB at 24 bit stereo)
IO_THREADS = 4              # file reads / writes in flight (os.preadv / os.pwrite release the GIL)


def _pwrite_all(fd: int, payload, offset: int) -> None:
    view, done = memoryview(payload).cast("B"), 0
    while done < len(view):
        done += os.pwrite(fd, view[done:], offset + done)


class _Solo:
    """The process group of a single rank."""
    rank, world = 0, 1

    def allreduce_max(self, values):
        return [float(v) for v in values]

    def broadcast_bytes(self, payload, src=0):
        return payload

    def barrier(self):
        pass

    def all_ok(self, ok=True, message=""):
        if not ok:
            raise RuntimeError(message)


def _vote(group, exc: Optional[BaseException]) -> None:
    """
    Collective: every rank says whether it got this far (`exc` = what stopped it, or None).  If any rank failed, EVERY rank
    raises - the one that failed its own exception, the others a RendezvousError that names it - so nobody walks into the
    next collective (RCCL's blocking init, the seam all-reduce inside wav_shard_seal) to wait for a peer that will not come.
    """
    try:
        group.all_ok(exc is None, "" if exc is None else f"{type(exc).__name__}: {exc}")
    except Exception:
        if exc is not None:
            raise exc from None
        raise
    if exc is not None:      # (a group whose all_ok does not raise)
        raise exc


def _global_scale(peak_in: float, peak_out: float):
    """main.py:53-55 and :85-90 on the maxima over all ranks -> (peak_in, overall_peak, scale_factor float64)."""
    if peak_in <= 0.0:                       # (a NaN peak stays NaN, as in main.py)
        peak_in = 1e-9
    overall = peak_out if math.isnan(peak_out) else max(peak_out, 1e-9)
    return peak_in, overall, np.float64(peak_in / overall)


def run_rank(in_path: str, out_dir: str, export_mode: str, bands, overlap: float, subtype: str, rank: int, world: int,
             group=None, engine: Optional[Callable] = None, device: int = 0, log=print, host_export: bool = False,
             times: Optional[dict] = None, plan_factory: Optional[Callable] = None, seam_factory: Optional[Callable] = None):
    """
    One rank's part of the job; returns {key: path} of the files (written by all ranks together).
    `group`: the process group (rendezvous.Rendezvous or anything with allreduce_max / broadcast_bytes / barrier /
    all_ok); None for a single rank.  `engine(local_stereo, shard, geo) -> (center, left, right)` replaces the GPU (the
    CPU tests plug the oracle + their own seam in) and implies the host flow.  `times`: filled with seconds per phase.
    `plan_factory(bands, device)` / `seam_factory(plan, rank, world, broadcast=, all_ok=)`: stand-ins for DevicePlan and
    sharding.RcclSeam (the CPU test of the failure containment: a rank that raises between open and seal).

    Failure containment: the ranks vote (`_vote`) after plan creation (before RCCL's blocking init), inside RcclSeam around
    that init, and between the last feed and wav_shard_seal (which contains the seam all-reduce); a rank that fails in
    between raises on EVERY rank instead of leaving its peers inside a collective.  What no vote can catch - a peer that
    dies inside the collective - ends in upx_comm_wait's timeout and ncclCommAbort.  The process then exits non-zero.
    """
    group = group if group is not None else _Solo()
    t_mark = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        if times is not None:
            times[name] = times.get(name, 0.0) + now - t_mark[0]
        t_mark[0] = now

    # ---- everything that can be refused is checked on EVERY rank before any GPU work or barrier ----------------
    meta = wav.info(in_path)
    total, sr, channels = meta["n_frames"], meta["rate"], meta["channels"]
    wav.subtype_layout(subtype)                                            # ValueError for an unknown subtype
    geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
    shard = geo.plan(total, world)[rank]
    names = export.export_file_names(os.path.splitext(os.path.basename(in_path))[0], export_mode, bands, overlap)
    if not names and rank == 0:
        log(f"Unknown export_mode '{export_mode}' -- no files written.")     # main.py:159-160
    try:
        kind = wav.device_kind(meta, in_path)
    except ValueError:
        kind = None
    on_device = engine is None and not host_export and kind is not None and channels in (1, 2) and bool(names)

    plan = seam = None
    try:
        if engine is None:
            failure = None
            try:
                if plan_factory is None:
                    from .extractor import DevicePlan
                    plan_factory = DevicePlan
                plan = plan_factory(bands, device)
            except Exception as exc:   # noqa: BLE001 - voted on, then raised
                failure = exc
            _vote(group, failure)                      # every rank has its plan before anybody enters RCCL's blocking init
            if world > 1:
                seam = (seam_factory or sharding.RcclSeam)(plan, rank, world, broadcast=group.broadcast_bytes,
                                                           all_ok=group.all_ok)
        spill = geo.spill if world > 1 else 0
        pieces = None
        if on_device:
            # raw bytes of the shard -> page-locked memory -> GPU, piece by piece: a piece is fed to the device (upload,
            # and the kernels of every chunk it completes) while the next ones are still being read from the file;
            # nothing is decoded on the host
            block = meta["bits"] // 8 * channels
            t_out = shard.own_len + (0 if shard.last else spill)
            failure = None
            try:
                raw = plan.host_empty(shard.t_in * block)
                plan.wav_shard_open(kind, channels, shard.t_in, shard.own_len, t_out, spill, seam)
                with ThreadPoolExecutor(max_workers=IO_THREADS) as pool:
                    futs = [pool.submit(wav.read_raw_range, in_path, shard.start + a, min(PIECE_FRAMES, shard.t_in - a), meta,
                                        raw[a * block:(a + min(PIECE_FRAMES, shard.t_in - a)) * block])
                            for a in range(0, shard.t_in, PIECE_FRAMES)]
                    for a, fut in zip(range(0, shard.t_in, PIECE_FRAMES), futs):
                        n = min(PIECE_FRAMES, shard.t_in - a)
                        if fut.result().size != n * block:
                            raise ValueError(f"{in_path}: file ends inside the sample data")
                        plan.wav_shard_feed(raw[a * block:(a + n) * block], n)          # in file order
            except Exception as exc:   # noqa: BLE001 - an I/O error in a reader thread, UPX_ERR_NOMEM ...: voted on, then raised
                failure = exc
            lap("read_s")
            # seal contains the seam all-reduce (upx_comm_seam_exchange): nobody enters it unless every rank has fed its shard
            _vote(group, failure)
            peaks = plan.wav_shard_seal()
            lap("device_begin_s")
            peak_in, overall_peak, scale_factor = _global_scale(*group.allreduce_max(peaks))
            # everything of the second half is queued; piece k of the payloads is complete after wav_shard_wait_piece(k)
            payloads, n_pieces, per = plan.wav_shard_finish_async(float(scale_factor), export_mode, _SUBTYPE_KIND[subtype],
                                                                  shard.own_len, PIECE_FRAMES)
            pieces = (n_pieces, per)
            lap("device_finish_s")
        else:
            failure = None
            try:
                local = wav.read_range(in_path, shard.start, shard.t_in, meta)     # own range + right halo, nothing else
                if local.ndim == 1:
                    local = np.column_stack([local, local])                        # main.py:47-48
                own = local[:shard.own_len]
                stereo = np.ascontiguousarray(local[:, :2])                        # main.py:49-50: wave[:,0], wave[:,1]
            except Exception as exc:   # noqa: BLE001
                failure = exc
            lap("read_s")
            _vote(group, failure)          # the engines below end in the seam all-reduce
            if engine is None:
                c, l, r = sharding.process_local_shard(plan, stereo, shard, geo, world, seam)
            else:
                c, l, r = engine(stereo, shard, geo)
            lap("device_begin_s")
            # one global scale (main.py:53-55, :85-97): maxima over the ranks; the input peak is over ALL channels
            mine = [float(np.max(np.abs(own), initial=0.0)),
                    max(float(np.max(np.abs(l), initial=0.0)), float(np.max(np.abs(c), initial=0.0)),
                        float(np.max(np.abs(r), initial=0.0)), 0.0)]
            peak_in, overall_peak, scale_factor = _global_scale(*group.allreduce_max(mine))
            # main.py:90-97: scale_factor is a float64 NumPy scalar there, so `final_x *= scale_factor` multiplies in
            # float64 and rounds once to float32
            for p in (c, l, r):
                p *= scale_factor
            arrays = export.export_arrays(export_mode, c, l, r, own[:, 0], own[:, 1])
            payloads = {key: np.frombuffer(wav.encode(arr, subtype)[2], dtype=np.uint8) for key, arr in arrays.items()}
            lap("device_finish_s")

        # ---- rank 0 creates the files; its outcome reaches every rank before anyone waits or writes -------------
        err = ""
        if rank == 0:
            try:
                log(f"Original peak = {peak_in:.4f}, L/C/R peak = {overall_peak:.4f}")
                log(f"Applying scale_factor = {scale_factor:.4f}")
                os.makedirs(out_dir, exist_ok=True)
                for key in payloads:
                    wav.create(os.path.join(out_dir, names[key]), total, sr, subtype, 2)   # header + full (sparse) size
            except Exception as exc:   # noqa: BLE001 - reported to every rank, raised by all_ok
                err = f"{type(exc).__name__}: {exc}"
        group.all_ok(not err, err)
        written = {}
        _, bits = wav.subtype_layout(subtype)
        frame = 2 * bits // 8
        if pieces is None:
            for key, payload in payloads.items():
                path = os.path.join(out_dir, names[key])
                wav.write_at(path, wav.info(path)["data_offset"] + shard.start * frame, payload)
                written[key] = path
        else:
            # a piece is written (a few threads: os.pwrite releases the GIL) as soon as it has come down, while the
            # later pieces are still being exported and downloaded
            n_pieces, per = pieces
            offsets = {key: wav.info(os.path.join(out_dir, names[key]))["data_offset"] + shard.start * frame for key in payloads}
            fds = {key: os.open(os.path.join(out_dir, names[key]), os.O_WRONLY) for key in payloads}
            try:
                with ThreadPoolExecutor(max_workers=IO_THREADS) as pool:
                    jobs = []
                    for k in range(n_pieces):
                        plan.wav_shard_wait_piece(k)
                        a, b = k * per * frame, min(shard.own_len, (k + 1) * per) * frame
                        for key, payload in payloads.items():
                            jobs.append(pool.submit(_pwrite_all, fds[key], payload[a:b], offsets[key] + a))
                    for j in jobs:
                        j.result()
            finally:
                for fd in fds.values():
                    os.close(fd)
            written = {key: os.path.join(out_dir, names[key]) for key in payloads}
        lap("write_s")
        group.barrier()
        if rank == 0:
            for key, path in written.items():
                log(f"Wrote => {path}")
            log("Done.")
        return written
    finally:
        if seam is not None:
            seam.close()
        if plan is not None:
            plan.close()


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
    ap.add_argument("--host-export", action="store_true", help="decode / scale / export with NumPy on the host")
    ap.add_argument("--timing", action="store_true", help="print this rank's seconds per phase")
    a = ap.parse_args(argv)

    from .extractor import chain_bands
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    in_path = os.path.join(a.in_dir, a.in_filename)
    if not os.path.isfile(in_path):
        raise FileNotFoundError(f"File not found: {in_path}")              # main.py:40-41
    with Rendezvous.from_env() as group:
        rank, world = group.rank, group.world
        sr = wav.info(in_path)["rate"]
        bands = chain_bands([float(v) for v in a.band_edges.split(",")], a.overlap, WINDOW_FUNCS[a.window], sr,
                            a.xover_mode, max_block_size=a.max_stft, device=local_rank, verbose=rank == 0)
        times = {}
        run_rank(in_path, a.out_dir, a.export_mode, bands, a.overlap, a.subtype, rank, world,
                 group if world > 1 else None, device=local_rank, log=print if rank == 0 else (lambda *_: None),
                 host_export=a.host_export, times=times)
        if a.timing:
            print(f"[rank {rank}] " + ", ".join(f"{k} {v:.3f}" for k, v in times.items()), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
