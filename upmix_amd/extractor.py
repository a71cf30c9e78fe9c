"""
The reference's call surface for the hot path, backed by the gfx950 kernels.

    chain_bands(band_edges, overlap, window_func, sr, xover_mode="raised_cosine")
        -> list[MultiBandExtractorAccu]                    center_extraction.py:518-580
    extract_center_left_right_multi_band_in_memory(L, R, sr, band_extractors)
        -> (center, left, right)  float32[T]               center_extraction.py:477-513
    MultiBandExtractorAccu(...).process_all_blocks(L, R)
        -> (c, l, r)                                       center_extraction.py:426-472

Same names, argument meaning, return order and error behaviour; extra
keyword-only arguments (``max_block_size``, ``threshold_factor``,
``xo_fraction``, ``device``) expose knobs the reference hard-codes.  The
arithmetic runs in libupmix_hip.so (float32 FFT; the reference uses float64 FFT
and float32 overlap-add) - parity is 1e-5 RMS, measured ~1e-8.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import os
import threading
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, hostmem
from .plan import (band_limit_gain, compute_block_size_for_low_freq, design_wola_synthesis_window,
                   hp_freq_to_crossover_width)


def _f32p(a: np.ndarray):
    return a.ctypes.data_as(_lib.f32p)


_LIVE_PLANS = [0]
_LIVE_PLANS_LOCK = threading.Lock()


class DevicePlan:
    """Owns a upx_plan: device copies of windows / gains / twiddles for a list of bands."""

    def __init__(self, extractors: Sequence["MultiBandExtractorAccu"], device: int = 0):
        if len(extractors) < 1:
            raise ValueError("at least one band is required")
        self._lib = _lib.load()
        self.n_bands = len(extractors)
        self.block_sizes = [int(b.block_size) for b in extractors]
        self.hops = [int(b.hop_size) for b in extractors]
        blocks = np.asarray(self.block_sizes, dtype=np.int32)
        hops = np.asarray(self.hops, dtype=np.int32)
        wa = np.ascontiguousarray(np.concatenate([np.asarray(b.analysis_window, dtype=np.float32) for b in extractors]))
        ws = np.ascontiguousarray(np.concatenate([np.asarray(b.synthesis_window, dtype=np.float32) for b in extractors]))
        gains = np.ascontiguousarray(np.concatenate([b.gain_vector().astype(np.float32) for b in extractors]))
        handle = C.c_void_p()
        _lib.check(self._lib.upx_plan_create(C.byref(handle), int(device), self.n_bands,
                                             blocks.ctypes.data_as(_lib.i32p), hops.ctypes.data_as(_lib.i32p),
                                             _f32p(wa), _f32p(ws), _f32p(gains)))
        self.handle = handle
        self.device = int(device)
        # a upx_plan is not thread-safe (include/upmix_hip.h); the reference's caller is a ThreadPoolExecutor
        # (center_extraction.py:499-501), so calls on ONE plan are serialised here; distinct plans run concurrently
        self.lock = threading.RLock()
        self._users = 0   # checked out of the plan cache (see _checked_out_plan)
        with _LIVE_PLANS_LOCK:
            _LIVE_PLANS[0] += 1

    # -- whole signal, host buffers -----------------------------------------
    def process(self, stereo: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """stereo float32 [T,2] -> (center, left, right) float32[T]."""
        x = np.ascontiguousarray(stereo, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("stereo must have shape [T, 2]")
        total = x.shape[0]
        out = self._result_planes(total)
        if total == 0:
            return tuple(out)
        # any length: the library streams long signals through the device in chunks (upx_process_chunked)
        with self.lock:
            _lib.check(self._lib.upx_process(self.handle, _f32p(x), total, *(_f32p(o) for o in out)))
        return tuple(out)

    def _result_planes(self, total: int):
        """Three float32[total] result arrays: pooled page-locked memory from the second call on that asks for this size,
        plain NumPy arrays on the first (hostmem.PinnedPool.take: a one-shot process must not pay for pinning)."""
        token = hostmem.POOL.new_call()
        return [hostmem.empty(total, np.float32, self.handle, lazy=token) for _ in range(3)]

    def process_lr(self, L, R) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """
        (L, R) as the reference's callers hold them -> (center, left, right) float32[T]: center_extraction.py:477-482 takes
        any two real arrays, and main.py:49-50, 78-80 passes two float64 COLUMN VIEWS of one [T, 2] array.  float32 /
        float64 arrays that are such a pair of columns, or contiguous each, go to the library as they are (upx_process_lr:
        cast and interleave on the device, bit-identical to the host cast); anything else (other dtypes, other strides,
        unequal dtypes) is cast and interleaved here first, as before.
        """
        a, b = np.asarray(L), np.asarray(R)
        if a.ndim != 1 or b.ndim != 1:
            raise ValueError("L and R must be one-dimensional")
        if a.shape[0] != b.shape[0]:
            # (the reference pads each to whole frames on its own and fails in the band sum; say so up front)
            raise ValueError(f"L and R differ in length ({a.shape[0]} vs {b.shape[0]})")
        total = a.shape[0]
        fmt = {np.dtype(np.float32): _lib.SAMPLE_F32, np.dtype(np.float64): _lib.SAMPLE_F64}.get(a.dtype)
        stride = 0
        if fmt is not None and b.dtype == a.dtype and a.dtype.isnative and total > 0:
            size = a.dtype.itemsize
            if a.strides[0] == 2 * size and b.strides[0] == 2 * size and b.ctypes.data == a.ctypes.data + size:
                stride = 2
            elif (a.strides[0] == size or total == 1) and (b.strides[0] == size or total == 1):
                stride = 1
        if stride == 0:
            return self.process(np.stack([np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)], axis=1))
        out = self._result_planes(total)
        with self.lock:
            _lib.check(self._lib.upx_process_lr(self.handle, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), fmt,
                                                stride, total, *(_f32p(o) for o in out)))
        return tuple(out)

    def process_tracks(self, tracks: Sequence[np.ndarray]) -> List[Tuple[np.ndarray, np.ndarray, np.ndarray]]:
        """
        A batch of independent tracks (float32 [T_t, 2] each) through this plan: one queue of work items, uploads /
        kernels / downloads of consecutive tracks overlapped (upx_process_tracks).  Every track's result is
        bit-identical to ``process(track)``.  The reference's analogue is one main.py run per file (main.py:36-80).
        """
        xs = [np.ascontiguousarray(t, dtype=np.float32) for t in tracks]
        for x in xs:
            if x.ndim != 2 or x.shape[1] != 2:
                raise ValueError("every track must have shape [T, 2]")
        n = len(xs)
        token = hostmem.POOL.new_call()
        outs = [[hostmem.empty(x.shape[0], np.float32, self.handle, lazy=token) for _ in range(3)] for x in xs]
        if n == 0:
            return []
        lens = (C.c_int64 * n)(*[x.shape[0] for x in xs])
        ins = (C.c_void_p * n)(*[x.ctypes.data for x in xs])
        planes = [(C.c_void_p * n)(*[o[k].ctypes.data for o in outs]) for k in range(3)]
        with self.lock:
            _lib.check(self._lib.upx_process_tracks(self.handle, n, ins, lens, planes[0], planes[1], planes[2]))
        return [tuple(o) for o in outs]

    def process_chunked(self, stereo: np.ndarray, chunk: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """The same through the streaming pipeline with an explicit chunk length (samples)."""
        x = np.ascontiguousarray(stereo, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != 2:
            raise ValueError("stereo must have shape [T, 2]")
        total = x.shape[0]
        out = self._result_planes(total)
        if total:
            with self.lock:
                _lib.check(self._lib.upx_process_chunked(self.handle, _f32p(x), total, *(_f32p(o) for o in out), int(chunk)))
        return tuple(out)

    def host_empty(self, shape, dtype=np.uint8) -> np.ndarray:
        """np.empty(shape, dtype) in pooled page-locked host memory (falls back to pageable memory beyond the pool's
        limit): the arrays this plan's host-buffer calls return, and the staging buffers of the file entries."""
        return hostmem.empty(shape, dtype, self.handle)

    # -- device-resident helpers --------------------------------------------
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _lib.check(self._lib.upx_dev_alloc(self.handle, C.byref(p), int(nbytes)))
        return p.value

    def free(self, ptr: int) -> None:
        _lib.check(self._lib.upx_dev_free(self.handle, C.c_void_p(ptr)))

    def memset(self, ptr: int, value: int, nbytes: int) -> None:
        _lib.check(self._lib.upx_dev_memset(self.handle, C.c_void_p(ptr), int(value), int(nbytes)))

    def h2d(self, dst: int, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src)
        _lib.check(self._lib.upx_copy_h2d(self.handle, C.c_void_p(dst), src.ctypes.data_as(C.c_void_p), src.nbytes))

    def d2h(self, dst: np.ndarray, src: int) -> None:
        assert dst.flags["C_CONTIGUOUS"]
        _lib.check(self._lib.upx_copy_d2h(self.handle, dst.ctypes.data_as(C.c_void_p), C.c_void_p(src), dst.nbytes))

    def sync(self) -> None:
        _lib.check(self._lib.upx_sync(self.handle))

    def process_device(self, d_in: int, t_in: int, own_len: int, d_c: int, d_l: int, d_r: int, t_out: int) -> None:
        _lib.check(self._lib.upx_process_device(self.handle, C.c_void_p(d_in), int(t_in), int(own_len),
                                                C.c_void_p(d_c), C.c_void_p(d_l), C.c_void_p(d_r), int(t_out)))

    def reserve(self, t_in: int, own_len: int, t_out: int) -> None:
        """Everything the first process_device call of this shape would allocate / upload / synchronise for, up front
        (upx_plan_reserve): the call itself then only enqueues kernels."""
        with self.lock:
            _lib.check(self._lib.upx_plan_reserve(self.handle, int(t_in), int(own_len), int(t_out)))

    def enable_timing(self, on: bool = True) -> None:
        _lib.check(self._lib.upx_plan_enable_timing(self.handle, 1 if on else 0))

    def pause_timing(self, paused: bool = True) -> None:
        """Stop / resume recording events without forgetting the calls recorded so far (a loop that times every n-th call)."""
        _lib.check(self._lib.upx_plan_enable_timing(self.handle, 3 if paused else 2))

    def band_times_ms(self) -> np.ndarray:
        ms = np.zeros(self.n_bands, dtype=np.float32)
        _lib.check(self._lib.upx_plan_band_times_ms(self.handle, _f32p(ms), self.n_bands))
        return ms

    def band_times_sum_ms(self, n_calls: int) -> np.ndarray:
        """Per-band kernel time summed over the last n_calls (<= 64) timed process_device calls; one sync."""
        ms = np.zeros(self.n_bands, dtype=np.float32)
        _lib.check(self._lib.upx_plan_band_times_sum_ms(self.handle, _f32p(ms), self.n_bands, int(n_calls)))
        return ms

    def band_times_calls_ms(self, n_calls: int) -> np.ndarray:
        """[n_calls, n_bands] kernel time of each of the last n_calls (<= 64) timed calls; one sync."""
        ms = np.zeros((int(n_calls), self.n_bands), dtype=np.float32)
        _lib.check(self._lib.upx_plan_band_times_calls_ms(self.handle, _f32p(ms), self.n_bands, int(n_calls)))
        return ms

    def band_phase_times_sum_ms(self, n_calls: int):
        """(analysis, synthesis) time per band summed over the last n_calls timed calls (two-kernel bands split;
        single-kernel bands: (0, whole time))."""
        a = np.zeros(self.n_bands, dtype=np.float32)
        s = np.zeros(self.n_bands, dtype=np.float32)
        _lib.check(self._lib.upx_plan_band_phase_times_sum_ms(self.handle, _f32p(a), _f32p(s), self.n_bands, int(n_calls)))
        return a, s

    def band_phase_kernel_name(self, band: int, phase: int) -> str:
        buf = C.create_string_buffer(160)
        _lib.check(self._lib.upx_plan_band_phase_kernel_name(self.handle, int(band), int(phase), buf, len(buf)))
        return buf.value.decode()

    def band_stream_starts(self, band: int) -> np.ndarray:
        """First frame of every stream of `band`'s launch group in the last process_device call, then the end frame
        (band-limited groups: the Ls/Rs table, then the centre table): upx_plan_band_stream_starts."""
        n = C.c_int32()
        _lib.check(self._lib.upx_plan_band_stream_starts(self.handle, int(band), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.int32)
        _lib.check(self._lib.upx_plan_band_stream_starts(self.handle, int(band), out.ctypes.data_as(_lib.i32p), n.value,
                                                         C.byref(n)))
        return out[:n.value]

    def band_info(self, band: int) -> dict:
        v = [C.c_int32() for _ in range(4)]
        _lib.check(self._lib.upx_plan_band_info(self.handle, int(band), *(C.byref(i) for i in v)))
        return {"workgroups": v[0].value, "threads": v[1].value, "lds_bytes": v[2].value,
                "blocks_per_stream": v[3].value}

    def band_fill(self, band: int) -> dict:
        """Workgroups of the last call's launch of `band` against the workgroup slots the chip holds at once (main kernel /
        band-limited analysis): upx_plan_band_fill."""
        v = [C.c_int32() for _ in range(4)]
        _lib.check(self._lib.upx_plan_band_fill(self.handle, int(band), *(C.byref(i) for i in v)))
        return {"workgroups": v[0].value, "slots": v[1].value, "workgroups_analysis": v[2].value,
                "slots_analysis": v[3].value}

    def band_kernel_name(self, band: int) -> str:
        """Kernel symbol (as rocprofv3 prints it) of the launch that carries `band`."""
        buf = C.create_string_buffer(160)
        _lib.check(self._lib.upx_plan_band_kernel_name(self.handle, int(band), buf, len(buf)))
        return buf.value.decode()

    def band_group(self, band: int):
        """(leader, size) of the merged launch that carries `band`."""
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self._lib.upx_plan_band_group(self.handle, int(band), C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_blocks_per_stream(self, blocks: int, band: int = -1) -> None:
        _lib.check(self._lib.upx_plan_set_blocks_per_stream(self.handle, int(band), int(blocks)))

    def absmax(self, ptr: int, n: int) -> float:
        r = C.c_float()
        _lib.check(self._lib.upx_absmax(self.handle, C.c_void_p(ptr), int(n), C.byref(r)))
        return float(r.value)

    def scale(self, ptr: int, n: int, factor: float) -> None:
        _lib.check(self._lib.upx_scale(self.handle, C.c_void_p(ptr), int(n), float(factor)))

    def wav_pipeline(self, pcm: np.ndarray, in_format: int, channels: int, n_frames: int, mode: str,
                     out_format: int = 16):
        """
        Raw PCM in -> final 2-channel sample data out, everything between on the device
        (decode, all bands, global peak scale, export layout, quantisation).  Returns
        ({"Sum"|"AB"|"Ls","C","Rs": uint8 payload}, {"peak_in", "overall_peak", "scale_factor"}).
        main.py:43-160; `mode` in ("stereo_sum", "split", "AB").
        """
        if mode not in self._MODES:
            raise ValueError(f"unknown export mode {mode!r}")
        code, names = self._MODES[mode]
        width = 4 if out_format == _lib.F32 else out_format // 8
        src = np.ascontiguousarray(pcm).view(np.uint8)
        outs = [self.host_empty(n_frames * 2 * width) for _ in names]
        ptrs = [o.ctypes.data_as(C.c_void_p) for o in outs] + [None] * (3 - len(outs))
        stats = (C.c_double * 3)()
        _lib.check(self._lib.upx_wav_pipeline(self.handle, src.ctypes.data_as(C.c_void_p), int(in_format), int(channels),
                                              int(n_frames), code, int(out_format), ptrs[0], ptrs[1], ptrs[2], stats))
        return dict(zip(names, outs)), {"peak_in": stats[0], "overall_peak": stats[1], "scale_factor": stats[2]}

    _MODES = {"stereo_sum": (_lib.EXPORT_STEREO_SUM, ("Sum",)), "split": (_lib.EXPORT_SPLIT, ("Ls", "C", "Rs")),
              "AB": (_lib.EXPORT_AB, ("AB",))}

    def wav_shard_begin(self, pcm: np.ndarray, in_format: int, channels: int, t_in: int, own_len: int, t_out: int,
                        spill: int = 0, seam=None):
        """
        First half of the WAV pipeline for one time shard (upx_wav_shard_begin): raw samples up, decode, all bands, the
        RCCL seam (`seam` = sharding.RcclSeam or None), peaks of the owned range.  -> (peak_in, peak_out) as the device
        found them (0 for silence, NaN if a NaN was seen); the caller reduces them over the ranks.
        """
        src = np.ascontiguousarray(pcm).view(np.uint8)
        peaks = (C.c_double * 2)()
        with self.lock:
            _lib.check(self._lib.upx_wav_shard_begin(self.handle, seam.handle if seam is not None else None,
                                                     src.ctypes.data_as(C.c_void_p), int(in_format), int(channels),
                                                     int(t_in), int(own_len), int(t_out), int(spill), peaks))
        return float(peaks[0]), float(peaks[1])

    # -- the same two halves, streamed (file -> GPU -> file: multi_gpu.run_rank) --------------------------------
    def wav_shard_open(self, in_format: int, channels: int, t_in: int, own_len: int, t_out: int, spill: int = 0, seam=None):
        with self.lock:
            _lib.check(self._lib.upx_wav_shard_open(self.handle, seam.handle if seam is not None else None, int(in_format),
                                                    int(channels), int(t_in), int(own_len), int(t_out), int(spill)))

    def wav_shard_feed(self, pcm: np.ndarray, n_frames: int) -> None:
        """The next n_frames of the shard's raw samples (uint8 view; must stay alive and unchanged until wav_shard_seal)."""
        with self.lock:
            _lib.check(self._lib.upx_wav_shard_feed(self.handle, pcm.ctypes.data_as(C.c_void_p), int(n_frames)))

    def wav_shard_seal(self):
        peaks = (C.c_double * 2)()
        with self.lock:
            _lib.check(self._lib.upx_wav_shard_seal(self.handle, peaks))
        return float(peaks[0]), float(peaks[1])

    def wav_shard_finish_async(self, scale: float, mode: str, out_format: int, n_frames: int, piece_frames: int = 1 << 22):
        """Queues scale + export + quantisation + download piece by piece; -> ({name: uint8 payload}, n_pieces,
        frames per piece).  Piece k of every payload is valid after wav_shard_wait_piece(k)."""
        if mode not in self._MODES:
            raise ValueError(f"unknown export mode {mode!r}")
        code, names = self._MODES[mode]
        width = 4 if out_format == _lib.F32 else out_format // 8
        outs = [self.host_empty(n_frames * 2 * width) for _ in names]
        ptrs = [o.ctypes.data_as(C.c_void_p) for o in outs] + [None] * (3 - len(outs))
        n_pieces = C.c_int32()
        with self.lock:
            _lib.check(self._lib.upx_wav_shard_finish_async(self.handle, float(scale), code, int(out_format), ptrs[0], ptrs[1],
                                                            ptrs[2], max(1, int(piece_frames)), C.byref(n_pieces)))
        per = max(1, min(int(piece_frames), n_frames))
        assert n_pieces.value == -(-n_frames // per)
        return dict(zip(names, outs)), n_pieces.value, per

    def wav_shard_wait_piece(self, piece: int) -> None:
        _lib.check(self._lib.upx_wav_shard_wait_piece(self.handle, int(piece)))

    def wav_shard_planes(self):
        """Device pointers (center, left, right) of the open shard's planes, for a seam applied by the caller."""
        ptrs = [C.c_void_p() for _ in range(3)]
        _lib.check(self._lib.upx_wav_shard_planes(self.handle, *(C.byref(q) for q in ptrs), None, None))
        return [q.value for q in ptrs]

    def wav_shard_peaks(self):
        """(peak_in, peak_out) of the open shard's owned range, recomputed (after a caller-applied seam)."""
        peaks = (C.c_double * 2)()
        _lib.check(self._lib.upx_wav_shard_peaks(self.handle, peaks))
        return float(peaks[0]), float(peaks[1])

    def wav_shard_finish(self, scale: float, mode: str, out_format: int, n_frames: int):
        """Second half: scale, export layout, quantisation on the device; -> {name: uint8 payload of n_frames frames}."""
        if mode not in self._MODES:
            raise ValueError(f"unknown export mode {mode!r}")
        code, names = self._MODES[mode]
        width = 4 if out_format == _lib.F32 else out_format // 8
        outs = [self.host_empty(n_frames * 2 * width) for _ in names]
        ptrs = [o.ctypes.data_as(C.c_void_p) for o in outs] + [None] * (3 - len(outs))
        with self.lock:
            _lib.check(self._lib.upx_wav_shard_finish(self.handle, float(scale), code, int(out_format), ptrs[0], ptrs[1],
                                                      ptrs[2]))
        return dict(zip(names, outs))

    def wav_pipeline_times_ms(self):
        ms = np.zeros(3, dtype=np.float32)
        _lib.check(self._lib.upx_wav_pipeline_times_ms(self.handle, _f32p(ms)))
        # begin = upload || decode || bands || peaks; begin_tail = its part after the last sample landed; finish = export || download
        return {"begin": float(ms[0]), "begin_tail": float(ms[1]), "finish": float(ms[2])}

    def stream_state(self, clear: bool = False):
        """(accumC, accumL, accumR) of the one-band streaming ring in natural order (upx_stream_state); clear = flush."""
        n = self.block_sizes[0]
        acc = [np.empty(n, dtype=np.float32) for _ in range(3)]
        _lib.check(self._lib.upx_stream_state(self.handle, *(_f32p(a) for a in acc), 1 if clear else 0))
        return tuple(acc)

    def stream_set_state(self, acc_c, acc_l, acc_r) -> None:
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in (acc_c, acc_l, acc_r)]
        _lib.check(self._lib.upx_stream_set_state(self.handle, *(_f32p(a) for a in arrs)))

    def seam_add_local(self, prev: Sequence[int], prev_own_len: int, nxt: Sequence[int], spill: int) -> None:
        _lib.check(self._lib.upx_seam_add_local(self.handle, *(C.c_void_p(p) for p in prev), int(prev_own_len),
                                                *(C.c_void_p(p) for p in nxt), int(spill)))

    def close(self) -> None:
        if getattr(self, "handle", None) is not None and self.handle:
            with self.lock:
                with _LIVE_PLANS_LOCK:
                    _LIVE_PLANS[0] -= 1
                    last = _LIVE_PLANS[0] == 0
                if last:
                    hostmem.POOL.trim(self.handle)   # idle page-locked blocks go back while a device context exists
                self._lib.upx_plan_destroy(self.handle)
                self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_kernel_names(extractors: Sequence["MultiBandExtractorAccu"]) -> List[str]:
    """
    The kernel(s) the library would select for every band of this list ("analysis|synthesis" for a band-limited group),
    without a GPU (upx_plan_kernel_names: the selection half of upx_plan_create).
    """
    lib = _lib.load()
    blocks = np.asarray([int(b.block_size) for b in extractors], dtype=np.int32)
    hops = np.asarray([int(b.hop_size) for b in extractors], dtype=np.int32)
    wa = np.ascontiguousarray(np.concatenate([np.asarray(b.analysis_window, dtype=np.float32) for b in extractors]))
    ws = np.ascontiguousarray(np.concatenate([np.asarray(b.synthesis_window, dtype=np.float32) for b in extractors]))
    gains = np.ascontiguousarray(np.concatenate([b.gain_vector().astype(np.float32) for b in extractors]))
    buf = C.create_string_buffer(256 * max(1, len(extractors)))
    _lib.check(lib.upx_plan_kernel_names(len(extractors), blocks.ctypes.data_as(_lib.i32p), hops.ctypes.data_as(_lib.i32p),
                                         _f32p(wa), _f32p(ws), _f32p(gains), buf, len(buf)))
    return buf.value.decode().splitlines()


class MultiBandExtractorAccu:
    """
    Per-band extractor with the constructor and attributes of the reference class
    (center_extraction.py:217-271).  The per-frame work (rfft x2, band limit,
    mask, irfft x3, overlap-add; :353-409) runs on the GPU.
    """

    def __init__(self, block_size: int, overlap: float, window_func: Callable[[int], np.ndarray], f_low: float,
                 f_high: float, sr: float, xover_mode: str = "hard_zero", xover_width_low_hz: float = 50.0,
                 xover_width_high_hz: float = 50.0, *, device: int = 0):
        self.block_size = block_size
        self.overlap = overlap
        self.hop_size = int(block_size * (1 - overlap))
        if self.hop_size < 1:
            raise ValueError("Overlap too large; hop size < 1 is not allowed.")
        self.analysis_window = window_func(block_size)
        self.synthesis_window = design_wola_synthesis_window(self.analysis_window, overlap)
        self.sr = sr
        self.f_low = f_low
        self.f_high = f_high
        self.xover_mode = xover_mode
        self.xover_width_low_hz = xover_width_low_hz
        self.xover_width_high_hz = xover_width_high_hz
        self.device = device
        # streaming state for process_stereo_chunk / flush_final (:269-271): the three accumulators live on the device
        # once the first block has gone through (upx_stream_chunk); .accumC / .accumL / .accumR read them back
        self._accum = [np.zeros(block_size, dtype=np.float32) for _ in range(3)]
        self._plan: Optional[DevicePlan] = None
        self._streaming = False    # the accumulators are on the device

    # what _band_signature() is made of: assigning any of them drops the memoised signature, so the next drop-in call keys
    # (and, if need be, builds) the plan for the new values
    _SIGNATURE_FIELDS = frozenset(("block_size", "hop_size", "sr", "f_low", "f_high", "xover_mode", "xover_width_low_hz",
                                   "xover_width_high_hz", "analysis_window", "synthesis_window"))

    def __setattr__(self, name, value):
        if name in MultiBandExtractorAccu._SIGNATURE_FIELDS:
            self.__dict__.pop("_signature", None)
            if self.__dict__.get("_plan") is not None:
                self.close()                       # the one-band plan of process_all_blocks was built from the old values
            if name in ("analysis_window", "synthesis_window") and isinstance(value, np.ndarray):
                value = value.view()
                value.flags.writeable = False      # an in-place edit would leave a stale plan behind silently: make it raise
        object.__setattr__(self, name, value)

    def close(self) -> None:
        """Release the device state this extractor owns (its one-band plan with the streaming ring)."""
        plan, self._plan = self._plan, None
        if plan is not None and plan.handle:
            if self._streaming:
                try:
                    self._accum = list(plan.stream_state(clear=False))
                except Exception:
                    pass
            plan.close()
        self._streaming = False

    # The reference's accumulator attributes (:269-271): host arrays until streaming starts, then the overlap-add ring lives
    # on the device.  Reading an attribute then returns a COPY read back from the device (one synchronisation + 3 N floats
    # down), marked read-only: the reference's in-place idiom `ext.accumC[:] = 0` (:402-424) would otherwise write into a
    # temporary and be silently lost - here it raises.  Whole-attribute assignment (`ext.accumC = x`, `ext.accumC += x`)
    # goes through the setter and reaches the device.
    def _accum_get(self, k: int) -> np.ndarray:
        plan = self._plan
        if self._streaming and plan is not None:
            with plan.lock:
                arr = plan.stream_state(clear=False)[k]
            arr.flags.writeable = False
            return arr
        return self._accum[k]

    def _accum_set(self, k: int, value) -> None:
        new = np.array(np.broadcast_to(np.asarray(value, dtype=np.float32), (self.block_size,)), dtype=np.float32, copy=True)
        plan = self._plan
        if self._streaming and plan is not None:
            with plan.lock:
                cur = list(plan.stream_state(clear=False))
                cur[k] = new
                plan.stream_set_state(*cur)
        else:
            self._accum[k] = new

    accumC = property(lambda self: self._accum_get(0), lambda self, v: self._accum_set(0, v))
    accumL = property(lambda self: self._accum_get(1), lambda self, v: self._accum_set(1, v))
    accumR = property(lambda self: self._accum_get(2), lambda self, v: self._accum_set(2, v))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gain_vector(self) -> np.ndarray:
        """float64[N/2+1]: what _band_limit does to a spectrum (:334-351)."""
        return band_limit_gain(self.block_size, self.sr, self.f_low, self.f_high, self.xover_mode,
                               self.xover_width_low_hz, self.xover_width_high_hz)

    def _device_plan(self) -> DevicePlan:
        if self._plan is None:
            self._plan = DevicePlan([self], self.device)
        return self._plan

    def process_all_blocks(self, L: np.ndarray, R: np.ndarray) -> tuple:
        """Whole-signal band output (c, l, r), float32, length len(L).  center_extraction.py:426-472"""
        return self._device_plan().process_lr(L, R)

    def process_stereo_chunk(self, blkL: np.ndarray, blkR: np.ndarray) -> tuple:
        """
        One block in, first hop_size samples of the running overlap-add out (center_extraction.py:353-409), in ONE
        library call (upx_stream_chunk): the block goes up (2 N floats), the frame is transformed on the GPU, added onto
        the overlap-add ring that stays on the device, and the emitted hop comes down (3 hop floats).  The float32
        accumulate / emit / shift is the reference's, addition for addition.
        """
        n, hop = self.block_size, self.hop_size
        plan = self._device_plan()
        l = np.ascontiguousarray(blkL, dtype=np.float32)
        r = np.ascontiguousarray(blkR, dtype=np.float32)
        if l.ndim != 1 or r.ndim != 1 or len(l) > n or len(r) > n:
            # the reference fails as well (forward_stft: block * window cannot broadcast, :366-367); a wrong chunker in the
            # caller must not come back as wrong audio
            raise ValueError(f"operands could not be broadcast together with shapes ({len(l)},) ({n},): a block is at most "
                             f"block_size = {n} samples")
        outs = [np.empty(hop, dtype=np.float32) for _ in range(3)]
        with plan.lock:
            if not self._streaming:
                if any(a.any() for a in self._accum):
                    plan.stream_set_state(*self._accum)       # accumulators a caller filled before the first block
                self._streaming = True
            _lib.check(plan._lib.upx_stream_chunk(plan.handle, _f32p(l), len(l), _f32p(r), len(r), *(_f32p(o) for o in outs)))
        return tuple(outs)

    def flush_final(self) -> tuple:
        """Remaining overlap-add tail; resets the accumulators.  center_extraction.py:411-424"""
        if self._streaming and self._plan is not None:
            with self._plan.lock:
                return self._plan.stream_state(clear=True)
        outs = tuple(a.copy() for a in self._accum)
        for a in self._accum:
            a[:] = 0
        return outs


_PLAN_CACHE: dict = {}
_PLAN_CACHE_LOCK = threading.Lock()
_PLAN_CACHE_SIZE = 4


def _band_signature(b: "MultiBandExtractorAccu") -> tuple:
    """
    Everything that determines a band's device state (ids of Python objects can be recycled, values cannot).  Hashing both
    windows costs 0.17 ms (C3 plan) to 0.7 ms (default plan: 1.2 MB) per call, so the signature is kept on the extractor and
    dropped whenever one of the attributes it is made of is assigned (MultiBandExtractorAccu.__setattr__); the window arrays
    are handed out read-only so that an in-place edit cannot go unnoticed - `ext.analysis_window[:] = w` raises, `ext.
    analysis_window = w` (what the reference's own code does, center_extraction.py:255-256) re-keys the plan.
    """
    sig = b.__dict__.get("_signature")
    if sig is None:
        sig = (int(b.block_size), int(b.hop_size), float(b.sr), float(b.f_low), float(b.f_high), str(b.xover_mode),
               float(b.xover_width_low_hz), float(b.xover_width_high_hz),
               hash(np.asarray(b.analysis_window, dtype=np.float32).tobytes()),
               hash(np.asarray(b.synthesis_window, dtype=np.float32).tobytes()))
        b.__dict__["_signature"] = sig
    return sig


def _env_knobs() -> tuple:
    """
    The UPX_* environment as a hashable key (the library reads its tuning knobs when a plan is created: a plan made under
    other settings is another plan).  os.environ keeps the raw bytes in a dict; filtering that costs ~2 us where decoding
    every variable through os.environ.items() cost 15-30.
    """
    data = getattr(os.environ, "_data", None)
    if isinstance(data, dict):
        return tuple(sorted(kv for kv in data.items() if kv[0].startswith(b"UPX_")))
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("UPX_")))


@contextlib.contextmanager
def _checked_out_plan(band_extractors: Sequence[MultiBandExtractorAccu], device: int):
    """
    The cached DevicePlan of a band list (small LRU: plans own device memory), safe to call from several threads:
    look-up, creation and eviction happen under one lock, and a plan that some thread is using is never evicted
    (the cache may exceed its size for that long).
    """
    key = (tuple(_band_signature(b) for b in band_extractors), device, _env_knobs())
    with _PLAN_CACHE_LOCK:
        plan = _PLAN_CACHE.pop(key, None)
        if plan is None:
            for k in [k for k, q in _PLAN_CACHE.items() if q._users == 0][:max(0, len(_PLAN_CACHE) - _PLAN_CACHE_SIZE + 1)]:
                _PLAN_CACHE.pop(k).close()
            plan = DevicePlan(band_extractors, device)
        _PLAN_CACHE[key] = plan   # most recently used last
        plan._users += 1
    try:
        yield plan
    finally:
        with _PLAN_CACHE_LOCK:
            plan._users -= 1


def extract_center_left_right_multi_band_in_memory(L: np.ndarray, R: np.ndarray, sr: float,
                                                   band_extractors: List[MultiBandExtractorAccu], *,
                                                   device: int = 0) -> tuple:
    """
    All bands on one GPU, summed in list order in float32; returns
    (final_center, final_left, final_right).  ``sr`` is accepted and unused, as in
    the reference (center_extraction.py:477-513).  Thread-safe (the reference's own caller
    is a thread pool): distinct band lists run on distinct plans, equal ones take turns.
    """
    # (L, R) go to the library as the caller holds them - main.py:49-50 hands two float64 column views of one [T, 2]
    # array - and are cast / interleaved on the device (DevicePlan.process_lr)
    with _checked_out_plan(band_extractors, device) as plan:
        return plan.process_lr(L, R)


def process_tracks(tracks: Sequence[np.ndarray], band_extractors: List[MultiBandExtractorAccu], *,
                   device: int = 0) -> List[tuple]:
    """
    A batch of independent stereo tracks ([T_t, 2] arrays, any real dtype) through ONE band plan on one GPU
    (BASELINE configs[4]); returns [(center, left, right), ...] in track order, each exactly what
    ``extract_center_left_right_multi_band_in_memory`` returns for that track alone.  The reference runs
    main.py:36-80 once per file; here the tracks share the plan and their transfers overlap the kernels.
    """
    with _checked_out_plan(band_extractors, device) as plan:
        return plan.process_tracks(tracks)


def chain_bands(band_edges: List[float], overlap: float, window_func: Callable[[int], np.ndarray], sr: float,
                xover_mode: str = "raised_cosine", *, max_block_size: int = 2 ** 16, threshold_factor: float = 32,
                xo_fraction: float = 0.25, device: int = 0, verbose: bool = True) -> List[MultiBandExtractorAccu]:
    """
    Consecutive bands [e_i, e_i+1] (+ Nyquist if missing); block size from the low
    edge; each band's low fade width is the previous band's high fade width.
    center_extraction.py:518-580 (prints the same per-band line).
    """
    if band_edges[-1] < (sr / 2.0):
        band_edges = list(band_edges) + [sr / 2.0]
    extractors: List[MultiBandExtractorAccu] = []
    prev_high = 0.0
    for i, (f_low, f_high) in enumerate(zip(band_edges[:-1], band_edges[1:])):
        block_size = compute_block_size_for_low_freq(f_low, sr, max_block_size, threshold_factor)
        xover_low, xover_high = prev_high, hp_freq_to_crossover_width(f_high, xo_fraction)
        if verbose:
            print(f"[Band {i+1}] f_low={f_low:.1f} Hz, f_high={f_high:.1f} Hz, block_size={block_size}, "
                  f"xover_low={xover_low:.1f} Hz, xover_high={xover_high:.1f} Hz")
        extractors.append(MultiBandExtractorAccu(block_size, overlap, window_func, f_low, f_high, sr, xover_mode,
                                                 xover_low, xover_high, device=device))
        prev_high = xover_high
    return extractors
