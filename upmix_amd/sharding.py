"""
Time sharding of the hot path.

Frames are independent given their input window; the only coupling is the
linear overlap-add.  A signal is cut on a grid of hop_max = max_b hop_b (every
smaller hop divides it), shard g owns samples [s_g, s_g+1) and every frame of
every band that STARTS there, so the band sum stays local.  It reads a right
halo of max_b (N_b - hop_b) input samples and its output spills the same number
of samples past s_g+1; that spill is the seam: one all-reduce over
seam[G][3][spill] (RCCL across GPUs), after which shard g+1 adds row g to its
head.  The reference has no counterpart (single process, center_extraction.py:477-513).

The arithmetic here is engine-agnostic: `engine(local_stereo, own_len, t_out)`
returns the three planes of one shard.  The product engine is the HIP library
(DevicePlan); the CPU tests plug the oracle in to check the seam algebra with
world_size-2 gloo.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


@dataclass
class Shard:
    index: int
    start: int        # first owned sample (global)
    own_len: int      # owned samples
    t_in: int         # input samples to hand to the engine, from `start`
    t_out: int        # output plane length: own_len (+ spill unless last)
    last: bool


class ShardGeometry:
    def __init__(self, block_sizes: Sequence[int], hops: Sequence[int]):
        self.block_sizes = [int(n) for n in block_sizes]
        self.hops = [int(h) for h in hops]
        self.hop_max = max(self.hops)
        for h in self.hops:
            if self.hop_max % h:
                raise ValueError(f"hop {h} does not divide hop_max {self.hop_max}: bands cannot share a shard grid")
        # shard boundaries sit on multiples of 2*hop_max so that local frame parity == global frame
        # parity in every band (the kernels transform frames in (odd, even) pairs)
        self.grid = 2 * self.hop_max
        self.spill = max(n - h for n, h in zip(self.block_sizes, self.hops))
        self.halo = self.spill

    def plan(self, total: int, n_shards: int) -> List[Shard]:
        """Cut [0, total) into n_shards contiguous ranges on the 2*hop_max grid (last one takes the remainder)."""
        if n_shards < 1:
            raise ValueError("n_shards must be >= 1")
        cells = -(-total // self.grid)
        if n_shards > 1 and cells // n_shards * self.grid < self.spill:
            raise ValueError(f"signal of {total} samples is too short for {n_shards} shards "
                             f"(each must own at least {self.spill} samples)")
        base, extra = divmod(cells, n_shards)
        shards, start = [], 0
        for g in range(n_shards):
            n_cells = base + (1 if g < extra else 0)
            last = g == n_shards - 1
            end = total if last else start + n_cells * self.grid
            own = end - start
            shards.append(Shard(g, start, own, min(total - start, own + (0 if last else self.halo)),
                                own + (0 if last else self.spill), last))
            start = end
        return shards


def pack_seam(planes: Sequence[np.ndarray], shard: Shard, n_shards: int, spill: int) -> np.ndarray:
    """seam[G][3][spill] with this shard's spill in its own row (zero elsewhere; the last shard has none)."""
    seam = np.zeros((n_shards, 3, spill), dtype=np.float32)
    if not shard.last:
        for p, plane in enumerate(planes):
            seam[shard.index, p, :] = plane[shard.own_len:shard.own_len + spill]
    return seam


def apply_seam(planes: Sequence[np.ndarray], shard: Shard, seam: np.ndarray) -> None:
    """Add the previous shard's spill onto this shard's head (in place)."""
    if shard.index == 0:
        return
    spill = seam.shape[2]
    n = min(spill, len(planes[0]))
    for p, plane in enumerate(planes):
        plane[:n] += seam[shard.index - 1, p, :n]


def run_shard(stereo: np.ndarray, geo: ShardGeometry, shard: Shard, n_shards: int,
              engine: Callable[[np.ndarray, int, int], Tuple[np.ndarray, np.ndarray, np.ndarray]],
              allreduce: Callable[[np.ndarray], np.ndarray]) -> Tuple[np.ndarray, ...]:
    """One rank's work: engine on the local window, seam all-reduce, head fix-up, trim to the owned range."""
    local = stereo[shard.start:shard.start + shard.t_in]
    planes = [np.array(p, dtype=np.float32, copy=True) for p in engine(local, shard.own_len, shard.t_out)]
    if n_shards > 1:
        seam = allreduce(pack_seam(planes, shard, n_shards, geo.spill))
        apply_seam(planes, shard, seam)
    return tuple(p[:shard.own_len] for p in planes)


# ---------------------------------------------------------------------------
# product engines
# ---------------------------------------------------------------------------
def process_sharded_single_device(plan, stereo: np.ndarray, max_shard: int = 1 << 27):
    """
    Whole signal on ONE device in several launches (signals longer than a launch can
    index, or larger than HBM): shards run one after another, the seam is added on
    the device with upx_seam_add_local.
    """
    total = stereo.shape[0]
    geo = ShardGeometry(plan.block_sizes, plan.hops)
    n_shards = max(1, -(-total // max_shard))
    shards = geo.plan(total, n_shards)
    outs = [np.empty(total, dtype=np.float32) for _ in range(3)]
    cap_in = max(s.t_in for s in shards)
    cap_out = max(s.t_out for s in shards)
    d_in = plan.alloc(cap_in * 8)
    bufs = [[plan.alloc(cap_out * 4) for _ in range(3)] for _ in range(2)]
    try:
        prev: Optional[Shard] = None
        for s in shards:
            cur, old = bufs[s.index % 2], bufs[(s.index + 1) % 2]
            plan.h2d(d_in, stereo[s.start:s.start + s.t_in])
            plan.process_device(d_in, s.t_in, s.own_len, cur[0], cur[1], cur[2], s.t_out)
            if prev is not None:
                plan.seam_add_local(old, prev.own_len, cur, min(geo.spill, s.t_out))
            for o, d in zip(outs, cur):
                plan.d2h(o[s.start:s.start + s.own_len], d)
            prev = s
    finally:
        plan.free(d_in)
        for pair in bufs:
            for b in pair:
                plan.free(b)
    return tuple(outs)


def process_local_shard(plan, local: np.ndarray, shard: Shard, geo: ShardGeometry, world: int,
                        seam: Optional["RcclSeam"] = None):
    """
    One rank's GPU work on the samples it has read: `local` = stereo[shard.start : shard.start + shard.t_in]
    (own range + right halo).  Upload, every band, the overlap-add seam over RCCL, download of the owned range;
    returns (center, left, right) float32[shard.own_len].
    """
    if world > 1 and seam is None:
        raise ValueError("a multi-rank run needs an RcclSeam")
    spill = geo.spill if world > 1 else 0
    local = np.ascontiguousarray(local, dtype=np.float32)
    d_in = plan.alloc(max(shard.t_in, 1) * 8)
    planes = [plan.alloc((shard.own_len + spill) * 4) for _ in range(3)]
    try:
        plan.h2d(d_in, local)
        plan.process_device(d_in, shard.t_in, shard.own_len, planes[0], planes[1], planes[2], shard.t_out)
        if world > 1:
            seam.exchange(planes, shard.own_len, spill)
            seam.wait()          # bounded: a peer that never entered the all-reduce must not hold the download below for ever
        outs = [np.empty(shard.own_len, dtype=np.float32) for _ in range(3)]
        for o, d in zip(outs, planes):
            plan.d2h(o, d)
    finally:
        plan.free(d_in)
        for d in planes:
            plan.free(d)
    return tuple(outs)


def process_rank(plan, stereo: np.ndarray, rank: int, world: int, seam: Optional["RcclSeam"] = None):
    """
    One rank's share of a time-sharded run on its own GPU (one process per GPU):
    upload shard `rank` of `stereo` (+ right halo), run every band, exchange the overlap-add seam over
    RCCL and return (shard, (center, left, right)) for the samples this rank owns.
    `stereo` is the whole [T,2] signal (a memory-mapped file is fine: only the shard is touched).
    """
    geo = ShardGeometry(plan.block_sizes, plan.hops)
    shard = geo.plan(stereo.shape[0], world)[rank]
    local = stereo[shard.start:shard.start + shard.t_in]
    return shard, process_local_shard(plan, local, shard, geo, world, seam)


class _Watchdog:
    """
    Ends the process if the body does not return in time.  For the one blocking collective call that has no error path of
    its own (ncclCommInitRank: a rank whose peer never arrives waits inside it for ever; ctypes has released the GIL, so
    the timer thread runs).  The main thread is inside a C call that will never return, so no exception can reach it: the
    watchdog writes the reason to stderr and sends the process SIGTERM (default action: status 143; launchers report it as a
    failed rank).  An embedding application may have installed a Python SIGTERM handler: Python runs handlers on the main
    thread only - the thread that is blocked -, so such a handler never runs and the process would stay; after `grace`
    seconds the watchdog therefore ends the process directly (exit status 143) whatever handlers exist.  A fresh process is the only
    restart, never a re-exec.
    """

    GRACE_S = 3.0

    def __init__(self, seconds: float, what: str, grace: Optional[float] = None):
        self.seconds, self.what, self._timer = float(seconds), what, None
        self.grace = self.GRACE_S if grace is None else float(grace)

    def _fire(self):
        import os
        import signal
        import sys
        import time
        print(f"[upmix_amd] {self.what} did not return within {self.seconds:g} s: a peer never arrived; terminating",
              file=sys.stderr, flush=True)
        os.kill(os.getpid(), signal.SIGTERM)
        # still here: SIGTERM is handled (or ignored) by the host application and its handler cannot run, see above
        time.sleep(max(self.grace, 0.0))
        print(f"[upmix_amd] SIGTERM did not end the process within {self.grace:g} s (a handler is installed); exit status 143",
              file=sys.stderr, flush=True)
        os._exit(143)

    def __enter__(self):
        import threading
        if self.seconds > 0:
            self._timer = threading.Timer(self.seconds, self._fire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def comm_timeout(env=None) -> float:
    """Seconds a rank waits for its peers in a collective before it gives up: UPX_COMM_TIMEOUT, else UPX_RDZV_TIMEOUT, else 600."""
    import os
    env = os.environ if env is None else env
    for key in ("UPX_COMM_TIMEOUT", "UPX_RDZV_TIMEOUT"):
        try:
            v = float(env.get(key, ""))
            if v > 0:
                return v
        except ValueError:
            pass
    return 600.0


class RcclSeam:
    """
    One RCCL communicator per process/GPU for the seam all-reduce (upx_comm_* in the C ABI).
    `broadcast(payload_or_None) -> bytes` hands rank 0's 128-byte RCCL id to every rank (rendezvous.Rendezvous
    .broadcast_bytes in the product entries).

    Failure containment (the reference fails with plain exceptions, main.py:40-41; a collective has no error path):
    `all_ok(ok, message)` - rendezvous.Rendezvous.all_ok - is voted BEFORE the blocking ncclCommInitRank (rank 0 has an
    id, every rank got this far) and AFTER it (every rank has a communicator; a rank that has one while a peer failed
    aborts it); a watchdog ends the process if the init itself never returns; `wait()` bounds the wait for an
    exchange and aborts the communicator when a peer never entered it (upx_comm_wait).
    """

    def __init__(self, plan, rank: int, world: int, broadcast: Callable[[Optional[bytes]], bytes],
                 all_ok: Optional[Callable[[bool, str], None]] = None, timeout: Optional[float] = None):
        self._lib = _lib.load()
        self.plan = plan
        self.handle = None
        self.timeout = comm_timeout() if timeout is None else float(timeout)
        vote = all_ok if all_ok is not None else (lambda ok=True, message="": None)
        uid, err = None, ""
        if rank == 0:
            try:
                buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
                _lib.check(self._lib.upx_comm_unique_id(buf))
                uid = buf.raw
            except Exception as exc:   # noqa: BLE001 - reported to every rank by the vote
                err = f"{type(exc).__name__}: {exc}"
        vote(not err, err)
        if err:
            raise _lib.UpmixHipError(err)
        uid = broadcast(uid)
        handle = C.c_void_p()
        with _Watchdog(self.timeout, f"rank {rank}: ncclCommInitRank"):
            rc = self._lib.upx_comm_create(C.byref(handle), plan.handle, rank, world, uid)
        err = "" if rc == _lib.UPX_OK else self._lib.upx_last_error().decode("utf-8", "replace")
        if rc == _lib.UPX_OK:
            self.handle = handle
        try:
            vote(not err, err)
        except Exception:
            self.abort()
            raise
        if err:
            raise _lib.UpmixHipError(err)

    def exchange(self, d_planes: Sequence[int], own_len: int, spill: int) -> None:
        _lib.check(self._lib.upx_comm_seam_exchange(self.handle, *(C.c_void_p(p) for p in d_planes),
                                                    int(own_len), int(spill)))

    def wait(self, timeout: Optional[float] = None) -> None:
        """Until the last queued exchange has run; aborts the communicator and raises when `timeout` (default: the
        communicator's) runs out or RCCL reports an error - instead of a stream synchronisation that never returns."""
        _lib.check(self._lib.upx_comm_wait(self.handle, -1.0 if timeout is None else float(timeout)))

    def abort(self) -> None:
        if self.handle:
            self._lib.upx_comm_abort(self.handle)

    def selftest(self, d_planes: Sequence[int], own_len: int, spill: int, n_rows: int, my_row: int) -> None:
        """pack -> ncclAllReduce -> add with an n_rows seam on this communicator (see upx_comm_seam_selftest)."""
        _lib.check(self._lib.upx_comm_seam_selftest(self.handle, *(C.c_void_p(p) for p in d_planes), int(own_len),
                                                    int(spill), int(n_rows), int(my_row)))

    def close(self) -> None:
        if self.handle:
            self._lib.upx_comm_destroy(self.handle)
            self.handle = None
