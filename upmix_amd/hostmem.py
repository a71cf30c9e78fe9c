"""
Page-locked host buffers for the arrays that cross PCIe (upx_host_alloc in the C ABI).

A call into fresh pageable NumPy arrays spends most of its time outside the link: the kernel faults in and zeroes
every new page, the runtime stages the copy through its own pinned bounce buffer, and the arrays' release unmaps the
pages again (measured on the MI355X box, 10 min of 48 kHz stereo: 8 ms into arrays that were touched before, 39 ms
into fresh ones).  The pool hands out NumPy arrays that live in page-locked blocks instead: the copy
engine writes straight into them at link speed, and a block goes back to the pool when the last array (or view) on
it is garbage collected, so the next call reuses it - no allocation, no faults.

The reference returns fresh NumPy arrays (center_extraction.py:503-513) that the caller owns and scales in place
(main.py:95-97); arrays from the pool behave the same (owned by the caller, writable, any lifetime).
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import sys
import threading
from typing import Dict, List

import numpy as np

from . import _lib

_GRANULE = 1 << 21   # block sizes are multiples of 2 MiB


class _Lease:
    """Owner object of one block: NumPy arrays made from it keep it alive; its death returns the block."""

    def __init__(self, pool: "PinnedPool", ptr: int, cap: int, nbytes: int):
        self._pool, self._ptr, self._cap = pool, ptr, cap
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        pool, self._pool = self._pool, None
        if pool is not None:
            pool._give_back(self._ptr, self._cap)


class PinnedPool:
    """Blocks of page-locked memory of one device plan's process, recycled by capacity."""

    _ASKED_MAX = 256         # capacities remembered for lazy pinning (oldest forgotten first)

    def __init__(self, limit_bytes: int):
        self.limit = int(limit_bytes)
        self._free: Dict[int, List[int]] = {}
        # blocks handed back by dying arrays: a finaliser may run at ANY allocation point (cyclic garbage collection),
        # also inside take() on the same thread, so it must not take the pool's lock: it only appends here (atomic)
        self._returned: "collections.deque" = collections.deque()
        self._held = 0            # bytes pinned by this pool (free + leased)
        self._lock = threading.Lock()
        self._plan_handle = None  # any live upx_plan of the process (allocation needs a device context)
        self.closed = False
        self.plans_live = True    # False between the close of the process' last plan and the creation of the next one
        # lazy pinning (take(lazy=token)): capacity -> the call that first asked for a block of it
        self._asked: Dict[int, int] = {}
        self._calls = 0
        self._libref = None       # the loaded library (bound on first use: the sweeper must not look it up at shutdown)
        self._sweeper = None      # thread that unpins blocks which come back while no plan is alive (trim)

    def new_call(self) -> int:
        """Token of one host-buffer call (DevicePlan.process ...): its result arrays are requested with lazy=token."""
        with self._lock:
            self._calls += 1
            return self._calls

    def take(self, nbytes: int, plan_handle, lazy: int = 0) -> np.ndarray:
        """
        uint8[nbytes] in page-locked memory; plain pageable memory if the pool is at its limit or pinning fails.

        lazy = a new_call() token: pinning a block costs ~0.25 ms per MiB (92 ms for the three result planes of 10 min of
        audio against 41 ms for the whole call into pageable arrays), which only pays when the block is used again.  The
        reference's flow is ONE call per process (main.py:78-80), so the FIRST call that asks for a capacity gets pageable
        arrays (what the reference returns, center_extraction.py:503-513); a later call that asks for the same capacity
        has proven reuse and pins.  Idle blocks of the right capacity are always used.
        """
        nbytes = int(nbytes)
        if nbytes <= 0 or self.closed or self.limit <= 0:
            return np.empty(max(nbytes, 0), dtype=np.uint8)
        cap = -(-nbytes // _GRANULE) * _GRANULE
        lib = self._libref = self._libref or _lib.load()
        with self._lock:
            self._collect()
            stack = self._free.get(cap)
            ptr = stack.pop() if stack else None
            if ptr is None and lazy:
                first = self._asked.setdefault(cap, lazy)
                while len(self._asked) > self._ASKED_MAX:        # sizes that differ on every call (a directory of tracks)
                    self._asked.pop(next(iter(self._asked)))     # must not grow this for ever: the oldest request goes
                if first == lazy:
                    # a plan is alive again (this call comes from one): the sweeper of an earlier idle period must stop
                    self._plan_handle = plan_handle
                    self.plans_live = True
                    return np.empty(nbytes, dtype=np.uint8)
            if ptr is None:
                # make room by releasing idle blocks of other sizes before giving up
                while self._held + cap > self.limit and any(self._free.values()):
                    k = max((k for k, v in self._free.items() if v), default=None)
                    if k is None:
                        break
                    lib.upx_host_free(plan_handle, C.c_void_p(self._free[k].pop()))
                    self._held -= k
                if self._held + cap > self.limit:
                    return np.empty(nbytes, dtype=np.uint8)
                p = C.c_void_p()
                if lib.upx_host_alloc(plan_handle, C.byref(p), cap) != _lib.UPX_OK or not p.value:
                    return np.empty(nbytes, dtype=np.uint8)
                ptr = p.value
                self._held += cap
            self._plan_handle = plan_handle
            self.plans_live = True
        return np.asarray(_Lease(self, ptr, cap, nbytes))

    def _give_back(self, ptr: int, cap: int) -> None:
        # Runs in a finaliser, i.e. at any allocation point of any thread: append only.  No lock, no accounting and above
        # all no runtime call here (hipHostFree synchronises the device); take(), trim() and the sweeper thread do all of
        # that under the pool's lock.
        self._returned.append((ptr, cap))

    def _collect(self) -> None:
        """Move returned blocks to the free lists (call with the lock held)."""
        while True:
            try:
                ptr, cap = self._returned.popleft()
            except IndexError:
                return
            self._free.setdefault(cap, []).append(ptr)

    def _release_idle(self, plan_handle) -> None:
        """Unpin every idle block (call with the lock held).  plan_handle None: page-locked memory is not tied to a device."""
        lib = self._libref = self._libref or _lib.load()
        self._collect()
        for cap, stack in self._free.items():
            while stack:
                if lib.upx_host_free(plan_handle, C.c_void_p(stack.pop())) == _lib.UPX_OK:
                    self._held -= cap

    def _sweep(self) -> None:
        # Between the close of the process' last plan and the next plan: results that outlived every plan come back one by
        # one; nothing will ask the pool for memory soon, so they are unpinned as they arrive instead of staying page-locked
        # for the rest of the process.  Ends when nothing is leased any more, a plan exists again or the pool is closed.
        import time
        while True:
            time.sleep(0.25)
            if sys.is_finalizing():
                return
            if not self._returned and not self.plans_live and not self.closed:
                continue               # nothing came back since the last look: no lock, no runtime call (hipHostFree synchronises)
            with self._lock:
                if self.plans_live or self.closed:
                    self._sweeper = None
                    return
                try:
                    self._release_idle(None)
                except Exception:      # library gone: the runtime frees the blocks at unload
                    self._sweeper = None
                    return
                if self._held <= 0:
                    self._sweeper = None
                    return

    def trim(self, plan_handle) -> None:
        """Release every idle block (called when the last plan closes, while a device context still exists)."""
        with self._lock:
            self._release_idle(plan_handle)
            self.plans_live = False     # (DevicePlan.close calls this for the process' last plan)
            self._asked.clear()
            # (never at interpreter shutdown - a plan closed by its finaliser: a thread started then cannot run and
            # Thread.start() would wait for it for ever)
            if (self._held > 0 and self._sweeper is None and not self.closed and not sys.is_finalizing()
                    and threading.main_thread().is_alive()):
                self._sweeper = threading.Thread(target=self._sweep, name="upx-pinned-pool-sweeper", daemon=True)
                self._sweeper.start()

    def close(self) -> None:
        with self._lock:
            self.closed = True

    def pinned_bytes(self) -> int:
        """Bytes this pool holds page-locked right now (idle + leased blocks)."""
        with self._lock:
            return self._held

    def reset_for_tests(self, plan_handle=None) -> dict:
        """
        The pool as a fresh process finds it: idle blocks unpinned, the lazy-pinning memory ("which call first asked for
        this size") forgotten.  Blocks still leased to live arrays stay theirs.  Returns {"held": bytes still pinned
        (= leased), "idle_released": bytes unpinned}.  For tests whose assertions depend on which call is the first of
        its size - they must not reach into the pool's private fields.
        """
        with self._lock:
            before = self._held
            self._release_idle(plan_handle)
            self._asked.clear()
            return {"held": self._held, "idle_released": before - self._held}


def default_limit(env=os.environ, meminfo: str = "/proc/meminfo") -> int:
    """
    Cap of the pool when UPX_PINNED_POOL_MB is not set: 8 GiB, but never more than a quarter of the host's RAM divided
    by the ranks that share the host (LOCAL_WORLD_SIZE, else WORLD_SIZE: one process per GPU, each with its own pool) -
    eight ranks on a 2 TiB node may pin 64 GiB together, eight ranks on a 128 GiB node 4 GiB each.
    """
    cap = 8192 << 20
    try:
        ranks = max(1, int(env.get("LOCAL_WORLD_SIZE") or env.get("WORLD_SIZE") or 1))
        with open(meminfo) as fh:
            for line in fh:
                if line.startswith("MemTotal:"):
                    total = int(line.split()[1]) << 10
                    cap = min(cap, total // 4 // ranks)
                    break
    except (OSError, ValueError):
        pass
    return cap


def _limit_from_env() -> int:
    """UPX_PINNED_POOL_MB (read once): cap of the pool in MiB, 0 disables it (every result is a plain pageable NumPy
    array, as the reference returns them); unset: default_limit()."""
    v = os.environ.get("UPX_PINNED_POOL_MB")
    if v is None:
        return default_limit()
    try:
        return int(v) << 20
    except ValueError:
        return default_limit()


POOL = PinnedPool(_limit_from_env())


def is_pinned(arr) -> bool:
    """Whether a NumPy array (or any view of one) lives in a block of the page-locked pool."""
    base = arr
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    return isinstance(base, _Lease)


def empty(nbytes_or_shape, dtype, plan_handle, lazy: int = 0) -> np.ndarray:
    """np.empty(shape, dtype) in pooled page-locked memory (lazy = POOL.new_call() token: see PinnedPool.take)."""
    shape = (nbytes_or_shape,) if np.isscalar(nbytes_or_shape) else tuple(nbytes_or_shape)
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    return POOL.take(n, plan_handle, lazy).view(dt).reshape(shape)
