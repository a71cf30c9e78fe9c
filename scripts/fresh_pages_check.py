"""Where the time of `upx_process` into freshly allocated NumPy arrays goes (GPU box): pipeline into touched buffers, the same into
fresh arrays, the release of the arrays, and the same after MADV_HUGEPAGE + MADV_POPULATE_WRITE of the fresh ranges.
Usage: python scripts/fresh_pages_check.py   (DESIGN.md 7 quotes it)"""
import os, sys, time, ctypes, mmap
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import upmix_amd as ux
from upmix_amd import _lib
bands = ux.chain_bands([0, 30, 120, 480, 1920, 7680], 0.75, ux.make_blackman_harris, 48000, max_block_size=8192, verbose=False)
plan = ux.DevicePlan(bands)
n = 28800000
rng = np.random.default_rng(2)
x = (0.1 * rng.standard_normal((n, 2))).astype(np.float32)
lib, check, f32p = plan._lib, _lib.check, _lib.f32p
def run(outs): check(lib.upx_process(plan.handle, x.ctypes.data_as(f32p), n, *(o.ctypes.data_as(f32p) for o in outs)))
warm = [np.zeros(n, np.float32) for _ in range(3)]
run(warm); 
for rep in range(3):
    t0 = time.perf_counter(); run(warm); t1 = time.perf_counter()
    fresh = [np.empty(n, np.float32) for _ in range(3)]; t2 = time.perf_counter()
    run(fresh); t3 = time.perf_counter()
    del fresh; t4 = time.perf_counter()
    print("warm %.2f  alloc %.2f  run-fresh %.2f  free %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
# direct madvise test
libc = ctypes.CDLL("libc.so.6", use_errno=True)
a = np.empty(n, np.float32)
addr = (a.ctypes.data + 4095) // 4096 * 4096
ln = (a.nbytes - 8192) // 4096 * 4096
t0 = time.perf_counter(); r1 = libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(ln), 14); e1 = ctypes.get_errno()
r2 = libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(ln), 23); e2 = ctypes.get_errno(); t1 = time.perf_counter()
print("madvise HUGEPAGE rc", r1, e1, " POPULATE_WRITE rc", r2, e2, " %.2f ms for %d MB" % ((t1-t0)*1e3, ln >> 20))
print(open("/proc/version").read().strip()[:80])
# fully populated (madvise) fresh arrays, populated BEFORE the call
for rep in range(3):
    fresh = [np.empty(n, np.float32) for _ in range(3)]
    t0 = time.perf_counter()
    for a in fresh:
        addr = (a.ctypes.data + 4095) // 4096 * 4096
        ln = (a.nbytes - 8192) // 4096 * 4096
        libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(ln), 14)
        libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(ln), 23)
    t1 = time.perf_counter(); run(fresh); t2 = time.perf_counter()
    run(fresh); t3 = time.perf_counter()
    del fresh; t4 = time.perf_counter()
    print("populate %.2f  run (populated, new range) %.2f  run again (same range) %.2f  free %.2f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
