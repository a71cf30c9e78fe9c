"""
ctypes binding of libupmix_hip.so (C ABI: include/upmix_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  Build it with ``python __graft_entry__.py`` (hipcc,
--offload-arch=gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UPMIX_HIP_LIB", os.path.join(_HERE, "libupmix_hip.so"))   # override: kernel experiments

UPX_OK = 0
UPX_ERR_INVALID = -1
UPX_ERR_UNSUPPORTED = -2
UPX_ERR_HIP = -3
UPX_ERR_NO_DEVICE = -4
UPX_ERR_RCCL = -5
UPX_ERR_NOMEM = -6
UNIQUE_ID_BYTES = 128
PCM16, PCM24, PCM32, F32 = 16, 24, 32, 1032
EXPORT_STEREO_SUM, EXPORT_SPLIT, EXPORT_AB = 0, 1, 2
SAMPLE_F32, SAMPLE_F64 = 0, 1

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/upmix_hip.h one to one
SIGNATURES = {
    "upx_abi_version": (C.c_int, []),
    "upx_last_error": (C.c_char_p, []),
    "upx_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "upx_supported": (C.c_int, [C.c_int32, C.c_int32]),
    "upx_plan_create": (C.c_int, [vpp, C.c_int, C.c_int, i32p, i32p, f32p, f32p, f32p]),
    "upx_plan_kernel_names": (C.c_int, [C.c_int, i32p, i32p, f32p, f32p, f32p, C.c_char_p, C.c_size_t]),
    "upx_plan_destroy": (None, [C.c_void_p]),
    "upx_plan_set_blocks_per_stream": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "upx_process": (C.c_int, [C.c_void_p, f32p, C.c_int64, f32p, f32p, f32p]),
    "upx_process_chunked": (C.c_int, [C.c_void_p, f32p, C.c_int64, f32p, f32p, f32p, C.c_int64]),
    "upx_process_lr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, f32p, f32p, f32p]),
    "upx_process_tracks": (C.c_int, [C.c_void_p, C.c_int32, vpp, C.POINTER(C.c_int64), vpp, vpp, vpp]),
    "upx_dev_alloc": (C.c_int, [C.c_void_p, vpp, C.c_size_t]),
    "upx_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "upx_dev_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "upx_host_alloc": (C.c_int, [C.c_void_p, vpp, C.c_size_t]),
    "upx_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "upx_copy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "upx_copy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "upx_sync": (C.c_int, [C.c_void_p]),
    "upx_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int64]),
    "upx_plan_reserve": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "upx_plan_enable_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "upx_plan_band_times_ms": (C.c_int, [C.c_void_p, f32p, C.c_int]),
    "upx_plan_band_times_sum_ms": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_int]),
    "upx_plan_band_times_calls_ms": (C.c_int, [C.c_void_p, f32p, C.c_int, C.c_int]),
    "upx_plan_band_phase_times_sum_ms": (C.c_int, [C.c_void_p, f32p, f32p, C.c_int, C.c_int]),
    "upx_plan_band_phase_kernel_name": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "upx_plan_band_stream_starts": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int32, i32p]),
    "upx_plan_band_info": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, i32p, i32p]),
    "upx_plan_band_fill": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, i32p, i32p]),
    "upx_plan_band_kernel_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    "upx_plan_band_group": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p]),
    "upx_absmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, f32p]),
    "upx_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]),
    "upx_wav_pipeline": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "upx_wav_shard_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                      C.c_int64, C.c_int64, C.POINTER(C.c_double)]),
    "upx_wav_shard_finish": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "upx_wav_shard_open": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "upx_wav_shard_feed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "upx_wav_shard_seal": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "upx_wav_shard_finish_async": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int64, C.POINTER(C.c_int32)]),
    "upx_wav_shard_wait_piece": (C.c_int, [C.c_void_p, C.c_int32]),
    "upx_wav_shard_planes": (C.c_int, [C.c_void_p, vpp, vpp, vpp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "upx_wav_shard_peaks": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "upx_wav_pipeline_times_ms": (C.c_int, [C.c_void_p, f32p]),
    "upx_stream_chunk": (C.c_int, [C.c_void_p, f32p, C.c_int32, f32p, C.c_int32, f32p, f32p, f32p]),
    "upx_stream_state": (C.c_int, [C.c_void_p, f32p, f32p, f32p, C.c_int]),
    "upx_stream_set_state": (C.c_int, [C.c_void_p, f32p, f32p, f32p]),
    "upx_comm_unique_id": (C.c_int, [C.c_char_p]),
    "upx_comm_create": (C.c_int, [vpp, C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "upx_comm_destroy": (None, [C.c_void_p]),
    "upx_comm_seam_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "upx_comm_wait": (C.c_int, [C.c_void_p, C.c_double]),
    "upx_comm_abort": (C.c_int, [C.c_void_p]),
    "upx_comm_reserve": (C.c_int, [C.c_void_p, C.c_int64]),
    "upx_comm_seam_add_len": (C.c_int64, [C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "upx_comm_seam_selftest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                         C.c_int]),
    "upx_seam_add_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int64]),
}

_lib: Optional[C.CDLL] = None


class UpmixHipError(RuntimeError):
    """A call into libupmix_hip.so failed (HIP / RCCL / no device)."""


def load() -> C.CDLL:
    """dlopen the library and bind every symbol of the header.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UpmixHipError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python __graft_entry__.py`). "
            "upmix_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Translate a upx_status into the exception type the reference would raise."""
    if rc == UPX_OK:
        return
    msg = load().upx_last_error().decode("utf-8", "replace")
    if rc == UPX_ERR_INVALID:
        raise ValueError(msg)
    if rc == UPX_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == UPX_ERR_NOMEM:
        raise MemoryError(msg)
    raise UpmixHipError(f"upx error {rc}: {msg}")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().upx_device_count(C.byref(n))
    if rc == UPX_ERR_NO_DEVICE:
        return 0
    check(rc)
    return n.value
