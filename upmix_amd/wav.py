"""
Minimal WAV codec (stdlib + NumPy) so the WAV-in / WAV-out surface of main.py
(main.py:43, :119, :132-140, :153) does not depend on python-soundfile, which is
absent from the image.  Reads PCM 8/16/24/32-bit and IEEE float 32/64 (plain
and WAVE_FORMAT_EXTENSIBLE); ``read`` returns float64 in [-1, 1) like
``soundfile.read``.  ``write`` defaults to PCM_16 like ``soundfile.write`` does
for .wav.  Byte-level parity with libsndfile is not claimed.

Files beyond 4 GiB: a RIFF header holds 32-bit sizes, so such files are read and written as RF64 (EBU Tech 3306:
``RF64`` + ``ds64`` chunk with 64-bit sizes, ``data`` chunk size 0xFFFFFFFF).  Writing promotes automatically when
header + samples exceed 4 GiB - BASELINE configs[3] (2 h at 96 kHz stereo) is 4.1 GB at 24 bit and 5.5 GB as
float.  Every reader here takes the FIRST ``data`` chunk of a file.
"""
from __future__ import annotations

import os
import struct
from typing import Optional, Tuple

import numpy as np

_PCM, _FLOAT, _EXT = 1, 3, 0xFFFE
_U32_MAX = 0xFFFFFFFF


# ---- header ---------------------------------------------------------------------------------------------------
def _parse(fh, path: str) -> dict:
    """Walk the chunks up to the first ``data`` chunk; the file position is left undefined."""
    head = fh.read(12)
    if len(head) < 12 or head[:4] not in (b"RIFF", b"RF64", b"BW64") or head[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    rf64 = head[:4] != b"RIFF"
    fh.seek(0, 2)
    file_size = fh.tell()
    fh.seek(12)
    fmt, data, big_data = None, None, None
    pos = 12
    while True:
        hdr = fh.read(8)
        if len(hdr) < 8:
            break
        tag, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
        if tag == b"ds64":
            body = fh.read(size)
            if len(body) >= 16:
                big_data = struct.unpack("<Q", body[8:16])[0]
        elif tag == b"fmt ":
            fmt = fh.read(size)
        elif tag == b"data":
            if size == _U32_MAX and rf64 and big_data is not None:
                size = big_data
            data = (pos + 8, size)
            break
        else:
            fh.seek(size, 1)
        if size & 1:
            fh.seek(1, 1)
        pos += 8 + size + (size & 1)
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == _EXT and len(fmt) >= 26:
        code = struct.unpack("<H", fmt[24:26])[0]
    if channels < 1 or bits < 8 or bits % 8:
        raise ValueError(f"{path}: unsupported WAV layout ({channels} channels, {bits} bits)")
    size = min(data[1], max(0, file_size - data[0]))
    block = bits // 8 * channels
    return {"code": code, "bits": bits, "channels": channels, "rate": int(rate), "n_frames": size // block,
            "data_offset": data[0], "rf64": rf64}


def info(path: str) -> dict:
    """-> dict(code, bits, channels, rate, n_frames, data_offset, rf64): header only, the payload is not read."""
    with open(path, "rb") as fh:
        return _parse(fh, path)


def _header(code: int, channels: int, samplerate: int, bits: int, n_frames: int) -> bytes:
    """RIFF header (44 bytes) or, when the file would pass 4 GiB, the RF64 form (80 bytes)."""
    block = channels * bits // 8
    size = int(n_frames) * block
    fmt = struct.pack("<HHIIHH", code, channels, int(samplerate), int(samplerate) * block, block, bits)
    riff = 4 + 8 + len(fmt) + 8 + size + (size & 1)
    if riff <= _U32_MAX and size <= _U32_MAX:
        return (b"RIFF" + struct.pack("<I", riff) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" +
                struct.pack("<I", size))
    ds64 = struct.pack("<QQQI", riff + 8 + 28, size, int(n_frames), 0)
    return (b"RF64" + struct.pack("<I", _U32_MAX) + b"WAVE" + b"ds64" + struct.pack("<I", len(ds64)) + ds64 + b"fmt " +
            struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", _U32_MAX))


# ---- decode / encode ----------------------------------------------------------------------------------------------
def _decode(raw, code: int, bits: int, path: str) -> np.ndarray:
    if code == _FLOAT and bits == 32:
        return np.frombuffer(raw, dtype="<f4").astype(np.float64)
    if code == _FLOAT and bits == 64:
        return np.frombuffer(raw, dtype="<f8").astype(np.float64)
    if code == _PCM and bits == 8:
        return (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    if code == _PCM and bits == 16:
        return np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    if code == _PCM and bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        return v.astype(np.float64) / 8388608.0
    if code == _PCM and bits == 32:
        return np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    raise ValueError(f"{path}: unsupported WAV encoding (format {code}, {bits} bits)")


def encode(data: np.ndarray, subtype: str = "PCM_16"):
    """-> (format code, bits, little-endian payload bytes) of float data in [-1, 1]: the quantisation of `write`."""
    x = np.asarray(data).astype(np.float64)
    if subtype == "FLOAT":
        return _FLOAT, 32, x.astype("<f4").tobytes()
    if subtype not in ("PCM_16", "PCM_24", "PCM_32"):
        raise ValueError(f"unsupported subtype {subtype!r}")
    bits = int(subtype[4:])
    full = float(2 ** (bits - 1) - 1)
    q = np.clip(np.rint(x * full), -full - 1, full).astype(np.int64)
    if bits == 16:
        return _PCM, bits, q.astype("<i2").tobytes()
    if bits == 32:
        return _PCM, bits, q.astype("<i4").tobytes()
    u = (q & 0xFFFFFF).astype(np.uint32).reshape(-1)
    return _PCM, bits, np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()


def subtype_layout(subtype: str) -> Tuple[int, int]:
    """-> (format code, bits) of an output subtype, without encoding anything."""
    if subtype == "FLOAT":
        return _FLOAT, 32
    if subtype not in ("PCM_16", "PCM_24", "PCM_32"):
        raise ValueError(f"unsupported subtype {subtype!r}")
    return _PCM, int(subtype[4:])


# ---- whole files --------------------------------------------------------------------------------------------------
def read(path: str) -> Tuple[np.ndarray, int]:
    """-> (data float64 [T] or [T, channels], sample_rate)."""
    meta = info(path)
    return read_range(path, 0, meta["n_frames"], meta), meta["rate"]


def write(path: str, data: np.ndarray, samplerate: int, subtype: str = "PCM_16") -> None:
    """data [T] or [T, channels], float in [-1, 1]; subtype PCM_16 | PCM_24 | PCM_32 | FLOAT."""
    x = np.asarray(data)
    if x.ndim == 1:
        x = x[:, None]
    code, bits, payload = encode(x, subtype)
    with open(path, "wb") as fh:
        fh.write(_header(code, x.shape[1], samplerate, bits, x.shape[0]))
        fh.write(payload)
        if len(payload) & 1:
            fh.write(b"\x00")


# ---- raw access for the device-side codec (upx_wav_pipeline / upx_wav_shard_*) ---------------------------------
def device_kind(meta: dict, path: str = "") -> int:
    """Sample format code of the device codec (16 / 24 / 32 integer PCM, 1032 float32); ValueError if it has none."""
    if meta["code"] == _PCM and meta["bits"] in (16, 24, 32):
        return meta["bits"]
    if meta["code"] == _FLOAT and meta["bits"] == 32:
        return 1032
    raise ValueError(f"{path}: encoding (format {meta['code']}, {meta['bits']} bits) is not handled by the device codec")


def read_raw_range(path: str, start: int, count: int, meta: Optional[dict] = None, out: Optional[np.ndarray] = None):
    """The undecoded bytes of frames [start, start + count) (clipped to the file) as uint8; only those bytes are read,
    straight into `out` (e.g. page-locked memory) when given."""
    meta = meta or info(path)
    start = max(0, min(int(start), meta["n_frames"]))
    count = max(0, min(int(count), meta["n_frames"] - start))
    block = meta["bits"] // 8 * meta["channels"]
    nbytes = count * block
    if out is not None and (out.dtype != np.uint8 or out.ndim != 1 or out.size < nbytes or not out.flags.c_contiguous):
        raise ValueError(f"read_raw_range: `out` must be a contiguous uint8 array of at least {nbytes} bytes")
    buf = np.empty(nbytes, dtype=np.uint8) if out is None else out[:nbytes]
    with open(path, "rb", buffering=0) as fh:
        fh.seek(meta["data_offset"] + start * block)
        view, got = memoryview(buf), 0
        while got < nbytes:
            n = fh.readinto(view[got:])
            if not n:
                raise ValueError(f"{path}: file ends inside the sample data")
            got += n
    return buf


def read_raw(path: str):
    """-> (samples uint8[...] raw little-endian payload, fmt code, channels, sample_rate, n_frames).
    fmt: 16 / 24 / 32 (integer PCM) or 1032 (float32); anything else raises ValueError."""
    meta = info(path)
    kind = device_kind(meta, path)
    raw = read_raw_range(path, 0, meta["n_frames"], meta)
    return raw, kind, int(meta["channels"]), int(meta["rate"]), int(meta["n_frames"])


def write_raw(path: str, payload: np.ndarray, samplerate: int, kind: int, channels: int = 2) -> None:
    """payload: uint8 little-endian interleaved samples in `kind` (16/24/32 PCM or 1032 float32)."""
    bits = 32 if kind == 1032 else kind
    code = _FLOAT if kind == 1032 else _PCM
    block = channels * bits // 8
    body = np.ascontiguousarray(payload).view(np.uint8).reshape(-1)
    with open(path, "wb") as fh:
        fh.write(_header(code, channels, samplerate, bits, body.size // block))
        fh.write(memoryview(body))
        if body.size & 1:
            fh.write(b"\x00")


# ---- ranged access for the multi-GPU driver (each rank touches only its time shard) --------------------------------
def read_range(path: str, start: int, count: int, meta=None) -> np.ndarray:
    """Frames [start, start + count) (clipped to the file) as float64 [n] or [n, channels]; only those bytes are read."""
    meta = meta or info(path)
    raw = read_raw_range(path, start, count, meta)
    x = _decode(raw, meta["code"], meta["bits"], path)
    return x.reshape(-1, meta["channels"]) if meta["channels"] > 1 else x


def create(path: str, n_frames: int, samplerate: int, subtype: str = "PCM_16", channels: int = 2) -> int:
    """Write the header of a file of n_frames frames and size the file (sparse); -> byte offset of the sample data.
    The payload is filled in afterwards with write_at (any process, any order).  RF64 beyond 4 GiB."""
    code, bits = subtype_layout(subtype)
    size = int(n_frames) * (channels * bits // 8)
    head = _header(code, channels, samplerate, bits, n_frames)
    with open(path, "wb") as fh:
        fh.write(head)
        fh.truncate(len(head) + size + (size & 1))
    return len(head)


def write_at(path: str, byte_offset: int, payload) -> None:
    """payload: bytes or a uint8 array, written at `byte_offset` of an existing file."""
    with open(path, "r+b", buffering=0) as fh:
        fh.seek(byte_offset)
        view = memoryview(payload).cast("B") if not isinstance(payload, (bytes, bytearray)) else memoryview(payload)
        done = 0
        while done < len(view):
            done += fh.write(view[done:])


def output_bytes(n_frames: int, subtype: str, channels: int = 2) -> int:
    """Size of an output file of n_frames frames (header included)."""
    code, bits = subtype_layout(subtype)
    size = int(n_frames) * (channels * bits // 8)
    return len(_header(code, channels, 48000, bits, n_frames)) + size + (size & 1)
