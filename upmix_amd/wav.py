"""
Minimal WAV codec (stdlib + NumPy) so the WAV-in / WAV-out surface of main.py
(main.py:43, :119, :132-140, :153) does not depend on python-soundfile, which is
absent from the image.  Reads PCM 8/16/24/32-bit and IEEE float 32/64 (plain
and WAVE_FORMAT_EXTENSIBLE); ``read`` returns float64 in [-1, 1) like
``soundfile.read``.  ``write`` defaults to PCM_16 like ``soundfile.write`` does
for .wav.  Byte-level parity with libsndfile is not claimed.
"""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

_PCM, _FLOAT, _EXT = 1, 3, 0xFFFE


def read(path: str) -> Tuple[np.ndarray, int]:
    """-> (data float64 [T] or [T, channels], sample_rate)."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        body = blob[pos + 8:pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == _EXT and len(fmt) >= 26:
        code = struct.unpack("<H", fmt[24:26])[0]
    width = bits // 8
    count = len(data) // (width * channels) * channels
    raw = data[:count * width]
    if code == _FLOAT and bits == 32:
        x = np.frombuffer(raw, dtype="<f4").astype(np.float64)
    elif code == _FLOAT and bits == 64:
        x = np.frombuffer(raw, dtype="<f8").astype(np.float64)
    elif code == _PCM and bits == 8:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif code == _PCM and bits == 16:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif code == _PCM and bits == 24:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        x = v.astype(np.float64) / 8388608.0
    elif code == _PCM and bits == 32:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    else:
        raise ValueError(f"{path}: unsupported WAV encoding (format {code}, {bits} bits)")
    if channels > 1:
        x = x.reshape(-1, channels)
    return x, int(rate)


def write(path: str, data: np.ndarray, samplerate: int, subtype: str = "PCM_16") -> None:
    """data [T] or [T, channels], float in [-1, 1]; subtype PCM_16 | PCM_24 | PCM_32 | FLOAT."""
    x = np.asarray(data)
    if x.ndim == 1:
        x = x[:, None]
    channels = x.shape[1]
    x = x.astype(np.float64)
    if subtype == "FLOAT":
        code, bits, payload = _FLOAT, 32, x.astype("<f4").tobytes()
    elif subtype in ("PCM_16", "PCM_24", "PCM_32"):
        bits = int(subtype[4:])
        full = float(2 ** (bits - 1) - 1)
        q = np.clip(np.rint(x * full), -full - 1, full).astype(np.int64)
        code = _PCM
        if bits == 16:
            payload = q.astype("<i2").tobytes()
        elif bits == 32:
            payload = q.astype("<i4").tobytes()
        else:
            u = (q & 0xFFFFFF).astype(np.uint32).reshape(-1)
            payload = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
    else:
        raise ValueError(f"unsupported subtype {subtype!r}")
    block = channels * bits // 8
    fmt = struct.pack("<HHIIHH", code, channels, int(samplerate), int(samplerate) * block, block, bits)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(payload) + (len(payload) & 1)) + b"WAVE")
        fh.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        fh.write(b"data" + struct.pack("<I", len(payload)) + payload)
        if len(payload) & 1:
            fh.write(b"\x00")


# ---- raw access for the device-side codec (upx_wav_pipeline) ---------------------------------------
def read_raw(path: str):
    """-> (samples uint8[...] raw little-endian payload, fmt code, channels, sample_rate, n_frames).
    fmt: 16 / 24 / 32 (integer PCM) or 1032 (float32); anything else raises ValueError."""
    with open(path, "rb") as fh:
        blob = fh.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack("<I", blob[pos + 4:pos + 8])[0]
        if tag == b"fmt ":
            fmt = blob[pos + 8:pos + 8 + size]
        elif tag == b"data":
            data = (pos + 8, min(size, len(blob) - pos - 8))
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == _EXT and len(fmt) >= 26:
        code = struct.unpack("<H", fmt[24:26])[0]
    if code == _PCM and bits in (16, 24, 32):
        kind = bits
    elif code == _FLOAT and bits == 32:
        kind = 1032
    else:
        raise ValueError(f"{path}: encoding (format {code}, {bits} bits) is not handled by the device codec")
    width = bits // 8
    n_frames = data[1] // (width * channels)
    raw = np.frombuffer(blob, dtype=np.uint8, count=n_frames * width * channels, offset=data[0])
    return raw, kind, int(channels), int(rate), int(n_frames)


def write_raw(path: str, payload: np.ndarray, samplerate: int, kind: int, channels: int = 2) -> None:
    """payload: uint8 little-endian interleaved samples in `kind` (16/24/32 PCM or 1032 float32)."""
    bits = 32 if kind == 1032 else kind
    code = _FLOAT if kind == 1032 else _PCM
    block = channels * bits // 8
    body = np.ascontiguousarray(payload).view(np.uint8).tobytes()
    fmt = struct.pack("<HHIIHH", code, channels, int(samplerate), int(samplerate) * block, block, bits)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body) + (len(body) & 1)) + b"WAVE")
        fh.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        fh.write(b"data" + struct.pack("<I", len(body)) + body)
        if len(body) & 1:
            fh.write(b"\x00")


# ---- ranged access for the multi-GPU driver (each rank touches only its time shard) --------------------------------
def info(path: str):
    """-> dict(code, bits, channels, rate, n_frames, data_offset): header only, the payload is not read."""
    with open(path, "rb") as fh:
        head = fh.read(12)
        if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
            raise ValueError(f"{path}: not a RIFF/WAVE file")
        fmt, data = None, None
        pos = 12
        while True:
            hdr = fh.read(8)
            if len(hdr) < 8:
                break
            tag, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
            if tag == b"fmt ":
                fmt = fh.read(size)
                if size & 1:
                    fh.seek(1, 1)
            elif tag == b"data":
                data = (pos + 8, size)
                break
            else:
                fh.seek(size + (size & 1), 1)
            pos += 8 + size + (size & 1)
        fh.seek(0, 2)
        file_size = fh.tell()
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, channels, rate, _, _, bits = struct.unpack("<HHIIHH", fmt[:16])
    if code == _EXT and len(fmt) >= 26:
        code = struct.unpack("<H", fmt[24:26])[0]
    size = min(data[1], file_size - data[0])
    return {"code": code, "bits": bits, "channels": channels, "rate": int(rate),
            "n_frames": size // (bits // 8 * channels), "data_offset": data[0]}


def _decode(raw: bytes, code: int, bits: int, path: str) -> np.ndarray:
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


def read_range(path: str, start: int, count: int, meta=None) -> np.ndarray:
    """Frames [start, start + count) (clipped to the file) as float64 [n] or [n, channels]; only those bytes are read."""
    meta = meta or info(path)
    start = max(0, min(int(start), meta["n_frames"]))
    count = max(0, min(int(count), meta["n_frames"] - start))
    block = meta["bits"] // 8 * meta["channels"]
    with open(path, "rb") as fh:
        fh.seek(meta["data_offset"] + start * block)
        raw = fh.read(count * block)
    x = _decode(raw, meta["code"], meta["bits"], path)
    return x.reshape(-1, meta["channels"]) if meta["channels"] > 1 else x


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


def create(path: str, n_frames: int, samplerate: int, subtype: str = "PCM_16", channels: int = 2) -> int:
    """Write the header of a file of n_frames frames and size the file; -> byte offset of the sample data.
    The payload is filled in afterwards with write_at (any process, any order)."""
    code, bits, _ = encode(np.zeros(0), subtype)
    block = channels * bits // 8
    size = int(n_frames) * block
    fmt = struct.pack("<HHIIHH", code, channels, int(samplerate), int(samplerate) * block, block, bits)
    head = (b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + size + (size & 1)) + b"WAVE" +
            b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"data" + struct.pack("<I", size))
    with open(path, "wb") as fh:
        fh.write(head)
        fh.truncate(len(head) + size + (size & 1))
    return len(head)


def write_at(path: str, byte_offset: int, payload: bytes) -> None:
    with open(path, "r+b") as fh:
        fh.seek(byte_offset)
        fh.write(payload)
