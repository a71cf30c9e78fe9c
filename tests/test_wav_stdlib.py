"""
upmix_amd/wav.py and the device codec's payloads against an INDEPENDENT WAV implementation that is in the image: the
standard library's `wave` module (PCM 16 / 24 / 32 bit; it knows neither IEEE float nor RF64).

The reference reads and writes through python-soundfile / libsndfile (main.py:43, 119-153), which this image does not
have, so byte-level parity with libsndfile stays unpinned.  What these tests pin instead:
  * container: header fields (channels, sample width, rate, frame count), little-endian byte order and the payload bytes of
    every PCM file wav.py writes are what `wave` reads, and every PCM file `wave` writes is what wav.py reads - mono and
    stereo, the odd-length pad byte included;
  * quantisation: it is a CHOICE, stated here: round-half-even of x * (2^(b-1) - 1), clipped to [-2^(b-1), 2^(b-1) - 1]
    (libsndfile's default float -> int conversion scales by 2^(b-1) - 1 as well when normalisation is on; its rounding is
    lrint = half-even).  Decoding divides by 2^(b-1), as soundfile.read does.
"""
import os
import struct
import wave

import numpy as np
import pytest

from upmix_amd import wav

BITS = {"PCM_16": 16, "PCM_24": 24, "PCM_32": 32}


def ints_from_wave_bytes(raw: bytes, width: int) -> np.ndarray:
    """Little-endian signed integers of `width` bytes each, decoded without wav.py."""
    if width == 2:
        return np.array(struct.unpack("<%dh" % (len(raw) // 2), raw), dtype=np.int64)
    if width == 4:
        return np.array(struct.unpack("<%di" % (len(raw) // 4), raw), dtype=np.int64)
    out = np.empty(len(raw) // 3, dtype=np.int64)
    for i in range(len(out)):
        out[i] = int.from_bytes(raw[3 * i:3 * i + 3], "little", signed=True)
    return out


def ints_to_wave_bytes(v, width: int) -> bytes:
    return b"".join(int(x).to_bytes(width, "little", signed=True) for x in v)


def expected_quantisation(x: np.ndarray, bits: int) -> np.ndarray:
    """The documented choice, written out with Python integers and round() (half-even), no NumPy rounding."""
    full = 2 ** (bits - 1) - 1
    out = []
    for v in np.asarray(x, dtype=np.float64).reshape(-1):
        q = round(float(v) * full)              # Python's round: half to even, exact on floats
        out.append(max(-full - 1, min(full, q)))
    return np.array(out, dtype=np.int64)


@pytest.mark.parametrize("subtype", sorted(BITS))
@pytest.mark.parametrize("channels", [1, 2])
@pytest.mark.parametrize("frames", [0, 1, 777])          # 777 mono frames at 24 bit: odd payload -> pad byte
def test_files_written_by_wav_py_are_what_the_stdlib_reads(tmp_path, subtype, channels, frames):
    bits = BITS[subtype]
    rng = np.random.default_rng(bits * 10 + channels)
    x = rng.uniform(-1.0, 1.0, size=(frames, channels))
    if frames > 8:
        full = 2 ** (bits - 1) - 1
        # the corners of the quantiser: exact halves (ties), full scale, beyond full scale (clipped), zero, negative zero
        x.reshape(-1)[:8] = [0.5 / full, 1.5 / full, -0.5 / full, 2.5 / full, 1.0, -1.0, 1.7, -1.7][:x.size]
    path = str(tmp_path / "a.wav")
    wav.write(path, x if channels > 1 else x[:, 0], 44100, subtype)
    with wave.open(path, "rb") as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (channels, bits // 8, 44100, frames)
        assert w.getcomptype() == "NONE"
        raw = w.readframes(frames)
    assert len(raw) == frames * channels * bits // 8
    got = ints_from_wave_bytes(raw, bits // 8)
    assert np.array_equal(got, expected_quantisation(x, bits))
    # the payload bytes in the file are exactly what encode() returns, behind a 44-byte canonical header, padded to even
    blob = open(path, "rb").read()
    code, b, payload = wav.encode(x, subtype)
    assert (code, b) == (1, bits) and blob[44:44 + len(payload)] == payload == raw
    assert len(blob) == 44 + len(payload) + (len(payload) & 1)
    assert struct.unpack("<I", blob[4:8])[0] == len(blob) - 8 and struct.unpack("<I", blob[40:44])[0] == len(payload)


@pytest.mark.parametrize("width", [2, 3, 4])
@pytest.mark.parametrize("channels", [1, 2])
def test_files_written_by_the_stdlib_are_what_wav_py_reads(tmp_path, width, channels):
    bits = 8 * width
    rng = np.random.default_rng(width)
    lo, hi = -2 ** (bits - 1), 2 ** (bits - 1) - 1
    v = rng.integers(lo, hi + 1, size=501 * channels, dtype=np.int64)
    v[:4] = [lo, hi, 0, -1]
    path = str(tmp_path / "b.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(width)
        w.setframerate(48000)
        w.writeframes(ints_to_wave_bytes(v, width))
    meta = wav.info(path)
    assert (meta["channels"], meta["bits"], meta["rate"], meta["n_frames"], meta["code"]) == (channels, bits, 48000, 501, 1)
    data, rate = wav.read(path)
    assert rate == 48000 and data.dtype == np.float64 and data.shape == ((501, channels) if channels > 1 else (501,))
    assert np.array_equal(data.reshape(-1), v.astype(np.float64) / float(2 ** (bits - 1)))     # soundfile.read's scaling
    # the undecoded route of the device codec sees the same bytes
    raw, kind, ch, sr, n = wav.read_raw(path)
    assert (kind, ch, sr, n) == (bits, channels, 48000, 501) and raw.tobytes() == ints_to_wave_bytes(v, width)
    # ranged reads (what a multi-GPU rank does) agree with the stdlib's positioned reads
    with wave.open(path, "rb") as w:
        w.setpos(100)
        want = ints_from_wave_bytes(w.readframes(57), width)
    part = wav.read_range(path, 100, 57)
    assert np.array_equal(part.reshape(-1), want.astype(np.float64) / float(2 ** (bits - 1)))


def test_round_trip_through_both_implementations(tmp_path):
    """wav.py -> stdlib -> wav.py: quantised values survive unchanged (decode o encode is the identity on the PCM grid)."""
    for subtype, bits in BITS.items():
        full = 2 ** (bits - 1) - 1
        grid = np.array([-full - 1, -full, -1, 0, 1, 12345 % full, full], dtype=np.int64)
        x = np.stack([grid / (full + 1.0), grid[::-1] / (full + 1.0)], axis=1)       # what a decoded file holds
        p1, p2 = str(tmp_path / f"{subtype}_1.wav"), str(tmp_path / f"{subtype}_2.wav")
        wav.write(p1, x, 96000, subtype)
        with wave.open(p1, "rb") as r, wave.open(p2, "wb") as w:
            w.setparams(r.getparams())
            w.writeframes(r.readframes(r.getnframes()))
        assert open(p1, "rb").read() == open(p2, "rb").read()          # the stdlib re-writes the same bytes, header included
        back, rate = wav.read(p2)
        # x * full rounds to the grid value except at the two ends, where the grid is asymmetric: -full-1 maps to -full (|x| = 1
        # times 2^(b-1) - 1), which is the documented scaling, not an error
        want = np.clip(np.rint(x * full), -full - 1, full) / (full + 1.0)
        assert rate == 96000 and np.array_equal(back, want)


@pytest.mark.gpu
@pytest.mark.parametrize("in_bits,out_bits", [(16, 16), (24, 24), (32, 16), (16, 32)])
def test_device_codec_payloads_through_the_stdlib(tmp_path, in_bits, out_bits):
    """A file written by the stdlib goes through the device codec (upx_wav_pipeline: decode, all bands, peak scale, export
    layout, quantisation on the GPU); the payloads that come back, written with wav.write_raw, are files the stdlib reads -
    and their samples are the host export's (wav.encode of export.py's arrays), value for value."""
    import upmix_amd as ux
    from upmix_amd import export
    rng = np.random.default_rng(in_bits + out_bits)
    frames = 30000
    width = in_bits // 8
    hi = 2 ** (in_bits - 1) - 1
    v = np.clip(np.rint(rng.standard_normal(frames * 2) * 0.1 * hi), -hi - 1, hi).astype(np.int64)
    src = str(tmp_path / "in.wav")
    with wave.open(src, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(width)
        w.setframerate(48000)
        w.writeframes(ints_to_wave_bytes(v, width))
    raw, kind, ch, sr, n = wav.read_raw(src)
    bands = ux.chain_bands([0, 300, 3000], 0.75, ux.make_blackman_harris, sr, max_block_size=4096, threshold_factor=64,
                           verbose=False)
    plan = ux.DevicePlan(bands)
    outs, stats = plan.wav_pipeline(raw, kind, ch, n, "split", out_bits)
    # the host flow on the same file (main.py:43-157 restated in export.py / wav.py)
    data, _ = wav.read(src)
    c, l, r = ux.extract_center_left_right_multi_band_in_memory(data[:, 0], data[:, 1], sr, bands)
    export.scale_to_input_peak(c, l, r, export.input_peak(data))
    host = export.export_arrays("split", c, l, r)
    for name, payload in outs.items():
        path = str(tmp_path / f"{name}.wav")
        wav.write_raw(path, payload, sr, out_bits, 2)
        with wave.open(path, "rb") as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (2, out_bits // 8, sr, n)
            got = ints_from_wave_bytes(w.readframes(n), out_bits // 8)
        want = ints_from_wave_bytes(wav.encode(host[name], f"PCM_{out_bits}")[2], out_bits // 8)
        assert np.array_equal(got, want), name
    plan.close()
