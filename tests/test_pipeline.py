"""
Hand-over logic of the streamed host calls (upmix_amd/csrc/upx_pipeline.h: upx_process, upx_process_chunked,
upx_process_tracks) on the CPU, with injected failures.  The same header is compiled into libupmix_hip.so; here it
runs against stub submit / complete steps (tests/emu/emu.cpp: emu_pipeline).  Round 1 hung when a DOWNLOAD failed
with two or more chunks left: the timeouts below are the regression test.
"""
import ctypes

import pytest


@pytest.fixture(scope="module")
def emu():
    import __graft_entry__ as ge
    lib = ctypes.CDLL(ge.build_emulator())
    ll = ctypes.c_longlong
    lib.emu_pipeline.argtypes = [ll, ll, ll, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ll), ctypes.POINTER(ll),
                                 ctypes.c_char_p, ctypes.c_int]
    lib.emu_pipeline.restype = ctypes.c_int
    return lib


def run(lib, n, fail_submit=-1, fail_complete=-1, submit_us=50, complete_us=50):
    a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
    msg = ctypes.create_string_buffer(128)
    rc = lib.emu_pipeline(n, fail_submit, fail_complete, submit_us, complete_us, ctypes.byref(a), ctypes.byref(b), msg, 128)
    return rc, a.value, b.value, msg.value.decode()


@pytest.mark.timeout(60)
def test_all_items_flow_through(emu):
    for n in (0, 1, 2, 3, 17):
        for su, cu in ((0, 0), (200, 10), (10, 200)):
            rc, subs, comps, msg = run(emu, n, submit_us=su, complete_us=cu)
            assert (rc, subs, comps, msg) == (0, n, n, "")


@pytest.mark.timeout(60)
def test_download_failure_returns_instead_of_hanging(emu):
    # the failing download is followed by 2+ items: the submitter must be released and the call must return
    for n in (3, 4, 10, 50):
        for at in range(0, n):
            for su, cu in ((0, 0), (300, 10), (10, 300)):
                rc, subs, comps, msg = run(emu, n, fail_complete=at, submit_us=su, complete_us=cu)
                assert rc == -3 and msg == "injected download failure", (n, at, rc, msg)
                assert comps == at and subs <= min(n, at + 2)


@pytest.mark.timeout(60)
def test_submit_failure_releases_the_completer(emu):
    for n in (1, 2, 5, 20):
        for at in range(0, n):
            rc, subs, comps, msg = run(emu, n, fail_submit=at, submit_us=20, complete_us=100)
            assert rc == -3 and msg == "injected submit failure"
            assert subs == at and comps <= at


@pytest.mark.timeout(60)
def test_first_failure_wins(emu):
    rc, subs, comps, msg = run(emu, 10, fail_submit=5, fail_complete=1, submit_us=0, complete_us=2000)
    assert rc == -3 and msg in ("injected download failure", "injected submit failure")


def wav_schedule(emu, t_in, own_len, t_out, grid=4096, spill=6144, chunk=1 << 22, uniform=0, bpf=4, rate=21.0):
    ll = ctypes.c_longlong
    emu.emu_wav_schedule.argtypes = [ll, ll, ll, ll, ll, ll, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.POINTER(ll),
                                     ctypes.c_int]
    emu.emu_wav_schedule.restype = ctypes.c_int
    rows = (ll * (4 * 4096))()
    n = emu.emu_wav_schedule(t_in, own_len, t_out, grid, spill, chunk, uniform, bpf, rate, rows, 4096)
    return [tuple(rows[4 * i:4 * i + 4]) for i in range(n)]


def check_schedule(chunks, t_in, own_len, t_out, grid, spill, min_own):
    assert chunks and chunks[0][0] == 0
    pos = 0
    for i, (start, own, tin, tout) in enumerate(chunks):
        last = i == len(chunks) - 1
        assert start == pos and start % grid == 0 and own > 0
        assert tin == min(own + spill, t_in - start) and tin >= own            # the halo, clipped to the shard's input
        assert tout == (t_out - start if last else min(own + spill, t_out - start))   # never beyond the planes
        assert tin < 1 << 29 and tout < 1 << 29                                 # what one launch can index
        if len(chunks) > 1:
            assert own >= min_own, (i, own)                                     # a seam ends inside the next chunk, and more
        pos += own
    assert pos == own_len


def test_wav_chunk_schedule_properties(emu):
    """upx::wav_schedule (the chunk cut of upx_wav_shard_open, round 4) without a GPU: the chunks tile the owned range on the
    shard grid, carry the right halo / spill clipped to the shard's buffers, none is shorter than the smallest allowed or
    longer than a launch can index, short shards and ungridded plans stay whole - for ragged lengths, both sample widths,
    shards with and without a successor, uniform and balanced cuts, and 2^29 + frames."""
    grid, spill = 4096, 6144
    for own_len in (1, 5000, 65535, 65536, 300_000, 400_000, 1_000_001, 28_800_000, 86_400_000, (1 << 29) + 300_000, 691_200_000):
        for last in (True, False):
            t_out = own_len + (0 if last else spill)
            t_in = own_len + (0 if last else spill)
            for chunk, uniform in ((32768, 1), (32768, 0), (1 << 22, 0), (1 << 22, 1)):
                if own_len > 100_000_000 and chunk < (1 << 22):
                    continue                                                     # (hundreds of thousands of chunks: pointless)
                for bpf in (2, 4, 6, 8):
                    c = wav_schedule(emu, t_in, own_len, t_out, grid, spill, chunk, uniform, bpf)
                    check_schedule(c, t_in, own_len, t_out, grid, spill, max(chunk, 4 * spill))
                    if own_len < 2 * max(chunk, 4 * spill) and own_len < (1 << 29):
                        assert len(c) == 1
                    if uniform and len(c) > 1:
                        assert all(r[1] == max(chunk, 4 * spill) for r in c[:-1])
    # the plans' hops share no grid, or chunking is switched off: one chunk, whatever the length
    assert wav_schedule(emu, 10_000_000, 10_000_000, 10_000_000, grid=0) == [(0, 10_000_000, 10_000_000, 10_000_000)]
    assert len(wav_schedule(emu, 10_000_000, 10_000_000, 10_000_000, chunk=0)) == 1
    # BASELINE configs[2] as 16-bit stereo: a handful of chunks of 4-7 M frames, the modelled end within 30 % of the upload
    c3 = wav_schedule(emu, 28_800_000, 28_800_000, 28_800_000)
    assert 3 <= len(c3) <= 7 and all(4_000_000 <= r[1] <= 9_000_000 for r in c3), c3
    # kernels much slower than the link (mono 16 bit, a slow plan): the chunks grow - each one's kernels hide the next upload;
    # kernels much faster than the link: what matters is a small LAST chunk (the only kernels left exposed)
    slow = wav_schedule(emu, 100_000_000, 100_000_000, 100_000_000, bpf=2, rate=5.0)
    fast = wav_schedule(emu, 100_000_000, 100_000_000, 100_000_000, bpf=8, rate=200.0)
    assert all(b[1] >= a[1] for a, b in zip(slow, slow[1:])) and len(slow) >= 3, slow
    assert fast[-1][1] < 2 * (1 << 22), fast
