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
