"""
Time sharding + seam algebra on CPU: world_size-2 gloo (one process per shard)
with the ORACLE plugged in as the per-shard engine, compared with the
unsharded oracle.  The GPU engine runs the same host logic (bench.py, -m gpu
tests); this covers the N > 1 path where no GPU is available.
"""
import os
import socket

import numpy as np
import pytest

from conftest import rms
from oracle import upmix_oracle as orc
from upmix_amd import sharding


def oracle_engine(bands):
    def run(local, own_len, t_out):
        fin = [np.zeros(t_out, np.float32) for _ in range(3)]
        for b in bands:
            res = orc.band_process(local[:, 0].astype(np.float64), local[:, 1].astype(np.float64), b,
                                   own_len=own_len, out_len=t_out)
            for f, r in zip(fin, res):
                f += r
        return fin
    return run


def test_geometry_and_plan():
    geo = sharding.ShardGeometry([8192, 8192, 8192, 4096, 1024, 256], [2048, 2048, 2048, 1024, 256, 64])
    assert (geo.hop_max, geo.grid, geo.spill, geo.halo) == (2048, 4096, 6144, 6144)
    shards = geo.plan(1_000_003, 4)
    assert shards[0].start == 0 and shards[-1].start + shards[-1].own_len == 1_000_003
    for a, b in zip(shards[:-1], shards[1:]):
        assert a.start + a.own_len == b.start and a.start % geo.grid == 0
        assert a.t_out == a.own_len + 6144 and a.t_in == a.own_len + 6144 and not a.last
    assert shards[-1].last and shards[-1].t_out == shards[-1].own_len
    with pytest.raises(ValueError):
        geo.plan(10000, 4)
    with pytest.raises(ValueError):
        sharding.ShardGeometry([1024, 512], [256, 205])


def test_sharded_equals_unsharded_in_process():
    bands = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=1024)
    geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
    x = orc.synthetic_stereo(20011, 21)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), bands)
    for n_shards in (2, 3, 5):
        shards = geo.plan(len(x), n_shards)
        eng = oracle_engine(bands)
        planes = [[np.array(p, copy=True) for p in eng(x[s.start:s.start + s.t_in], s.own_len, s.t_out)]
                  for s in shards]
        seam = sum(sharding.pack_seam(p, s, n_shards, geo.spill) for p, s in zip(planes, shards))
        out = [np.empty(len(x), np.float32) for _ in range(3)]
        for p, s in zip(planes, shards):
            sharding.apply_seam(p, s, seam)
            for o, q in zip(out, p):
                o[s.start:s.start + s.own_len] = q[:s.own_len]
        for o, r in zip(out, ref):
            assert rms(o.astype(np.float64) - r) < 1e-8     # float32 association differs only at the seams
            assert float(np.max(np.abs(o - r))) < 1e-6


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bands = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=1024)
    geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
    x = orc.synthetic_stereo(30000, 22)
    shard = geo.plan(len(x), world)[rank]

    def allreduce(seam):
        t = torch.from_numpy(seam)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    out = sharding.run_shard(x, geo, shard, world, oracle_engine(bands), allreduce)
    # (the byte broadcast of the RCCL unique id is upmix_amd.rendezvous': tests/test_rendezvous.py)
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), start=shard.start, c=out[0], l=out[1], r=out[2])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_seam_reduce(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    bands = orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=1024)
    x = orc.synthetic_stereo(30000, 22)
    ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64), bands)
    got = [np.empty(len(x), np.float32) for _ in range(3)]
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        for g, k in zip(got, "clr"):
            g[int(z["start"]):int(z["start"]) + len(z[k])] = z[k]
    for g, r in zip(got, ref):
        assert rms(g.astype(np.float64) - r) < 1e-8


def _stuck_in_a_collective(seconds):
    import time
    from upmix_amd import sharding as sh
    with sh._Watchdog(seconds, "rank 0: ncclCommInitRank"):
        time.sleep(60)      # stands for the blocking init a peer never joins


def _stuck_with_a_sigterm_handler(seconds):
    import signal
    import time
    from upmix_amd import sharding as sh
    # an embedding application's handler: Python would run it on the main thread, which never comes back from the C call
    signal.signal(signal.SIGTERM, lambda *_: None)
    with sh._Watchdog(seconds, "rank 0: ncclCommInitRank", grace=0.5):
        # a blocking call that signals do not interrupt (time.sleep resumes after a handled signal, PEP 475), like the C call
        time.sleep(60)


def test_watchdog_ends_the_process_even_when_the_application_handles_sigterm():
    """with a Python SIGTERM handler installed and the main thread blocked, SIGTERM alone never ends the process;
    after the grace period the watchdog leaves through os._exit(143)."""
    import multiprocessing
    import time
    ctx = multiprocessing.get_context("spawn")
    p = ctx.Process(target=_stuck_with_a_sigterm_handler, args=(0.5,))
    t0 = time.monotonic()
    p.start()
    p.join(30)
    assert p.exitcode == 143 and time.monotonic() - t0 < 25, p.exitcode


def test_watchdog_ends_a_process_stuck_in_a_blocking_collective():
    """RCCL's blocking init has no error path: when a peer never arrives, the watchdog ends the process (non-zero) instead of
    letting it wait for ever; a body that returns in time cancels it."""
    import multiprocessing
    import time
    from upmix_amd import sharding as sh
    ctx = multiprocessing.get_context("spawn")
    p = ctx.Process(target=_stuck_in_a_collective, args=(0.5,))
    t0 = time.monotonic()
    p.start()
    p.join(30)
    assert p.exitcode not in (0, None) and time.monotonic() - t0 < 25, p.exitcode
    with sh._Watchdog(0.3, "quick"):
        pass
    time.sleep(0.6)         # cancelled: this process is still here
    assert sh.comm_timeout({"UPX_COMM_TIMEOUT": "12.5", "UPX_RDZV_TIMEOUT": "3"}) == 12.5
    assert sh.comm_timeout({"UPX_RDZV_TIMEOUT": "3"}) == 3.0 and sh.comm_timeout({}) == 600.0
