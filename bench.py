#!/usr/bin/env python3
"""
bench.py - headline benchmark: stereo Msamples/s upmixed (6 bands, STFT <= 8192).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (all 6 bands of BASELINE.json configs[2]:
10 min of 48 kHz stereo, edges 0/30/120/480/1920/7680 Hz, STFT sizes
[8192,8192,8192,4096,1024,256], Blackman-Harris, 75 % overlap, raised-cosine
crossovers) over synthetic stereo already resident in HBM.  With N > 1 ranks the
signal is N x 10 min, time-sharded on the hop_max grid (one 10-min shard per
GPU, weak scaling) and every step ends with the single RCCL all-reduce of the
overlap-add seam.  torch.distributed (gloo) is used only
for the rendezvous, barriers and the max-over-ranks of the wall time; all GPU
work goes through libupmix_hip.so.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 48000
SECONDS = 600
EDGES = [0, 30, 120, 480, 1920, 7680]
MAX_STFT = 8192
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE_BAND = 20  # 8 B stereo in + 12 B Ls/C/Rs out, per band


def synth(total, seed):
    rng = np.random.default_rng(seed)
    m = rng.standard_normal(total)
    s = rng.standard_normal(total)
    x = np.empty((total, 2), dtype=np.float32)
    x[:, 0] = 0.1 * (m + 0.5 * s)
    x[:, 1] = 0.1 * (m - 0.5 * s)
    return x


def cpu_baseline(target_seconds=15.0):
    """
    The oracle in the reference's scheduling shape (ThreadPoolExecutor(), one task per band, sequential
    frame loop per band) on a bounded prefix of the same workload: a 10 s calibration slice sizes the
    timed sample so that it costs about `target_seconds` of CPU wall time.
    """
    from oracle import upmix_oracle as orc
    bands = orc.plan_bands(EDGES, 0.75, orc.win_blackman_harris, SR, max_block_size=MAX_STFT)

    def run(seconds):
        total = int(SR * seconds)
        x = synth(total, 2).astype(np.float64)
        t0 = time.perf_counter()
        orc.extract_multi_band_threadpool(x[:, 0], x[:, 1], bands)
        return total, time.perf_counter() - t0

    total, dt = run(10.0)
    sample_seconds = float(min(SECONDS, max(10.0, 10.0 * target_seconds / max(dt, 1e-3))))
    if sample_seconds > 10.0:
        total, dt = run(sample_seconds)
    else:
        sample_seconds = 10.0
    # the same frame loops one band after the other (no thread pool): SURVEY 8(d) asks for both
    xs = synth(int(SR * 10.0), 2).astype(np.float64)
    t0 = time.perf_counter()
    orc.extract_multi_band(xs[:, 0], xs[:, 1], bands, per_band=orc.band_process_streaming)
    serial = len(xs) / (time.perf_counter() - t0) / 1e6
    return {
        "value": round(total / dt / 1e6, 4), "unit": "Msamples/s", "cores": len(bands), "kind": "port",
        "bands_serial_value": round(serial, 4),
        "sample": f"first {sample_seconds:g} s of the same workload (seed 2), oracle/upmix_oracle.py "
                  f"extract_multi_band_threadpool: ThreadPoolExecutor(), one task per band (={len(bands)} threads), "
                  f"float64 numpy.fft, {os.cpu_count()} host cpus visible, {dt:.1f} s wall",
    }


def load_pmc_traffic(kernel_tag):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary, if present."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
        return rec.get(kernel_tag, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=SECONDS, help="audio per GPU (default: 600 = configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import upmix_amd as ux
    from upmix_amd import sharding, _lib
    n_dev = _lib.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if local_rank >= n_dev:
        # rehearsal of the N-rank path on a box with fewer GPUs (ranks share a device); not a valid bench line
        print(f"[bench] WARNING: LOCAL_RANK {local_rank} >= {n_dev} visible device(s): sharing device {local_rank % n_dev}",
              file=sys.stderr)
    local_rank = local_rank % n_dev

    nominal = int(SR * args.seconds)
    bands = ux.chain_bands(EDGES, 0.75, ux.make_blackman_harris, SR, max_block_size=MAX_STFT, verbose=False,
                           device=local_rank)
    plan = ux.DevicePlan(bands, device=local_rank)
    geo = sharding.ShardGeometry([b.block_size for b in bands], [b.hop_size for b in bands])
    # N ranks: one signal of N x 10 min cut on the shard grid; rank g owns shard g (+ right halo, + spill)
    shards = geo.plan(nominal * world, world)
    shard = shards[rank]
    own, t_in, t_out = shard.own_len, shard.t_in, shard.t_out
    spill = geo.spill if world > 1 else 0

    # synthetic stereo: shard g = seed (2, g) (N=1: seed 2); right halo = head of the next shard
    x = synth(own, 2 if world == 1 else (2, rank))
    if t_in > own:
        x = np.concatenate([x, synth(shards[rank + 1].own_len, (2, rank + 1))[:t_in - own]])
    d_in = plan.alloc(t_in * 8)
    d_out = [plan.alloc((own + spill) * 4) for _ in range(3)]
    plan.h2d(d_in, x)
    del x

    comm = None
    if world > 1 and os.environ.get("UPX_BENCH_REHEARSAL") == "1":
        # Rehearsal on a box with fewer GPUs than ranks (ranks share a device, which RCCL refuses): the seam goes
        # through host memory + gloo.  Exercises everything but RCCL; the JSON line says so and is not a bench result.
        class _GlooSeam:
            def exchange(self, planes, own_len, spill_):
                import torch
                host = [np.empty(own_len + spill_, dtype=np.float32) for _ in range(3)]
                for h, d in zip(host, planes):
                    plan.d2h(h, d)
                seam = sharding.pack_seam(host, shard, world, spill_)
                t = torch.from_numpy(seam)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                sharding.apply_seam(host, shard, t.numpy())
                for h, d in zip(host, planes):
                    plan.h2d(d, h)

            def close(self):
                pass
        comm = _GlooSeam()
    elif world > 1:
        comm = sharding.RcclSeam(plan, rank, world, broadcast=lambda b: sharding.broadcast_bytes_gloo(dist, b))

    def barrier():
        plan.sync()
        if dist is not None:
            dist.barrier()

    def step():
        plan.process_device(d_in, t_in, own, d_out[0], d_out[1], d_out[2], t_out)
        if comm is not None:
            comm.exchange(d_out, own, spill)

    for _ in range(args.warmup):
        step()
    plan.enable_timing(True)
    band_ms = np.zeros(len(bands))
    barrier()
    t0 = time.perf_counter()
    # HIP events on the plan's stream around each band kernel, kept per call by the library (64 calls) and read
    # after the loop: no synchronisation inside the timed region
    for i in range(args.steps):
        step()
        if (i + 1) % 64 == 0 and i + 1 < args.steps:
            band_ms += plan.band_times_sum_ms(64)
    plan.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if args.steps % 64 or args.steps == 0:
        band_ms += plan.band_times_sum_ms(args.steps % 64) if args.steps % 64 else 0
    else:
        band_ms += plan.band_times_sum_ms(64)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    band_ms /= max(args.steps, 1)

    if rank == 0:
        total_samples = nominal * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_samples * args.steps / elapsed / 1e6
        # launches: one per group of merged bands (same STFT size / hop / windows); the dominant kernel is the
        # launch with the largest time.  Algorithmic bytes of a launch = 20 B x samples x bands it carries.
        sizes = [b.block_size for b in bands]
        groups = {}
        for i in range(len(bands)):
            leader, size = plan.band_group(i)
            groups[leader] = size
        dom = max(groups, key=lambda g: band_ms[g])
        dom_ms = float(band_ms[dom])
        algo_bytes = ALGO_BYTES_PER_SAMPLE_BAND * own * groups[dom]
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        tag = plan.band_kernel_name(dom)   # kernel symbol as rocprofv3 prints it
        out = {
            "metric": "stereo Msamples/sec upmixed (6-band, STFT<=8192)",
            "value": round(value, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]: {args.seconds:g} s of 48 kHz stereo per GPU, 6 bands "
                            f"(edges 0/30/120/480/1920/7680 Hz), STFT {sizes}, Blackman-Harris 75% WOLA, "
                            f"raised-cosine crossovers XO 0.25, export Ls/C/Rs planes",
                "samples_per_gpu": nominal,
                "x_realtime": round(total_samples / SR / (elapsed / args.steps), 1),
                "parallelism": "1 GPU" if world == 1 else (
                    f"time-sharded x{world}, one RCCL seam all-reduce per step"
                    if os.environ.get("UPX_BENCH_REHEARSAL") != "1" else f"REHEARSAL x{world} (host seam, shared device) - not a result"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": tag,
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": load_pmc_traffic(tag),
                "algorithmic_bytes_per_launch": algo_bytes,
                "bands_in_launch": groups[dom],
                "avg_launch_ms": round(dom_ms, 4),
            },
            "per_launch_ms": {f"bands {g}..{g + n - 1} (STFT {sizes[g]})": round(float(band_ms[g]), 4)
                              for g, n in sorted(groups.items())},
            "all_bands_algorithmic_GBps": round(ALGO_BYTES_PER_SAMPLE_BAND * len(bands) * own
                                                / (float(band_ms.sum()) * 1e-3) / 1e9, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
