"""One-GPU check of what the N > 1 bench / multi_gpu processes do in ONE process: torch (its bundled ROCm runtime) imported and a
gloo group initialised FIRST, then libupmix_hip.so (system ROCm) with an RCCL communicator obtained through dlopen.  Prints which
librccl / libamdhip64 files are mapped and runs the seam self-test (pack -> ncclAllReduce -> add) plus one band plan.
Usage (GPU box): python scripts/rccl_with_torch_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29611")
import numpy as np
import torch
import torch.distributed as dist

dist.init_process_group(backend="gloo", rank=0, world_size=1)
import upmix_amd as ux
from upmix_amd import sharding
from oracle import upmix_oracle as orc


def mapped(pattern):
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if pattern in l})


bands = ux.chain_bands([0, 300, 3000], 0.75, ux.make_blackman_harris, 48000, max_block_size=1024, verbose=False, device=0)
plan = ux.DevicePlan(bands, device=0)
seam = sharding.RcclSeam(plan, 0, 1, broadcast=lambda b: sharding.broadcast_bytes_gloo(dist, b))
print("librccl mapped:", mapped("librccl"))
print("libamdhip64 mapped:", mapped("libamdhip64"))
own, spill = 50000, 6144
rng = np.random.default_rng(5)
host = [rng.standard_normal(own + spill).astype(np.float32) for _ in range(3)]
d = [plan.alloc((own + spill) * 4) for _ in range(3)]
for p, h in zip(d, host):
    plan.h2d(p, h)
seam.exchange(d, own, spill)
seam.selftest(d, own, spill, 8, 5)
plan.sync()
ok = True
for p, h in zip(d, host):
    got = np.empty_like(h)
    plan.d2h(got, p)
    want = h.copy()
    want[:spill] += h[own:own + spill]
    ok &= bool(np.array_equal(got, want))
x = orc.synthetic_stereo(60000, 3)
outs = plan.process(x)
ref = orc.extract_multi_band(x[:, 0].astype(np.float64), x[:, 1].astype(np.float64),
                             orc.plan_bands([0, 300, 3000], 0.75, orc.win_blackman_harris, 48000, max_block_size=1024))
err = max(float(np.sqrt(np.mean((o.astype(np.float64) - r) ** 2))) for o, r in zip(outs, ref))
t = torch.tensor([1.0], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("seam self-test", "ok" if ok else "FAILED", " band plan rms err %.2e" % err)
sys.stdout.flush()
os._exit(0 if ok and err < 1e-6 else 1)
