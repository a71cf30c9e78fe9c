"""One-GPU check of what a rank of an N > 1 run does, in ONE process and WITHOUT torch: the process group of
upmix_amd.rendezvous (world 1 here), libupmix_hip.so with an RCCL communicator obtained through dlopen, the seam self-test
(pack -> ncclAllReduce -> add) and one band plan; prints which librccl / libamdhip64 files are mapped (one of each:
the process holds ONE HIP runtime) and leaves through a normal interpreter exit.
Usage (GPU box): python scripts/rccl_rank_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import upmix_amd as ux
from upmix_amd import sharding
from upmix_amd.rendezvous import Rendezvous
from oracle import upmix_oracle as orc


def mapped(pattern):
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if pattern in l})


group = Rendezvous.from_env()
bands = ux.chain_bands([0, 300, 3000], 0.75, ux.make_blackman_harris, 48000, max_block_size=1024, verbose=False, device=0)
plan = ux.DevicePlan(bands, device=0)
seam = sharding.RcclSeam(plan, group.rank, group.world, broadcast=group.broadcast_bytes, all_ok=group.all_ok)
print("librccl mapped:", mapped("librccl"))
print("libamdhip64 mapped:", mapped("libamdhip64"))
print("torch imported:", "torch" in sys.modules)
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
print("max over ranks:", group.allreduce_max([1.0]))
print("seam self-test", "ok" if ok else "FAILED", " band plan rms err %.2e" % err)
for p in d:
    plan.free(p)
seam.close()
plan.close()
group.close()
sys.exit(0 if ok and err < 1e-6 else 1)
