import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import upmix_amd as ux
from oracle import upmix_oracle as orc
x = orc.synthetic_stereo(200000, 10)
for n, lo, hi in ((4096, 0., 300.), (1024, 3000., 24000.), (256, 7680., 24000.), (8192, 120., 480.)):
    b = ux.MultiBandExtractorAccu(n, 0.75, ux.make_blackman_harris, lo, hi, 48000, "raised_cosine", 10., 100.)
    plan = ux.DevicePlan([b])
    base = plan.process(x)
    again = plan.process(x)
    print(n, "repeat equal:", [bool(np.array_equal(u, v)) for u, v in zip(base, again)], plan.band_info(0))
    for f in (2, 4, 16, 1000):
        plan.set_blocks_per_stream(f)
        got = plan.process(x)
        for name, u, v in zip("CLR", base, got):
            d = np.nonzero(u != v)[0]
            if len(d):
                print(f"  N={n} F={f} {name}: {len(d)} differ, first {d[:5]}, last {d[-3:]}, max {np.max(np.abs(u-v)):.2e} hop={n//4}")
    plan.close()
