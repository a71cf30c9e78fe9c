#!/bin/bash
# Round 5: stream seams inside the fused launch (UPX_SEAM_INKERNEL: 0 never, 1 always, 2 when the launch does not fill the chip)
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r5f
UPX_SEAM_INKERNEL=1 timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5f/pytest_inkernel1.txt 2>&1; echo "pytest (UPX_SEAM_INKERNEL=1) rc=$?"; tail -3 gpurun_out/r5f/pytest_inkernel1.txt
for rep in 1 2; do
for wl in c1 c2 c3; do
  for m in 0 1 2; do
    steps=200; [ $wl = c3 ] && steps=30
    UPX_SEAM_INKERNEL=$m timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 20 --no-cpu-baseline --no-e2e > gpurun_out/r5f/${wl}_m${m}_$rep.json 2> gpurun_out/r5f/${wl}_m${m}_$rep.err || { echo FAILED; tail -3 gpurun_out/r5f/${wl}_m${m}_$rep.err; }
    python - $wl $m gpurun_out/r5f/${wl}_m${m}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%s seam_inkernel=%s  %.4f ms/step  " % (sys.argv[1], sys.argv[2], d["ms_per_step"]) + "  ".join("%.4f" % l["ms"] for l in d["launches"]), flush=True)
PY
  done
done
done
