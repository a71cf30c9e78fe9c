#!/bin/bash
# launch order experiment (UPX_FIRST_BAND): which launch writes the planes instead of read-modify-writing them
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4b; mkdir -p $O
for rep in 1 2; do
  for fb in -1 5 4 3; do
    UPX_FIRST_BAND=$fb timeout -k 10 120 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/fb_${fb}_$rep.json 2>/dev/null
    python - $O/fb_${fb}_$rep.json $fb <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("first band", sys.argv[2], "%.4f ms/step" % d["ms_per_step"], " ".join("%.4f" % l["ms"] for l in d["launches"]))
PY
  done
done
