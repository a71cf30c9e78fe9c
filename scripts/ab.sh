#!/bin/bash
# A/B of kernel builds inside ONE gpurun call (boxes differ by +-2 %): scripts/ab.sh <workload> <lib.so>...
# Each library runs the bench twice, interleaved; prints ms per step and per kernel.
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/ab
wl=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    tag=$(basename "$lib" .so)
    UPMIX_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload "$wl" --steps 20 --warmup 3 --no-cpu-baseline --no-e2e \
        > gpurun_out/ab/${tag}_${wl}_$rep.json 2> gpurun_out/ab/${tag}_${wl}_$rep.err || { echo "FAILED $tag"; tail -5 gpurun_out/ab/${tag}_${wl}_$rep.err; exit 1; }
    python - "$tag" "$rep" gpurun_out/ab/${tag}_${wl}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%-12s rep %s  %.4f ms/step  " % (sys.argv[1], sys.argv[2], d["ms_per_step"]) + "  ".join("%.3f" % l["ms"] for l in d["launches"]))
PY
  done
done
