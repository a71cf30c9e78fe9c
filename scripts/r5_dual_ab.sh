#!/bin/bash
# Round 5: A/B of the dual-stream experiment kernels (UPX_DUAL) and the scratch size of the band-limited path inside ONE
# gpurun call.  Prints ms per step and per launch group.
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r5b
run() {   # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e \
      > gpurun_out/r5b/$tag.json 2> gpurun_out/r5b/$tag.err || { echo "FAILED $tag"; tail -5 gpurun_out/r5b/$tag.err; return 1; }
  python - "$tag" gpurun_out/r5b/$tag.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("%-28s %.4f ms/step  " % (sys.argv[1], d["ms_per_step"]) + "  ".join("%.4f" % l["ms"] for l in d["launches"]), flush=True)
PY
}
for rep in 1 2; do
  run base_$rep UPX_X=0 || exit 1
  run dual1_$rep UPX_DUAL=1 || exit 1
  run dual_nofence_$rep UPX_DUAL=1 UPMIX_HIP_LIB=$PWD/exp/ab/dual_nofence.so || exit 1
done
for mb in 12 24 48 96 384; do
  run scratch_$mb UPX_ZOOM_SCRATCH_MB=$mb || exit 1
done
