#!/bin/bash
# Sweep of an experiment build's environment knobs inside ONE gpurun call: scripts/sweep_env.sh <lib.so> <workload> "A=1 B=2" "A=2 B=2" ...
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
lib=$1; wl=$2; shift 2
mkdir -p gpurun_out/sweep
for rep in 1 2; do
  for cfg in "" "$@"; do
    tag=$(echo "$cfg" | tr ' =' '__')
    env $cfg UPMIX_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --workload "$wl" --steps 20 --warmup 3 --no-cpu-baseline --no-e2e \
        > gpurun_out/sweep/s_${tag}_$rep.json 2> gpurun_out/sweep/s_${tag}_$rep.err || { echo "FAILED $cfg"; tail -3 gpurun_out/sweep/s_${tag}_$rep.err; exit 1; }
    python - "$cfg" "$rep" gpurun_out/sweep/s_${tag}_$rep.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%-34s rep %s  %.4f ms/step  " % (sys.argv[1] or "(default)", sys.argv[2], d["ms_per_step"]) + "  ".join("%.3f" % l["ms"] for l in d["launches"]))
PY
  done
done
