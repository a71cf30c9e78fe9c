#!/bin/bash
# A/B of builds and knob settings inside ONE gpurun call (boxes differ by +-2 %):
#   scripts/ab_env.sh <workload> <tag>[@lib.so][:VAR=VALUE[,VAR=VALUE...]] ...
# e.g. scripts/ab_env.sh default base conc0:UPX_ZOOM_CONCURRENT=0 bfly@exp/ab/bfly.so
# Every variant runs the bench twice, interleaved (a b a b); prints ms per step and per launch entry.  The knobs need the
# opt-in (UPX_TUNING=1), which this script sets.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/ab
export UPX_TUNING=1
wl=$1; shift
for rep in 1 2; do
  for spec in "$@"; do
    head=${spec%%:*}; vars=""; [[ "$spec" == *:* ]] && vars=${spec#*:}
    tag=${head%%@*}; lib=upmix_amd/libupmix_hip.so; [[ "$head" == *@* ]] && lib=${head#*@}
    envs=(UPMIX_HIP_LIB=$PWD/$lib)
    IFS=',' read -ra kv <<< "$vars"; for e in "${kv[@]}"; do [[ -n "$e" ]] && envs+=("$e"); done
    out=gpurun_out/ab/${tag}_${wl}_$rep
    env "${envs[@]}" timeout -k 10 300 python bench.py --workload "$wl" --steps 20 --warmup 3 --no-cpu-baseline --no-e2e \
        > $out.json 2> $out.err || { echo "FAILED $tag"; tail -5 $out.err; exit 1; }
    python - "$tag" "$rep" $out.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%-14s rep %s  %.4f ms/step  " % (sys.argv[1], sys.argv[2], d["ms_per_step"]) + "  ".join("%.3f" % l["ms"] for l in d["launches"]), flush=True)
PY
  done
done
