#!/bin/bash
# round 4, first GPU call: suite, the new bench lines (c1, c2), c3, the bare --gpus 2 rehearsal, small-T stream sweep
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
for wl in c1 c2 c3; do
  timeout -k 10 300 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl FAILED"; tail -5 $O/bench_$wl.err; }
done
UPX_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc $?"
for f in 4 6 8 12 16; do
  for wl in c1 c2; do
    UPX_MIN_STREAM_FRAMES=$f timeout -k 10 120 python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline --no-e2e > $O/minf_${wl}_$f.json 2>/dev/null
    python - $O/minf_${wl}_$f.json $wl $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "min frames", sys.argv[3], "%.4f ms/step" % d["ms_per_step"], " ".join("%.4f(%s/%s)" % (l["ms"], l["workgroups"], l["slots"]) for l in d["launches"]))
PY
  done
done
