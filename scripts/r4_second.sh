#!/bin/bash
# round 4, second GPU call: suite on the chunked WAV pipeline / device ring / streamed files, then the bench lines
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4b; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for wl in c3 c4share c1 c2; do
  timeout -k 10 400 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl FAILED"; tail -5 $O/bench_$wl.err; }
  python - $O/bench_$wl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["name"], d["ms_per_step"], "ms;", " ".join("%.4f" % l["ms"] for l in d["launches"]))
print("   e2e", json.dumps(d["e2e"])[:1500])
PY
done
# launch order experiment (UPX_FIRST_BAND): which launch writes the planes instead of read-modify-writing them
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
