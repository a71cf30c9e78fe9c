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
