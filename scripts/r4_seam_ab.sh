#!/bin/bash
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4seam; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do for v in 0 1; do
  UPX_SEAM_VEC=$v timeout -k 10 120 python bench.py --workload c3 --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/v${v}_$rep.json 2>/dev/null
  python - $O/v${v}_$rep.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("seam vec", sys.argv[2], "%.4f ms/step" % d["ms_per_step"], " ".join("%.4f" % l["ms"] for l in d["launches"]))
PY
done; done
for s in 431; do timeout -k 10 250 python scripts/gpu_fuzz.py $s 250 > $O/f_$s.log 2>&1; tail -1 $O/f_$s.log; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > /dev/null 2>&1
f=$(ls -t $GRAFT_REPO_ROOT/$O/prof/*/*kernel_stats.csv | head -1); grep -i "seam" $f | cut -c1-140
