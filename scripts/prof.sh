#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench workload.  Usage on the GPU box: bash scripts/prof.sh <tag> [bench args]
# Leaves gpurun_out/prof_<tag>/.../*_kernel_stats.csv and prints its top rows.
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-reserve "$@" > $OUT/bench.json 2> $OUT/bench.err
f=$(ls -t $OUT/*/*kernel_stats.csv | head -1)
cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:95]:95s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
tail -c 600 $OUT/bench.json
