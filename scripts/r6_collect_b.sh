#!/bin/bash
# Round 6 evidence, call B (GPU box): suite, the final bench lines of every workload (with launches[].traffic from call A's
# counters), the driver's command, the cold line, the fresh-process drop-in cost + where a plan's creation time goes, fuzz
# (every tenth plan 5-30 M samples as one device-resident launch per group), the bare --gpus 2 rehearsal.
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
O=$R/gpurun_out/r6final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for wl in c3 c1 c2 c4share default batch; do
  timeout -k 10 400 python bench.py --workload $wl > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl FAILED"; tail -5 $O/bench_$wl.err; }
done
for wl in ov50 ov875 ov60 wide65536; do
  timeout -k 10 300 python bench.py --workload $wl --no-e2e > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo "bench $wl FAILED"; tail -5 $O/bench_$wl.err; }
done
for wl in c3 c1 c2 c4share default batch ov50 ov875 ov60 wide65536; do
  python - $O/bench_$wl.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["name"], d["value"], "Msamples/s", d["ms_per_step"], "ms;", " ".join("%.4f" % l["ms"] for l in d["launches"]), "frac", d["roofline"]["frac"], d["all_bands_frac"],
      "traffic", [l.get("traffic_ratio") for l in d["launches"]])
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2>/dev/null
timeout -k 10 200 python bench.py --preheat-ms 0 --no-cpu-baseline --no-e2e > $O/bench_nopreheat.json 2>/dev/null
timeout -k 10 200 python scripts/drop_in_fresh_process.py > $O/fresh_process.json 2>/dev/null; cat $O/fresh_process.json
timeout -k 10 100 python3 scripts/plan_create_breakdown.py > $O/plan_breakdown_c3.txt 2>&1
timeout -k 10 100 python3 scripts/plan_create_breakdown.py --max-stft 65536 > $O/plan_breakdown_default.txt 2>&1
UPX_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $O/rehearsal.json 2> $O/rehearsal.err; echo "rehearsal rc $?"
for seed in 601 602; do timeout -k 10 500 python scripts/gpu_fuzz.py $seed 150 > $O/fuzz_$seed.log 2>&1; tail -1 $O/fuzz_$seed.log; done
echo done
