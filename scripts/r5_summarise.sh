#!/bin/bash
# after scripts/r5_collect_a.sh has merged its files into gpurun_out/: profiles/r05_<workload>_{kernel_stats.csv,bench.json,pmc_summary.json}, pmc_traffic.json
cd "$(dirname "$0")/.."
for wl in c4share default c1 c2 c3; do python scripts/make_profile_summary.py r05_$wl r5_$wl r5_$wl --keep > /dev/null 2>&1 || echo "summary $wl failed"; done
cp gpurun_out/prof_r5_batch/kernel_stats.csv profiles/r05_batch_kernel_stats.csv; tail -1 gpurun_out/prof_r5_batch/bench.json > profiles/r05_batch_bench.json
python - <<'PY'
import json, bench
d = json.load(open('profiles/pmc_traffic.json'))
print(list(d), d['_kernel_sources_sha256_16'], bench.kernel_sources_sha())
PY
