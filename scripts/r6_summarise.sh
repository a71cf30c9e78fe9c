#!/bin/bash
# after scripts/r6_collect_a.sh has merged its files into gpurun_out/: profiles/r06_<workload>_{kernel_stats.csv,bench.json,pmc_summary.json}, pmc_traffic.json
cd "$(dirname "$0")/.."
for wl in c4share default c1 c2 wide65536 c3; do python scripts/make_profile_summary.py r06_$wl r6_$wl r6_$wl --keep > /dev/null 2>&1 || echo "summary $wl failed"; done
for wl in batch ov50 ov875; do
  cp gpurun_out/prof_r6_$wl/kernel_stats.csv profiles/r06_${wl}_kernel_stats.csv; tail -1 gpurun_out/prof_r6_$wl/bench.json > profiles/r06_${wl}_bench.json
done
python - <<'PY'
import json, bench
d = json.load(open('profiles/pmc_traffic.json'))
print(list(d), d['_kernel_sources_sha256_16'], bench.kernel_sources_sha())
PY
