#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r5d
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5d/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5d/pytest.txt
timeout -k 10 200 python scripts/drop_in_fresh_process.py > gpurun_out/r5d/fresh_process.json 2> gpurun_out/r5d/fresh_process.err; cat gpurun_out/r5d/fresh_process.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5d/bench_full.json 2> gpurun_out/r5d/bench_full.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench_full.json"))
print(d["ms_per_step"], d["preheat"], d["roofline"].get("lane_pattern_streaming_GBps"), d.get("host_prep_us_per_call"))
PY
for wl in c1 c2; do timeout -k 10 200 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-e2e > gpurun_out/r5d/${wl}.json 2> gpurun_out/r5d/${wl}.err; python - gpurun_out/r5d/${wl}.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["name"], d["ms_per_step"], d.get("host_prep_us_per_call"), d["preheat"]["first_5_steps_ms_per_step"], d["preheat"]["five_steps_after_2s_idle_ms_per_step"])
PY
done
