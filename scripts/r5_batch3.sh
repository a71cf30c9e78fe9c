#!/bin/bash
# Round 5, GPU batch 3: whole GPU suite, short-signal variants, the fresh-process drop-in cost, the full bench line.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r5c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5c/pytest.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r5c/pytest.txt
for wl in c1 c2; do
  for v in 0 1; do
    UPX_KERNEL_VARIANT=$v timeout -k 10 200 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-e2e > gpurun_out/r5c/${wl}_v$v.json 2> gpurun_out/r5c/${wl}_v$v.err
    python - $wl $v gpurun_out/r5c/${wl}_v$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3]))
print("%s variant %s  %.4f ms/step  " % (sys.argv[1], sys.argv[2], d["ms_per_step"]) + "  ".join("%s %.4f" % (l["kernel"][:40], l["ms"]) for l in d["launches"]), flush=True)
PY
  done
done
timeout -k 10 200 python scripts/drop_in_fresh_process.py > gpurun_out/r5c/fresh_process.json 2> gpurun_out/r5c/fresh_process.err; cat gpurun_out/r5c/fresh_process.json
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r5c/bench_full.json 2> gpurun_out/r5c/bench_full.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c/bench_full.json"))
print(d["ms_per_step"], d["preheat"], d["roofline"].get("lane_pattern_streaming_GBps"))
print(d["e2e"]["drop_in_entry_float64_views"])
PY
