#!/bin/bash
# how the step scales with the signal length (are planes that fit the 256 MB Infinity Cache cheaper per sample?)
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4d; mkdir -p $O
for rep in 1 2; do
for sec in 75 150 300 600 1200; do
  timeout -k 10 200 python bench.py --workload c3 --seconds $sec --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $O/len_${sec}_$rep.json 2>/dev/null
  python - $O/len_${sec}_$rep.json $sec <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
n = d["config"]["samples_per_gpu"]
print("seconds", sys.argv[2], "%.4f ms/step" % d["ms_per_step"], "%.2f ps/sample" % (d["ms_per_step"] * 1e9 / n), " ".join("%.4f" % l["ms"] for l in d["launches"]))
PY
done
done
