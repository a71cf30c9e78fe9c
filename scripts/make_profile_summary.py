#!/usr/bin/env python3
"""
Copy the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked):
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py` (scripts/prof.sh <prof_tag>)
  profiles/<tag>_pmc_summary.json   per-kernel PMC averages (separate --pmc passes, scripts/pmc.sh <pmc_tag>)
  profiles/pmc_traffic.json         HBM bytes per launch per workload and kernel, corrected as MI355X_MICROARCH.md prescribes for
                                    gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  [FETCH_SIZE counts 1/2 of reads];
                                    entries of earlier tags are kept, `_kernel_sources_sha256_16` says which kernel
                                    sources the LAST update was collected from (bench.py refuses other builds)
Usage: python scripts/make_profile_summary.py <tag> <prof_tag> <pmc_tag> [--keep]
"""
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402

tag, prof_tag, pmc_tag = sys.argv[1], sys.argv[2], sys.argv[3]
keep = "--keep" in sys.argv
stats = os.path.join(root, "gpurun_out", f"prof_{prof_tag}", "kernel_stats.csv")
if os.path.exists(stats):
    shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
    line = open(os.path.join(root, "gpurun_out", f"prof_{prof_tag}", "bench.json")).read().strip().splitlines()[-1]
    open(os.path.join(root, "profiles", f"{tag}_bench.json"), "w").write(line + "\n")
summary = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "scripts", "pmc_summary.py"), pmc_tag]))
json.dump(summary, open(os.path.join(root, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
path = os.path.join(root, "profiles", "pmc_traffic.json")
# the workload the PMC passes ran (their bench lines are kept beside the counter files)
workload = "c3"
try:
    workload = json.loads(open(os.path.join(root, "gpurun_out", f"pmc_{pmc_tag}", "fetch.json")).read().strip().splitlines()[-1])["config"]["name"]
except Exception:
    pass
traffic = {}
if keep and os.path.exists(path):
    traffic = {k: v for k, v in json.load(open(path)).items() if not k.startswith("_")}
entry = {}
for k, v in summary.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        entry[k] = {
            "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
            "fetch_size_kib_raw": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"],
            "note": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request), separate --pmc passes, "
                    "averaged over the launches of this kernel in bench.py --steps 2 --warmup 1",
            "source": f"profiles/{tag}_pmc_summary.json",
        }
traffic[workload] = entry            # per workload: the same kernel moves other bytes on another signal / plan
traffic["_kernel_sources_sha256_16"] = bench.kernel_sources_sha()
json.dump(traffic, open(path, "w"), indent=1)
print(json.dumps(traffic, indent=1))
