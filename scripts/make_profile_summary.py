#!/usr/bin/env python3
"""
Copy the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked):
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py`
  profiles/<tag>_pmc_summary.json   per-kernel PMC averages (separate --pmc passes, scripts/pmc.sh)
  profiles/pmc_traffic.json         HBM bytes per launch per kernel, corrected as
                                    MI355X_MICROARCH.md prescribes for gfx950:
                                    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   [FETCH_SIZE counts 1/2 of reads]
Usage: python scripts/make_profile_summary.py <tag> <pmc_tag>
"""
import glob
import json
import os
import shutil
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, pmc_tag = sys.argv[1], sys.argv[2]
stats = sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_r01", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats[-1], os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
summary = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "scripts", "pmc_summary.py"), pmc_tag]))
json.dump(summary, open(os.path.join(root, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
traffic = {}
for k, v in summary.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        traffic[k] = {
            "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024),
            "fetch_size_kib_raw": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"],
            "note": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request), separate --pmc passes, "
                    "averaged over the launches of this kernel in bench.py --steps 2 --warmup 1",
            "source": f"profiles/{tag}_pmc_summary.json",
        }
json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
