#!/bin/bash
# Round 6 evidence, call A (GPU box): rocprofv3 kernel stats + PMC passes of every workload.  Merge, then on the build
# machine `bash scripts/r6_summarise.sh` writes profiles/r06_* and profiles/pmc_traffic.json (tied to the kernel sources'
# hash) BEFORE call B records the final bench lines, so that every committed line carries launches[].traffic.
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
O=$R/gpurun_out/r6final; mkdir -p $O
for wl in c3 c1 c2 c4share default batch ov50 ov875 wide65536; do
  timeout -k 10 300 bash scripts/prof.sh r6_$wl --workload $wl > $O/prof_$wl.log 2>&1 || echo "prof $wl failed"
  tail -1 $O/prof_$wl.log | cut -c1-120
done
for wl in c3 c4share default c1 c2 wide65536; do
  timeout -k 10 600 bash scripts/pmc.sh r6_$wl --workload $wl > $O/pmc_$wl.log 2>&1 || echo "pmc $wl failed"
  echo "pmc $wl done"
done
echo done
