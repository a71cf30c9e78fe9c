#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out/r4f; mkdir -p $O
for sdma in 1 0; do
  echo "== HSA_ENABLE_SDMA=$sdma"; HSA_ENABLE_SDMA=$sdma timeout -k 10 120 python scripts/wav_overlap_probe.py 2>&1 | tail -3
  echo "== HSA_ENABLE_SDMA=$sdma single chunk"; UPX_WAV_CHUNK=0 HSA_ENABLE_SDMA=$sdma timeout -k 10 120 python scripts/wav_overlap_probe.py 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/scripts/wav_overlap_probe.py > $O/prof.log 2>&1
f=$(ls -t $O/prof/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-150
ls $O/prof/*/ | head; g=$(ls -t $O/prof/*/*memory_copy_stats.csv 2>/dev/null | head -1); [ -n "$g" ] && cat $g | head
