#!/bin/bash
# (round 6: the library reads UPX_* knobs only with UPX_TUNING=1; the round-4/5 experiment knobs this script drives also need an
# experiment build: __graft_entry__.build_hip(extra_flags=["-DUPX_EXPERIMENTS"], lib="exp/ab/experiments.so") + UPMIX_HIP_LIB)
export UPX_TUNING=1
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_multi_gpu.py -m gpu -x -q -k "wav or stream or sharded or file" > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for cfg in "UPX_WAV_KERNEL_RATE=21" "UPX_WAV_KERNEL_RATE=16" "UPX_WAV_KERNEL_RATE=28" "UPX_WAV_KERNEL_RATE=40" "UPX_WAV_KERNEL_RATE=21 UPX_WAV_CHUNK=2097152" "UPX_WAV_KERNEL_RATE=21 UPX_WAV_CHUNK=8388608" "UPX_WAV_UNIFORM=1"; do
  for wl in c3 c4share; do
  env $cfg timeout -k 10 200 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/w.json 2> $O/w.err
  python - $O/w.json "$cfg" $wl <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["e2e"]
w = d["upx_wav_pipeline_pcm16_stereo_sum"]
print(sys.argv[3], sys.argv[2], "pinned %.2f ms (begin %.2f tail %.2f finish %.2f)" % (w["ms"], w["begin_ms"], w["begin_tail_ms"], w["finish_ms"]), "pageable %.2f" % d["upx_wav_pipeline_pcm16_stereo_sum_pageable_input"]["ms"], d.get("multi_gpu_run_rank_pcm24_stereo_sum", ""))
PY
  done
done
