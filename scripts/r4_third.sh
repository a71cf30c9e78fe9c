#!/bin/bash
# round 4, third GPU call: suite (incl. > 2^29 frames), WAV chunk size sweep (pinned / pageable source), c4share file -> file
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc $rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for ch in 2097152 4194304 8388608 16777216; do
  UPX_WAV_CHUNK=$ch timeout -k 10 200 python bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > $O/wavchunk_$ch.json 2> $O/wavchunk_$ch.err
  python - $O/wavchunk_$ch.json $ch <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["e2e"]
print("UPX_WAV_CHUNK", sys.argv[2], "pinned", d["upx_wav_pipeline_pcm16_stereo_sum"], "pageable", d["upx_wav_pipeline_pcm16_stereo_sum_pageable_input"])
PY
done
timeout -k 10 300 python bench.py --workload c4share --steps 5 --warmup 2 --no-cpu-baseline > $O/c4share.json 2> $O/c4share.err
python - $O/c4share.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["e2e"]
print({k: v for k, v in d.items() if "wav" in k or "multi" in k})
PY
df -T /tmp | cat
