#!/bin/bash
# after scripts/r4_final.sh (on the GPU box) has merged its files into gpurun_out/: copy what is judged into profiles/
cd "$(dirname "$0")/.."
for wl in c4share default c1 c2 c3; do python scripts/make_profile_summary.py r04_$wl r4_$wl r4_$wl --keep > /dev/null 2>&1 || echo "summary $wl failed"; done
cp gpurun_out/prof_r4_batch/kernel_stats.csv profiles/r04_batch_kernel_stats.csv; tail -1 gpurun_out/prof_r4_batch/bench.json > profiles/r04_batch_bench.json
for wl in c1 c2 c3 c4share default batch; do cp gpurun_out/r4final/bench_$wl.json profiles/r04_final_bench_$wl.json; done
cp gpurun_out/r4final/bench_driver_args.json profiles/r04_bench_driver_args.json
cp gpurun_out/r4final/bench_nopreheat.json profiles/r04_bench_nopreheat.json
cp gpurun_out/r4final/rehearsal.json profiles/r04_rehearsal_bare_gpus2.json
cp gpurun_out/r4final/stream_chunk_rate.txt profiles/r04_stream_chunk_rate.txt
(echo "# scripts/wav_overlap_probe.py on one MI355X (end of round 4): C3's signal as PCM16 from page-locked memory through upx_wav_pipeline"; cat gpurun_out/r4final/wav_overlap_probe.txt) > profiles/r04_wav_overlap_probe.txt
python - <<'PY'
import json, bench
d = json.load(open('profiles/pmc_traffic.json'))
print(list(d), d['_kernel_sources_sha256_16'], bench.kernel_sources_sha())
PY
