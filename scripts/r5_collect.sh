#!/bin/bash
# after scripts/r5_collect_b.sh (on the GPU box) has merged its files into gpurun_out/: copy what is judged into profiles/
cd "$(dirname "$0")/.."
for wl in c1 c2 c3 c4share default batch; do cp gpurun_out/r5final/bench_$wl.json profiles/r05_final_bench_$wl.json; done
cp gpurun_out/r5final/bench_driver_args.json profiles/r05_bench_driver_args.json
cp gpurun_out/r5final/bench_nopreheat.json profiles/r05_bench_nopreheat.json
cp gpurun_out/r5final/rehearsal.json profiles/r05_rehearsal_bare_gpus2.json
cp gpurun_out/r5final/fresh_process.json profiles/r05_fresh_process.json
(echo "# scripts/gpu_fuzz.py on the round-5 build (scripts/r5_collect_b.sh): seeds 501, 502, 150 random plans each, every plan also through upx_process_chunked"; for s in 501 502; do echo "seed $s: $(tail -1 gpurun_out/r5final/fuzz_$s.log)"; done; echo "GPU suite of the same call: $(tail -1 gpurun_out/r5final/tests.log)") > profiles/r05_fuzz_summary.txt
cat profiles/r05_fuzz_summary.txt
