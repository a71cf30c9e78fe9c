#!/bin/bash
# Per-phase cycle stamps of the band kernel (s_memtime after every `each` of one wave), synthetic data with C3-like
# gain occupancy.  Build here (no GPU needed), run on the GPU box:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -o scripts/phase_prof/prof.exe scripts/phase_prof/prof.hip
#   gpurun -- 'bash scripts/phase_prof/run.sh' ; python scripts/phase_prof/ana.py all > profiles/<tag>_phase_cycles.txt
# Arguments of prof.exe: log2N wide F workgroups wave n_gain gain_lo gain_hi accumulate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
for cfg in "13 1 56 256 0 2 0 0.04 0" "13 1 56 256 7 2 0 0.04 0" "12 1 28 512 0 1 0.02 0.08 1" "10 0 28 2048 0 1 0.08 0.32 1" "8 0 28 2048 0 1 0.32 2 1"; do
  set -- $cfg
  PROF_OUT=gpurun_out/prof/p_$1_$2_w$5.bin timeout -k 5 60 scripts/phase_prof/prof.exe $cfg || exit 1
done
