#!/bin/bash
# Per-phase cycle stamps of the band-limited kernels (upx_zoom.h) on a full grid of synthetic data.  Build here:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -o scripts/phase_prof/zprof.exe scripts/phase_prof/zprof.hip
# run on the GPU box: bash scripts/phase_prof/zrun.sh > gpurun_out/zoom_phase_cycles.txt ; copy to profiles/.
# zprof.exe log2N log2P kernel(0 analysis, 1 synthesis Ls/Rs, 2 synthesis C) stamps-per-period F frames
cd $GRAFT_REPO_ROOT/scripts/phase_prof
echo "# stamps per period: analysis = 2 frames x (units x [load+pass0+scatter | B | mid f, mid g (P = 256 only: P >= 512 exchanges in registers, one stamp) | last+ramp | B | reduce | B] + mask);"
echo "#                    synthesis = front | B1 | stage+scatter | mid f, mid g (P>=512) | B2 | last pass+OLA+emit"
for cfg in "13 8 0 26 28 14056" "13 8 1 5 28 14056" "13 8 2 5 28 14056" "13 9 0 16 28 14056" "13 9 1 7 28 14056" "16 9 0 114 28 1764" "16 9 1 7 28 1764" "12 9 0 16 28 28112" "12 9 1 7 28 28112"; do
  echo "== zprof.exe $cfg"
  timeout -k 5 60 ./zprof.exe $cfg | awk '/^N=/{l=$0} /^events/{print l; print} /^period|^  pos/{print}' || exit 1
done
