#!/bin/bash
# PMC passes for a bench workload (separate runs: two SQ sets, FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE), as
# MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes (no --kernel-trace / --stats in a --pmc run).
# Usage on the GPU box: bash scripts/pmc.sh <tag> [bench args, e.g. --workload default]
TAG=${1:-r02}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --preheat-ms 0 --no-cpu-baseline --no-e2e --no-reserve $BENCH_ARGS > $OUT/$name.json 2> $OUT/$name.err || echo "pmc $name failed"
}
BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
ls $OUT
