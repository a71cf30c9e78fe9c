#!/bin/bash
# Vector-memory path counters (TA / TCP / TCC) of a bench workload; same pass discipline as pmc.sh.
# Usage on the GPU box: bash scripts/pmc_mem.sh <tag> [bench args]; summary: python scripts/pmc_summary.py <tag>
TAG=${1:-mem}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  # (a set the hardware cannot collect aborts rocprofv3, which then never exits: bound every pass)
  timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e $BENCH_ARGS > $OUT/$name.json 2> $OUT/$name.err || echo "pmc $name failed"
  echo "pass $name done"
}
BENCH_ARGS="$*"
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum
run tcp3 TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
run tcp4 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run grbm GRBM_GUI_ACTIVE
ls $OUT
