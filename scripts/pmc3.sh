#!/bin/bash
# Instruction-cache counters for the bench workload.  Usage on the GPU box: bash scripts/pmc3.sh <tag>
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.err || echo "pmc $name failed"; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run ic2 SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_INPUT_VALID_READYB SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH
