#!/bin/bash
# SQ issue/wait counters per kernel of a solo (1 step in flight) bench run -> gpurun_out/pmc_sq/; summary by tools/pmc_sq_summary.py
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES \
  -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-combined --steps 8 --warmup 2 --inflight 1 ${BENCH_ARGS} > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq.log 2>&1
