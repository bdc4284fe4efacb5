#!/bin/bash
# Profiles of the default bench configuration (run on the GPU box from the repo root):
#   1. un-profiled run that writes the workload cache (the generator forks a GPU-using child: not under a profiler)
#   2. rocprofv3 --kernel-trace --stats of the pipelined default run        -> gpurun_out/prof_final/
#   3. SQ issue/wait counters of a solo (1 step in flight) run               -> gpurun_out/pmc_sq/
#   4. FETCH_SIZE / WRITE_SIZE passes of the solo run                        -> gpurun_out/pmc_FETCH_SIZE, pmc_WRITE_SIZE
# summaries: tools/pmc_sq_summary.py, tools/pmc_traffic_summary.py
set -e
R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_prof
python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_prof.log 2>&1
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-combined --no-prover --workload-cache $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o final -- python3 $R/bench.py $COMMON --steps 512 --warmup 64 > $R/gpurun_out/bench_prof.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES \
  -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/bench.py $COMMON --steps 8 --warmup 2 --inflight 1 > $R/gpurun_out/pmc_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o p -- python3 $R/bench.py $COMMON --steps 8 --warmup 2 --inflight 1 > $R/gpurun_out/pmc_$c.log 2>&1
done
