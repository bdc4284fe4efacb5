#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
python3 $R/tools/prof_combined.py $WL.1024 20 > $R/gpurun_out/comb_plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/comb_trace
BURST_KS=1,1,1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/comb_trace -o c -- python3 $R/tools/prof_combined.py $WL.1024 1 > $R/gpurun_out/comb_traced.log 2>&1
