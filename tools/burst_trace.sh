#!/bin/bash
# Kernel timeline of short bursts (K steps after a device sync) of the default bench workload, under rocprofv3 --kernel-trace.
# Run on the GPU box from the repo root; analyse the result with tools/timeline.py.
set -e
R=$GRAFT_REPO_ROOT
WL=/tmp/bpgpu_wl_burst      # (the workload of 256 distinct batches is 0.5 GB: kept out of gpurun_out/, which is copied back)
python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
BURST_KS=${BURST_KS:-1,1,20,20} python3 $R/tools/burst_probe.py $WL.1024 16 > $R/gpurun_out/burst_plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
BURST_KS=${BURST_KS:-1,1,20,20} timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/burst_trace -o b -- python3 $R/tools/burst_probe.py $WL.1024 16 > $R/gpurun_out/burst_traced.log 2>&1
