#!/bin/bash
# per-kernel durations of ONE un-pipelined batch (rocprofv3 --kernel-trace of single-step bursts), default and latency mode;
# analyse with tools/timeline.py
set -e
R=$GRAFT_REPO_ROOT
WL=/tmp/bpgpu_wl_burst      # (the workload of 256 distinct batches is 0.5 GB: kept out of gpurun_out/, which is copied back)
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/solo_trace $R/gpurun_out/solo_trace_lat
BURST_KS=1,1,1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/solo_trace -o s -- python3 $R/tools/burst_probe.py $WL.1024 1 > $R/gpurun_out/solo_traced.log 2>&1
export BURST_LATENCY_MODE=1
BURST_KS=1,1,1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/solo_trace_lat -o s -- python3 $R/tools/burst_probe.py $WL.1024 1 > $R/gpurun_out/solo_traced_lat.log 2>&1
cd $R
for d in solo_trace solo_trace_lat; do
  db=$(find gpurun_out/$d -name '*.db' | head -1)
  python3 tools/timeline.py $db 1 | tail -12 > gpurun_out/$d.txt      # the last two batches (the warm-up runs without a gap)
done
cat gpurun_out/solo_trace_lat.txt
