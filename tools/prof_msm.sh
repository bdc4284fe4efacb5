#!/bin/bash
# kernel timeline of one resident MSM of 2^LG terms (tools/prof_msm.py) under rocprofv3 --kernel-trace
set -e
R=$GRAFT_REPO_ROOT
LG=${1:-14}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/msm_trace_$LG
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/msm_trace_$LG -o m -- python3 $R/tools/prof_msm.py $LG > $R/gpurun_out/msm_traced_$LG.log 2>&1
