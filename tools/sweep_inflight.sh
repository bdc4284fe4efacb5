#!/bin/bash
# steps-in-flight / hardware-queue sweep of the default bench workload (needs a workload cache: bench.py --workload-cache X)
WL=${1:-gpurun_out/wlr}
for cfg in "16 16" "24 24" "32 32" "16 32" "24 48"; do
  set -- $cfg
  for steps in 64 2048; do
    GPU_MAX_HW_QUEUES=$1 BPGPU_INFLIGHT=$2 timeout -k 10 120 python bench.py --workload-cache $WL --no-prover --no-combined --no-cpu-baseline --steps $steps 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('queues $1 inflight $2 steps $steps:', round(d['value']), 'verif/s', round(d['ms_per_step'], 4), 'ms/step')
"
  done
done
