#!/bin/bash
# bursts of 20 steps (the driver's bench call) under the tuning knobs of the verification chain: median of 12 bursts per setting.
# The knobs are seeded through the environment (BPGPU_* -> per-context options of include/bpgpu.h); the first argument of run() is the
# number of contexts the steps alternate between.
R=$GRAFT_REPO_ROOT
WL=/tmp/bpgpu_wl_burst      # (the workload of 256 distinct batches is 0.5 GB: kept out of gpurun_out/, which is copied back)
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
KS=20,20,20,20,20,20,20,20,20,20,20,20,1024
run() {
  local inflight=$1; shift
  env "$@" BURST_KS=$KS python3 $R/tools/burst_probe.py $WL.1024 $inflight | grep K= > /tmp/b20.txt
  python3 - "inflight=$inflight $*" <<'PY'
import sys, re
v20, v1k = [], []
for l in open('/tmp/b20.txt'):
    m = re.match(r"K=\s*(\d+):.*=\s*([\d.]+) M/s", l)
    if m: (v20 if int(m.group(1)) == 20 else v1k).append(float(m.group(2)))
v20.sort()
print(f"{sys.argv[1]:72s} K=20: min {v20[0]:.2f} median {v20[len(v20)//2]:.2f} max {v20[-1]:.2f}   K=1024: {v1k}", flush=True)
PY
}
run 20 GPU_MAX_HW_QUEUES=24
run 8 GPU_MAX_HW_QUEUES=24
run 12 GPU_MAX_HW_QUEUES=24
run 16 GPU_MAX_HW_QUEUES=24
run 24 GPU_MAX_HW_QUEUES=24
run 20 GPU_MAX_HW_QUEUES=32
run 32 GPU_MAX_HW_QUEUES=48
run 20 GPU_MAX_HW_QUEUES=24 BPGPU_TABLE_NP=2
run 20 GPU_MAX_HW_QUEUES=24 BPGPU_TABLE_NP=1
run 20 GPU_MAX_HW_QUEUES=24 BPGPU_TABLE_NP=2 BPGPU_FIXED_LPM=32
run 20 GPU_MAX_HW_QUEUES=24 BPGPU_GROUPS_FORM=2
run 20 GPU_MAX_HW_QUEUES=24 BPGPU_HORNER_FORM=3
run 20 GPU_MAX_HW_QUEUES=24 BPGPU_HORNER_FORM=3 BPGPU_TABLE_NP=1 BPGPU_GROUPS_FORM=2
run 20 GPU_MAX_HW_QUEUES=24 BURST_LATENCY_MODE=1
