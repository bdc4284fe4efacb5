#!/bin/bash
# as burst20_sweep.sh, for the generator half of the back launch: lanes per MSM + butterfly against a proof per lane and a run of
# generators per wave (BPGPU_FIXED_CHUNK_GENS), and the Horner forms; K = 20 bursts (median of 12) and the sustained rate (K = 1024)
R=$GRAFT_REPO_ROOT
WL=/tmp/bpgpu_wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
KS=20,20,20,20,20,20,20,20,20,20,20,20,1024,1024
run() {
  local inflight=$1; shift
  env "$@" BURST_KS=$KS python3 $R/tools/burst_probe.py $WL.1024 $inflight 2>/dev/null | grep K= > /tmp/b20.txt
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
for g in ${SWEEP_GENS:-9 5 3 2 1}; do run 20 GPU_MAX_HW_QUEUES=24 BPGPU_FIXED_CHUNK_GENS=$g; done
for x in ${SWEEP_EXTRA}; do run 20 GPU_MAX_HW_QUEUES=24 $(echo $x | tr , ' '); done
