#!/bin/bash
# sweep hardware queues x steps in flight x Straus points per lane (one line per config)
for q in ${QS:-16 32}; do for s in ${INFL:-16 32}; do for np in ${NPS:-4}; do
  GPU_MAX_HW_QUEUES=$q BPGPU_STRAUS_NP=$np python bench.py --steps ${STEPS:-64} --warmup 16 --no-cpu-baseline --no-combined --no-prover --workload-cache gpurun_out/wl2 --inflight $s ${EXTRA} 2>&1 | tail -1 > /tmp/o.json
  python3 - "$q" "$s" "$np" <<'PY'
import json,sys
try:
    d=json.load(open('/tmp/o.json'))
    print("queues",sys.argv[1],"inflight",sys.argv[2],"np",sys.argv[3],"->",round(d["value"]),"v/s", round(d["ms_per_step"],3),"ms", {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()})
except Exception as e:
    print(sys.argv[1:], "fail", open('/tmp/o.json').read()[-300:])
PY
done; done; done
