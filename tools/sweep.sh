#!/bin/bash
# sweep Straus points-per-lane x steps-in-flight x table window on the GPU box (prints one line per config)
for np in ${NPS:-1 2 3 4}; do for s in ${INFL:-1 4}; do for c in ${CS:-8}; do
  BPGPU_STRAUS_NP=$np python bench.py --steps 48 --warmup 8 --no-cpu-baseline --workload-cache gpurun_out/wl2 --inflight $s --window-bits $c 2>&1 | tail -1 > /tmp/o.json
  python3 - "$np" "$s" "$c" <<'PY'
import json,sys
try:
    d=json.load(open('/tmp/o.json'))
    print("np",sys.argv[1],"inflight",sys.argv[2],"c",sys.argv[3],"->",round(d["value"]),"v/s", round(d["ms_per_step"],3),"ms", {k:round(v,2) for k,v in d["kernel_ms_per_step"].items()}, "combined", round(d["combined_batch_check"]["value"]))
except Exception as e:
    print(sys.argv[1:], "fail", open('/tmp/o.json').read()[-300:])
PY
done; done; done
