#!/bin/bash
# Shader clock and socket power (rocm-smi, every 0.2 s) while the verification workload idles, runs continuously and runs in 20-step bursts
# (tools/power_phases.py): is the sustained rate power-limited?  Output: gpurun_out/smi_phases.log
cd $GRAFT_REPO_ROOT
WL=/tmp/bpgpu_wl_burst
python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > gpurun_out/wl_burst.log 2>&1
(for i in $(seq 1 120); do echo "$(date +%s.%N | cut -c1-13) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics Package Power\|mclk" | sed 's/.*: //' | tr "\n" " ")"; sleep 0.2; done) > /tmp/smi.log &
python3 tools/power_phases.py $WL.1024 > /tmp/ph.log 2>/dev/null
kill %1 2>/dev/null
cat /tmp/ph.log
python3 - <<'PY'
ph = [(float(l.split()[0]), l.split(None, 1)[1].strip()) for l in open('/tmp/ph.log')]
sm = []
for l in open('/tmp/smi.log'):
    p = l.split(None, 1)
    if len(p) == 2: sm.append((float(p[0]), p[1].strip()))
for (t0, w), (t1, _) in zip(ph, ph[1:]):
    v = [x for t, x in sm if t0 + 0.5 < t < t1 - 0.1]
    print(f"--- {w}: {len(v)} samples")
    for x in v[:3] + v[-3:]: print("     ", x[:160])
PY
