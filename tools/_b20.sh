R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 20 --warmup 5 --workload-cache $WL 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prof   ', d['value'], d['ms_per_step'])"; done
for i in 1 2 3; do BPGPU_BENCH_NOPROF=1 python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 20 --warmup 5 --workload-cache $WL 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noprof ', d['value'], d['ms_per_step'])"; done
