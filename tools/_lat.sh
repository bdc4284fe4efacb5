set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "latency_mode or launch_variants or range_verify" > gpurun_out/lat_tests.log 2>&1 || { tail -20 gpurun_out/lat_tests.log; exit 1; }
tail -2 gpurun_out/lat_tests.log
R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
for gq in 0 1; do
echo "== latency mode, groups quad $gq"
BPGPU_GROUPS_QUAD=$gq BURST_LATENCY_MODE=1 BURST_KS=1,1,1,1 python3 $R/tools/burst_probe.py $WL.1024 1 | grep K=
done
