set -o pipefail
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/inv_tests.log 2>&1 || { tail -20 gpurun_out/inv_tests.log; exit 1; }
tail -2 gpurun_out/inv_tests.log
R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
for t in 2 1; do
echo "== latency mode, TNP $t"
BPGPU_TABLE_NP=$t BURST_LATENCY_MODE=1 BURST_KS=1,1,1,1 python3 $R/tools/burst_probe.py $WL.1024 1 | grep K=
done
echo "== throughput"
GPU_MAX_HW_QUEUES=24 BURST_KS=20,20,20,20,20,20,1024,1024 python3 $R/tools/burst_probe.py $WL.1024 20 | grep K=
python3 tools/bench_msm.py > gpurun_out/logs_bench_msm4.log 2>&1; cat gpurun_out/logs_bench_msm4.log
