set -e
R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
for q in 16 24; do for inf in 16 20 24 32; do
echo "== queues $q inflight $inf"
GPU_MAX_HW_QUEUES=$q BURST_KS=20,20,20,64,64,1024 python3 $R/tools/burst_probe.py $WL.1024 $inf | grep K=
done; done
