set -e
R=$GRAFT_REPO_ROOT
cd $R
WL=$R/gpurun_out/wl_burst
python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > gpurun_out/wl_burst.log 2>&1
for cfg in "16 4" "32 4" "32 2" "16 2"; do set -- $cfg
echo "== LPM $1 TNP $2 (q24 inflight 20)"
GPU_MAX_HW_QUEUES=24 BPGPU_FIXED_LPM=$1 BPGPU_TABLE_NP=$2 BURST_KS=1,1,1,20,20,20,64,64,1024,1024 python3 tools/burst_probe.py $WL.1024 20 | grep K=
done > gpurun_out/r02_s3_burst.log 2>&1
