set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_arith.py -m gpu -x -q > gpurun_out/r02_s2_arith.log 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "verify or range or r1cs or combined or config" > gpurun_out/r02_s2_tests.log 2>&1
WL=$R/gpurun_out/wl_burst
python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > gpurun_out/wl_burst.log 2>&1
for q in 1 0; do for tnp in 4 2; do
echo "== QUAD $q TNP $tnp"
BPGPU_HORNER_QUAD=$q BPGPU_TABLE_NP=$tnp BURST_KS=1,1,1,20,20,20,64,64,1024 python3 tools/burst_probe.py $WL.1024 16 | grep K=
done; done > gpurun_out/r02_s2_burst.log 2>&1
echo "== QUAD 1 TNP 4 q24 inflight 20" >> gpurun_out/r02_s2_burst.log
GPU_MAX_HW_QUEUES=24 BURST_KS=1,1,20,20,20,64,64,1024 python3 tools/burst_probe.py $WL.1024 20 | grep K= >> gpurun_out/r02_s2_burst.log 2>&1
