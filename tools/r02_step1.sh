set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_mirror.py -m gpu -x -q > gpurun_out/r02_s1_tests.log 2>&1
WL=$R/gpurun_out/wl_burst
python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 20 --warmup 5 --workload-cache $WL > gpurun_out/r02_s1_bench20.json 2> gpurun_out/r02_s1_bench20.err
for tnp in 4 8 2; do
echo "== TNP $tnp"
BPGPU_TABLE_NP=$tnp BURST_KS=1,1,1,20,20,20,64,64,1024 python3 tools/burst_probe.py $WL.1024 16 | grep K=
done > gpurun_out/r02_s1_burst.log 2>&1
