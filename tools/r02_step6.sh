set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or combined or pippenger or full_size or ipp" > gpurun_out/r02_s6_tests.log 2>&1
python3 tools/bench_msm.py > gpurun_out/r02_s6_bench_msm.log 2>&1
python3 tools/bench_mpc_msm.py > gpurun_out/r02_s6_bench_mpc_msm.log 2>&1
