set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q -k "msm or combined or pippenger or full_size or multirank or points_sum" > gpurun_out/r02_s5_tests.log 2>&1
bash tools/prof_combined.sh
python3 tools/bench_msm.py > gpurun_out/r02_s5_bench_msm.log 2>&1
