set -o pipefail
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -m gpu -x -q -k "msm or pippenger or golden or ark" > gpurun_out/m_tests.log 2>&1 || { tail -30 gpurun_out/m_tests.log; exit 1; }
tail -2 gpurun_out/m_tests.log
python3 tools/bench_msm.py 2>&1 | tail -6
python3 tools/bench_mpc_msm.py 2>&1 | tail -5
