set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_gpu_arith.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/q4_tests.log 2>&1 || { tail -20 gpurun_out/q4_tests.log; exit 1; }
tail -2 gpurun_out/q4_tests.log
bash tools/solo_trace.sh > /dev/null 2>&1
grep -E "back|final" gpurun_out/solo_trace.txt | tail -2; grep -E "back" gpurun_out/solo_trace_lat.txt | tail -1
python3 tools/bench_msm.py > gpurun_out/logs_bench_msm3.log 2>&1; cat gpurun_out/logs_bench_msm3.log
