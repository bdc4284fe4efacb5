set -o pipefail
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pippenger or msm" > gpurun_out/m_tests.log 2>&1 || { tail -20 gpurun_out/m_tests.log; exit 1; }
tail -2 gpurun_out/m_tests.log
python3 tools/bench_msm.py 2>&1 | tail -3
for lg in 17 20; do bash tools/prof_msm.sh $lg && python3 tools/timeline.py $(find gpurun_out/msm_trace_$lg -name "*.db" | head -1) 1 | grep -E "fine_sort|taskdesc|merge_heavy|coarse" | tail -4 ; rm -rf gpurun_out/msm_trace_$lg; done
