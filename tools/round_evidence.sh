#!/bin/bash
# Everything DESIGN.md quotes for a round, in three gpurun calls (each well under the 20-minute limit):
#   part A: GPU test suite, the driver's bench call twice, the default (2048-step) bench
#   part B: secondary logs (tools/round_logs.sh), solo traces (tools/solo_trace.sh);  part C: rocprofv3 passes (tools/profile_all.sh)
# Then, in the build container: tools/profile_collect.sh rNN and the cp lines at the bottom of profiles/README.md.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" = "A" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/ev_tests.log 2>&1 || { tail -20 gpurun_out/ev_tests.log; exit 1; }
  tail -2 gpurun_out/ev_tests.log
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/ev_bench20_1.json 2> gpurun_out/ev_bench20_1.err &&
  python3 bench.py --steps 20 --warmup 5 > gpurun_out/ev_bench20_2.json 2> gpurun_out/ev_bench20_2.err &&
  python3 bench.py > gpurun_out/ev_bench_default.json 2> gpurun_out/ev_bench_default.err
elif [ "$1" = "B" ]; then
  bash tools/round_logs.sh && bash tools/solo_trace.sh > /dev/null
else   # C: the rocprofv3 databases alone come close to the 64 MiB that a call may bring back
  bash tools/profile_all.sh ${2:-r03}
fi
