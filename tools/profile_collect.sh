#!/bin/bash
# gpurun_out/ (scratch, merged back from the GPU box after tools/profile_all.sh) -> profiles/ (tracked).  Run in the build container.
set -e
cd "$(dirname "$0")/.."
T=${1:-r03}
python3 tools/kernel_stats.py gpurun_out/prof_b20/b20_results.db profiles/${T}_bench20_kernel_stats.csv > /dev/null
python3 tools/kernel_stats.py gpurun_out/prof_final/final_results.db profiles/${T}_bench512_kernel_stats.csv > /dev/null
tail -n 1 gpurun_out/bench_prof_b20.json > profiles/${T}_bench20_profiled_run.json
tail -n 1 gpurun_out/bench_prof_final.json > profiles/${T}_bench512_profiled_run.json
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq/sq_results.db profiles/${T}_pmc_sq_summary.txt > /dev/null
python3 tools/pmc_traffic_summary.py gpurun_out profiles/${T} > /dev/null
python3 tools/pmc_constants.py gpurun_out profiles/${T} > /dev/null
for f in prove_stream3.txt prove_stream1.txt prove_stream3_kernel_stats.csv prove_stream1_kernel_stats.csv prove_pmc_sq.txt \
         msm_2e17_kernel_stats.csv msm_2e20_kernel_stats.csv msm_2e17_pmc_sq.txt msm_2e20_pmc_sq.txt msm_profile.txt; do
  [ -f gpurun_out/${T}_$f ] && cp gpurun_out/${T}_$f profiles/${T}_$f
done
ls -la profiles/${T}_* profiles/pmc_constants.json
