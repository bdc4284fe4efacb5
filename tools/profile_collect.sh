#!/bin/bash
# rocprofv3 databases of tools/profile_all.sh -> summaries (CSV / text / pmc_constants.json).  Two uses:
#   on the GPU box, at the end of profile_all.sh:  profile_collect.sh rNN gpurun_out/summ   (the databases exceed what a gpurun call
#     brings back, so they are summarised there and deleted);
#   in the build container:                       profile_collect.sh rNN                    copies gpurun_out/summ/* into profiles/ (tracked)
set -e
cd "$(dirname "$0")/.."
T=${1:-r03}
OUT=${2:-}
if [ -z "$OUT" ]; then
  cp gpurun_out/summ/${T}_* profiles/
  cp gpurun_out/summ/pmc_constants.json profiles/pmc_constants.json
  ls -la profiles/${T}_* profiles/pmc_constants.json
  exit 0
fi
mkdir -p $OUT
python3 tools/kernel_stats.py gpurun_out/prof_b20/b20_results.db $OUT/${T}_bench20_kernel_stats.csv > /dev/null
python3 tools/kernel_stats.py gpurun_out/prof_final/final_results.db $OUT/${T}_bench512_kernel_stats.csv > /dev/null
tail -n 1 gpurun_out/bench_prof_b20.json > $OUT/${T}_bench20_profiled_run.json
tail -n 1 gpurun_out/bench_prof_final.json > $OUT/${T}_bench512_profiled_run.json
python3 tools/pmc_sq_summary.py gpurun_out/pmc_sq/sq_results.db $OUT/${T}_pmc_sq_summary.txt > /dev/null
python3 tools/pmc_traffic_summary.py gpurun_out $OUT/${T} > /dev/null
for f in prove_stream3.txt prove_stream1.txt prove_stream3_kernel_stats.csv prove_stream1_kernel_stats.csv prove_pmc_sq.txt prove_pmc.json \
         msm_2e17_kernel_stats.csv msm_2e20_kernel_stats.csv msm_2e17_pmc_sq.txt msm_2e20_pmc_sq.txt msm_profile.txt; do
  [ -f gpurun_out/${T}_$f ] && cp gpurun_out/${T}_$f $OUT/${T}_$f
done
python3 tools/pmc_constants.py gpurun_out $OUT/${T} > /dev/null
