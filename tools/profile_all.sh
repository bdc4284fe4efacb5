#!/bin/bash
# All the rocprofv3 evidence of a round (run on the GPU box from the repo root; summaries are then copied into profiles/ by
# tools/profile_collect.py):
#   1. un-profiled run that writes the workload cache (the generator forks a GPU-using child: not under a profiler)
#   2. rocprofv3 --kernel-trace --stats of the DRIVER's command (--steps 20 --warmup 5), all legs  -> gpurun_out/prof_b20/
#   3. the same of a 512-step run (steady state)                                                  -> gpurun_out/prof_final/
#   4. SQ issue / wait counters of a solo (1 step in flight) run                                  -> gpurun_out/pmc_sq/
#   5. FETCH_SIZE / WRITE_SIZE passes of the solo run                                             -> gpurun_out/pmc_FETCH_SIZE, pmc_WRITE_SIZE
set -e
R=$GRAFT_REPO_ROOT
WL=/tmp/bpgpu_wl_prof
python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_prof.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b20 $R/gpurun_out/prof_final $R/gpurun_out/pmc_sq $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
ALL="--no-cpu-baseline --no-prover --workload-cache $WL"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b20 -o b20 -- python3 $R/bench.py $ALL --steps 20 --warmup 5 > $R/gpurun_out/bench_prof_b20.json 2> $R/gpurun_out/bench_prof_b20.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o final -- python3 $R/bench.py $ALL --steps 512 --warmup 64 > $R/gpurun_out/bench_prof_final.json 2> $R/gpurun_out/bench_prof_final.err
COMMON="--no-cpu-baseline --no-combined --no-prover --workload-cache $WL"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES \
  -d $R/gpurun_out/pmc_sq -o sq -- python3 $R/bench.py $COMMON --steps 8 --warmup 2 --inflight 1 > $R/gpurun_out/pmc_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_$c -o p -- python3 $R/bench.py $COMMON --steps 8 --warmup 2 --inflight 1 > $R/gpurun_out/pmc_$c.log 2>&1
done
# 6. the prover stream (configs[2]): kernel trace with 3 worker threads and with 1; PMC passes with 1      -> gpurun_out/${TAG}_prove_*
# 7. the Pippenger pipeline at 2^17 / 2^20 terms: kernel trace + PMC                                       -> gpurun_out/${TAG}_msm_2e*
TAG=${1:-r03}
cd $R
bash tools/prof_prove_stream.sh 3 12 ${TAG}_prove_stream3 > gpurun_out/${TAG}_prove_stream3.txt 2>&1
bash tools/prof_prove_stream.sh 1 6 ${TAG}_prove_stream1 > gpurun_out/${TAG}_prove_stream1.txt 2>&1
bash tools/pmc_prove.sh ${TAG} 4 > gpurun_out/${TAG}_prove_pmc.txt 2>&1
bash tools/prof_msm_all.sh ${TAG} > gpurun_out/${TAG}_msm_profile.txt 2>&1
# 8. summaries where the databases are; the databases themselves stay on the box (a gpurun call brings back at most 64 MiB)
bash tools/profile_collect.sh ${TAG} gpurun_out/summ
rm -rf gpurun_out/prof_b20 gpurun_out/prof_final gpurun_out/pmc_sq gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/prove_pmc_* \
       gpurun_out/msm_trace_* gpurun_out/msm_pmc_* gpurun_out/${TAG}_prove_stream*_trace gpurun_out/wl_prof*
ls gpurun_out/summ
