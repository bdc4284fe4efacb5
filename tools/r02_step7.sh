set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or combined or pippenger or full_size or prover_polys" > gpurun_out/r02_s7_tests.log 2>&1
python3 tools/bench_msm.py > gpurun_out/r02_s7_bench_msm.log 2>&1
bash tools/prof_msm.sh 14
WL=$R/gpurun_out/wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
python3 $R/tools/prof_combined.py $WL.1024 20 > $R/gpurun_out/comb_plain2.log 2>&1
