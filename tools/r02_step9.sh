set -e
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_multirank.py -m gpu -x -q > gpurun_out/r02_s9_tests.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r02_s9_bench20.json 2> gpurun_out/r02_s9_bench20.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prover > gpurun_out/r02_s9_bench20b.json 2> gpurun_out/r02_s9_bench20b.err
python3 bench.py --no-cpu-baseline > gpurun_out/r02_s9_bench_default.json 2> gpurun_out/r02_s9_bench_default.err
