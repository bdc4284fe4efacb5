#!/bin/bash
# The secondary measurements quoted in DESIGN.md (run on the GPU box from the repo root; logs land in gpurun_out/logs_*.log and
# are copied into profiles/ by hand): MSM micro-benchmark, MPC-sized MSMs, prover batches, the 2^14 shuffle, the reference's own
# criterion benches restated, bursts of K steps.
set -e
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/bench_msm.py > gpurun_out/logs_bench_msm.log 2>&1
python3 tools/bench_mpc_msm.py > gpurun_out/logs_bench_mpc_msm.log 2>&1
python3 tools/bench_prove.py > gpurun_out/logs_bench_prove.log 2>&1
python3 tools/bench_prove_stream.py > gpurun_out/logs_bench_prove_stream.log 2>&1
python3 tools/bench_shuffle.py > gpurun_out/logs_bench_shuffle.log 2>&1
python3 tools/bench_ref_benches.py > gpurun_out/logs_bench_ref_benches.log 2>&1
WL=/tmp/bpgpu_wl_burst      # (the workload of 256 distinct batches is 0.5 GB: kept out of gpurun_out/, which is copied back)
[ -f $WL.1024 ] || python3 bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > gpurun_out/wl_burst.log 2>&1
BURST_KS=1,1,1,4,4,20,20,20,64,64,256,256,1024,1024,4096 python3 tools/burst_probe.py $WL.1024 20 > gpurun_out/logs_burst_probe.log 2>&1
python3 tools/prof_combined.py $WL.1024 20 > gpurun_out/logs_combined_bursts.log 2>&1
# control flow of the multi-rank bench on this ONE-GPU box (two ranks share device 0 over gloo; not a measurement): the weak-scaling
# verification legs, the combined check's all-gather and the sharded 2^14-shuffle leg
BPGPU_BENCH_REHEARSAL=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --steps 6 --warmup 2 --window-bits 16 --inflight 3 --distinct-batches 24 > gpurun_out/logs_rehearsal2.json 2> gpurun_out/logs_rehearsal2.err
