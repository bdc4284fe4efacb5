R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
for t in 2 1; do
echo "== latency mode, TNP $t"
BPGPU_TABLE_NP=$t BURST_LATENCY_MODE=1 BURST_KS=1,1,1,1 python3 $R/tools/burst_probe.py $WL.1024 1 | grep K=
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/solo_trace_t1
BPGPU_TABLE_NP=1 BURST_LATENCY_MODE=1 BURST_KS=1,1,1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/solo_trace_t1 -o s -- python3 $R/tools/burst_probe.py $WL.1024 1 > $R/gpurun_out/solo_traced_t1.log 2>&1
cd $R; python3 tools/timeline.py $(find gpurun_out/solo_trace_t1 -name '*.db' | head -1) 1 | tail -7
