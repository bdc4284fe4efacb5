R=$GRAFT_REPO_ROOT
WL=$R/gpurun_out/wl_burst
[ -f $WL.1024 ] || python3 $R/bench.py --no-cpu-baseline --no-combined --no-prover --steps 4 --warmup 2 --workload-cache $WL > $R/gpurun_out/wl_burst.log 2>&1
for lpm in 32 64; do for q in 1 0; do
echo "== latency mode, LPM $lpm quad $q"
BPGPU_FIXED_LPM=$lpm BPGPU_HORNER_QUAD=$q BURST_LATENCY_MODE=1 BURST_KS=1,1,1,1 python3 $R/tools/burst_probe.py $WL.1024 1 | grep K=
done; done
