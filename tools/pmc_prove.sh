#!/bin/bash
# PMC passes over a stream of prover batches, one worker thread (tools/prof_prove_stream.py 1 N): SQ issue / wait counters and VALU
# wave-instructions per kernel, FETCH_SIZE / WRITE_SIZE; the timed stream = the dispatches after the driver's 0.3 s gap.
# Output: gpurun_out/${TAG}_prove_pmc_sq.txt, ${TAG}_prove_pmc.json (per-batch totals; tools/pmc_constants.py merges it)
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}; NBATCH=${2:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prove_pmc_sq $R/gpurun_out/prove_pmc_FETCH_SIZE $R/gpurun_out/prove_pmc_WRITE_SIZE
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES \
  -d $R/gpurun_out/prove_pmc_sq -o p -- python3 $R/tools/prof_prove_stream.py 1 $NBATCH > $R/gpurun_out/${TAG}_prove_pmc_sq.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/prove_pmc_$c -o p -- python3 $R/tools/prof_prove_stream.py 1 $NBATCH > $R/gpurun_out/${TAG}_prove_pmc_$c.log 2>&1
done
cd $R
python3 - $TAG $NBATCH <<'PY'
import sqlite3, glob, sys, json, collections
tag, nbatch = sys.argv[1], int(sys.argv[2])
def timed(dbpath):
    db = sqlite3.connect(dbpath)
    rows = db.execute("select kernel_name, counter_name, value, start, end, dispatch_id from counters_collection order by start").fetchall()
    starts = sorted({(r[3], r[4], r[5]) for r in rows})
    cut_t = 0
    for i in range(1, len(starts)):
        if starts[i][0] - starts[i - 1][1] > 200_000_000: cut_t = starts[i][0]
    return [r for r in rows if r[3] >= cut_t]
sq = timed(glob.glob("gpurun_out/prove_pmc_sq/**/*.db", recursive=True)[0])
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(float)
seen = set()
for k, c, v, s, e, d in sq:
    kk = k.split("(")[0]
    per[kk][c] += v
    if d not in seen:
        seen.add(d); cnt[kk] += 1; dur[kk] += e - s
out = []
tot_instr = 0
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    tot_instr += v.get("SQ_INSTS_VALU", 0)
    out.append(f"{k[:70]}\n    launches/batch {cnt[k] / nbatch:.1f}  avg {dur[k] / cnt[k] / 1e3:.0f} us  waves/launch {v.get('SQ_WAVES', 0) / cnt[k]:.0f}  "
               f"VALU wave-instr/batch {v.get('SQ_INSTS_VALU', 0) / nbatch:.3g} | waiting {100 * v.get('SQ_WAIT_ANY', 0) / wc:.1f}%  "
               f"issue-stall {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.1f}%  issuing {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.1f}%")
open(f"gpurun_out/{tag}_prove_pmc_sq.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:24]))
tr = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = timed(glob.glob(f"gpurun_out/prove_pmc_{cn}/**/*.db", recursive=True)[0])
    acc = collections.defaultdict(float); n = collections.Counter(); seen = set()
    for k, c, v, s, e, d in rows:
        if c != cn: continue
        kk = k.split("(")[0]; acc[kk] += v
        if d not in seen: seen.add(d); n[kk] += 1
    tr[cn] = {k: (acc[k], n[k]) for k in acc}
msm = [k for k in tr["FETCH_SIZE"] if "k_fixed_msm_ipp" in k]
traffic = None
if msm:
    f, nf = tr["FETCH_SIZE"][msm[0]]; w, nw = tr["WRITE_SIZE"].get(msm[0], (0, 1))
    traffic = int((2 * f / nf + w / max(nw, 1)) * 1024)
json.dump({"valu_wave_instr_per_batch": tot_instr / nbatch, "traffic_bytes_round_msm": traffic,
           "valu_wave_instr_per_batch_by_kernel": {k: v.get("SQ_INSTS_VALU", 0) / nbatch for k, v in per.items()},
           "source": f"profiles/{tag}_prove_pmc_sq.txt (tools/pmc_prove.sh: one worker thread, {nbatch} batches of 256 provers)"},
          open(f"gpurun_out/{tag}_prove_pmc.json", "w"), indent=1)
print(f"VALU wave-instructions per batch: {tot_instr / nbatch:.4g}; round-MSM traffic per launch: {traffic}")
PY
