#!/bin/bash
# rocprofv3 --kernel-trace of a stream of prover batches (tools/prof_prove_stream.py): per-kernel time per batch, GPU-busy share of
# the stream's wall clock.  Usage: prof_prove_stream.sh [threads] [nbatch] [tag]
set -e
R=$GRAFT_REPO_ROOT
T=${1:-3}; NBATCH=${2:-12}; TAG=${3:-prove_stream}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_trace
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/${TAG}_trace -o p -- python3 $R/tools/prof_prove_stream.py $T $NBATCH > $R/gpurun_out/${TAG}_traced.log 2>&1
cd $R
tail -1 gpurun_out/${TAG}_traced.log
python3 - $NBATCH $TAG <<'PY'
import sqlite3, glob, sys, csv
nbatch, tag = int(sys.argv[1]), sys.argv[2]
db = sqlite3.connect(glob.glob(f"gpurun_out/{tag}_trace/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - max(r[2] for r in rows[max(0, i - 50):i]) > 200_000_000: cut = i     # the 0.3 s sleep
rows = rows[cut:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for n, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(r[2] - r[1] for r in rows)
print(f"timed stream: {len(rows)} kernels ({len(rows) / nbatch:.0f} per batch), first launch to last completion {(t1 - t0) / 1e6:.2f} ms "
      f"= {(t1 - t0) / 1e6 / nbatch:.2f} ms/batch; GPU busy (union of kernel intervals) {busy / 1e6:.2f} ms = {100 * busy / (t1 - t0):.1f} %; "
      f"sum of kernel durations {ksum / 1e6:.2f} ms = {ksum / 1e6 / nbatch:.2f} ms/batch")
acc = {}
for n, s, e in rows:
    k = n.split("(")[0][:70]
    a = acc.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
with open(f"gpurun_out/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerBatch", "MsPerBatch", "AverageUs", "Percentage"])
    for k, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0]):
        w.writerow([k, round(c / nbatch, 2), round(t / 1e6 / nbatch, 4), round(t / c / 1e3, 2), round(100 * t / ksum, 2)])
for k, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0])[:22]:
    print(f"  {t / 1e6 / nbatch:8.3f} ms/batch {c / nbatch:6.1f}x  avg {t / c / 1e3:8.1f} us  {k}")
PY
