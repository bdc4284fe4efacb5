#!/bin/bash
# kernel timeline of one batch of 256 provers (tools/prof_prove.py) under rocprofv3 --kernel-trace; the LAST call's kernels are listed
set -e
R=$GRAFT_REPO_ROOT
NB=${1:-256}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prove_trace
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prove_trace -o p -- python3 $R/tools/prof_prove.py $NB > $R/gpurun_out/prove_traced.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob("gpurun_out/prove_trace/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the last call = everything after the last gap of more than 20 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - rows[i - 1][2] > 20_000_000: cut = i
rows = rows[cut:]
t0, t1 = rows[0][1], max(r[2] for r in rows)
busy = sum(r[2] - r[1] for r in rows)
print(f"last call: {len(rows)} kernels over {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms")
acc = {}
for n, s, e in rows:
    k = n.split("(")[0][:60]
    a = acc.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
for k, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0])[:25]:
    print(f"  {t / 1e6:8.3f} ms {c:5d}x  {k}")
# gaps > 0.3 ms between consecutive kernels (host work)
gaps = [(rows[i][1] - max(r[2] for r in rows[:i]), rows[i][0][:50]) for i in range(1, len(rows))]
big = [(g, n) for g, n in gaps if g > 300_000]
print(f"idle gaps > 0.3 ms: {len(big)}, total {sum(g for g, _ in big) / 1e6:.2f} ms")
for g, n in big[:30]: print(f"    {g / 1e6:7.2f} ms before {n}")
PY
