#!/bin/bash
# The Pippenger pipeline (k_pip.hip) at 2^17 and 2^20 terms: rocprofv3 kernel trace -> per-kernel stats of the timed calls, and a PMC
# pass (SQ issue / wait counters) -> waiting share per kernel.  Output: gpurun_out/${TAG}_msm_2e{17,20}_kernel_stats.csv, _pmc_sq.txt
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
for LG in 17 20; do
  rm -rf $R/gpurun_out/msm_trace_$LG $R/gpurun_out/msm_pmc_$LG
  timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/msm_trace_$LG -o m -- python3 $R/tools/prof_msm2.py $LG 5 > $R/gpurun_out/${TAG}_msm_2e${LG}_traced.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES \
    -d $R/gpurun_out/msm_pmc_$LG -o p -- python3 $R/tools/prof_msm2.py $LG 2 > $R/gpurun_out/${TAG}_msm_2e${LG}_pmc.log 2>&1
done
cd $R
for LG in 17 20; do
python3 - $LG $TAG <<'PY'
import sqlite3, glob, sys, csv
lg, tag = sys.argv[1], sys.argv[2]
db = sqlite3.connect(glob.glob(f"gpurun_out/msm_trace_{lg}/**/*.db", recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - rows[i - 1][2] > 200_000_000: cut = i
rows = rows[cut:]
reps = 5
acc = {}
for n, s, e in rows:
    k = n.split("(")[0][:80]
    a = acc.setdefault(k, [0, 0]); a[0] += e - s; a[1] += 1
tot = sum(v[0] for v in acc.values())
span = (max(r[2] for r in rows) - rows[0][1]) / reps
with open(f"gpurun_out/{tag}_msm_2e{lg}_kernel_stats.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerMsm", "UsPerMsm", "AverageUs", "Percentage"])
    for k, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0]):
        w.writerow([k, round(c / reps, 2), round(t / 1e3 / reps, 2), round(t / c / 1e3, 2), round(100 * t / tot, 2)])
print(f"2^{lg}: {len(rows) / reps:.0f} kernels per MSM, {span / 1e6:.3f} ms first launch to last completion, sum of kernels {tot / 1e6 / reps:.3f} ms")
for k, (t, c) in sorted(acc.items(), key=lambda x: -x[1][0])[:14]:
    print(f"  {t / 1e3 / reps:8.1f} us {c / reps:5.1f}x  {k}")
PY
python3 tools/pmc_sq_summary.py $(find gpurun_out/msm_pmc_$LG -name '*.db' | head -1) gpurun_out/${TAG}_msm_2e${LG}_pmc_sq.txt > /dev/null
grep -A2 "k_pip_bucket\|k_pip_merge\|k_pip_window\|k_pip_fine\|k_pip_coarse" gpurun_out/${TAG}_msm_2e${LG}_pmc_sq.txt | head -40
tail -1 gpurun_out/${TAG}_msm_2e${LG}_traced.log
done
