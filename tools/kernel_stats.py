#!/usr/bin/env python3
"""rocprofv3 --kernel-trace result database -> per-kernel stats CSV (what `--stats` prints).
Usage: kernel_stats.py gpurun_out/prof_final/final_results.db profiles/r01_bench_final_kernel_stats.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0][:110], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 2), r[4], r[5]])
for r in rows[:12]:
    print(f"{r[0][:70]:70s} calls {r[1]:5d} avg {r[3] / 1e3:9.1f} us  {100 * r[2] / tot:5.1f}%")
