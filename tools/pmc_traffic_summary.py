#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes (tools/profile_all.sh) -> per-kernel averages per launch (KB) as CSV.
Usage: pmc_traffic_summary.py gpurun_out profiles/r01"""
import csv
import sqlite3
import sys

src, dst = sys.argv[1], sys.argv[2]
res = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(f"{src}/pmc_{cn}/p_results.db")
    rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    with open(f"{dst}_pmc_{cn}.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "AvgValuePerLaunch", "Launches"])
        for r in sorted(rows, key=lambda r: -r[2]):
            w.writerow([r[0][:100], r[1], round(r[2], 1), r[3]])
    for r in rows:
        res.setdefault(r[0], {})[r[1]] = r[2]
for k, v in sorted(res.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{k[:70]:70s} FETCH_SIZE {v.get('FETCH_SIZE', 0):12.1f} KB  WRITE_SIZE {v.get('WRITE_SIZE', 0):12.1f} KB  "
          f"2F+W {(2 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) / 1024:9.1f} MB")
