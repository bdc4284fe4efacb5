#!/usr/bin/env python3
"""Timeline of the last kernels in a rocprofv3 --kernel-trace database: start/end relative to the first kernel after the
last gap of more than GAP_US, per queue.  Usage: timeline.py results.db [n_last_bursts]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
# split into bursts at idle gaps > 2 ms
bursts, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[1] - last_end > 2_000_000:
        bursts.append(cur)
        cur = []
    cur.append(r)
    last_end = max(last_end or 0, r[2])
bursts.append(cur)
for b in bursts[-nb:]:
    t0 = b[0][1]
    print(f"--- burst of {len(b)} kernels, span {(max(r[2] for r in b) - t0) / 1e3:.1f} us")
    for r in b:
        print(f"  q{r[3]:<6} {(r[1] - t0) / 1e3:9.1f} -> {(r[2] - t0) / 1e3:9.1f} us ({(r[2] - r[1]) / 1e3:8.1f})  {r[0][:60]}")
