#!/usr/bin/env python3
"""Occupancy picture of the last burst in a rocprofv3 --kernel-trace database (tools/burst_trace.sh): per time bin, how many launches
of each kernel of the verification chain are on the chip and how many waves they bring, against the 1 024 SIMDs.
Usage: burst_gantt.py results.db [bin_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
binus = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows = db.execute("select name, start, end from kernels order by start").fetchall()
bursts, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[1] - last_end > 2_000_000:
        bursts.append(cur)
        cur = []
    cur.append(r)
    last_end = max(last_end or 0, r[2])
bursts.append(cur)
b = bursts[-1]
WAVES = {"front": 112, "scalars": 1024, "windows": 1024, "groups": 128, "back": 320, "verdict": 16}


def kind(n):
    for k in WAVES:
        if "k_verify_" + k in n or ("horner_groups" in n and k == "groups"):
            return k
    return None


t0 = b[0][1]
span = max(r[2] for r in b) - t0
print(f"burst of {len(b)} kernels, span {span / 1e3:.1f} us")
nbin = int(span / 1e3 / binus) + 1
print("   t(us)  " + " ".join(f"{k:>8s}" for k in WAVES) + "   waves/SIMD (launched, not resident)")
for i in range(nbin):
    lo, hi = t0 + i * binus * 1e3, t0 + (i + 1) * binus * 1e3
    cnt = {k: 0.0 for k in WAVES}
    for n, s, e in b:
        k = kind(n)
        if k is None:
            continue
        ov = min(e, hi) - max(s, lo)
        if ov > 0:
            cnt[k] += ov / (hi - lo)
    w = sum(cnt[k] * WAVES[k] for k in WAVES)
    print(f"{i * binus:8.0f}  " + " ".join(f"{cnt[k]:8.1f}" for k in WAVES) + f"   {w / 1024:6.2f}")
for k in WAVES:
    d = [(e - s) / 1e3 for n, s, e in b if kind(n) == k]
    if d:
        st = [(s - t0) / 1e3 for n, s, e in b if kind(n) == k]
        en = [(e - t0) / 1e3 for n, s, e in b if kind(n) == k]
        print(f"{k:8s} x{len(d):3d}: duration min {min(d):7.1f} avg {sum(d) / len(d):7.1f} max {max(d):7.1f} us; first start {min(st):7.1f} last start {max(st):7.1f} first end {min(en):7.1f} last end {max(en):7.1f}")
