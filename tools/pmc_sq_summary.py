#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_* pass (tools/profile_all.sh): per kernel wave-instruction counts and the split of wave
cycles into waiting / issuing.  Usage: pmc_sq_summary.py gpurun_out/pmc_sq/sq_results.db [out.txt]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, avg(value), count(*), avg(duration), max(vgpr_count), max(lds_block_size), "
                  "max(scratch_size), max(grid_size), max(workgroup_size) from counters_collection group by kernel_name, counter_name").fetchall()
d, meta = collections.defaultdict(dict), {}
for r in rows:
    d[r[0]][r[1]] = r[2]
    meta[r[0]] = r[3:]
out = []
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e5:
        continue
    m = meta[k]
    out.append(f"{k[:70]}\n    launches {m[0]} avg {m[1] / 1e3:.0f} us  vgpr {m[2]} lds {m[3]} scratch {m[4]} grid {m[5]} wg {m[6]}\n"
               f"    waves {v['SQ_WAVES']:.0f}  VALU wave-instr {v['SQ_INSTS_VALU']:.3g}  wave quad-cycles {wc:.3g} | waiting {100 * v['SQ_WAIT_ANY'] / wc:.1f}%  "
               f"issue-stall {100 * v['SQ_WAIT_INST_ANY'] / wc:.1f}%  issuing {100 * v['SQ_ACTIVE_INST_ANY'] / wc:.1f}% (VALU {100 * v['SQ_ACTIVE_INST_VALU'] / wc:.1f}%)")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
