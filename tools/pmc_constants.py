#!/usr/bin/env python3
"""PMC passes of tools/profile_all.sh -> profiles/pmc_constants.json: VALU wave-instructions per 1024-proof step and HBM traffic per
launch of the kernels of the verification chain, tagged with the hash of the libbpgpu.so they were measured on (bench.py prints
them with `binary_matches`).  Usage: pmc_constants.py gpurun_out profiles/r02"""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
OUT = os.path.dirname(tag) or os.path.join(ROOT, "profiles")      # summaries live beside the tag prefix (profiles/rNN or gpurun_out/summ/rNN)
KERNELS = {"verify_front": "k_verify_front", "verify_scalars": "k_verify_scalars", "verify_windows": "k_verify_windows",
           "verify_groups": "k_verify_horner_groups", "verify_back": "k_verify_back", "verify_verdict": "k_verify_verdict"}


def per_kernel(dbpath, counter, col="value"):
    db = sqlite3.connect(dbpath)
    rows = db.execute(f"select kernel_name, avg({col}), count(*) from counters_collection where counter_name = ? group by kernel_name",
                      (counter,)).fetchall()
    out = {}
    for short, needle in KERNELS.items():
        for name, v, cnt in rows:
            if needle in name:
                out[short] = (v, cnt)
    return out


valu = per_kernel(f"{src}/pmc_sq/sq_results.db", "SQ_INSTS_VALU")
waves = per_kernel(f"{src}/pmc_sq/sq_results.db", "SQ_WAVES")
solo = per_kernel(f"{src}/pmc_sq/sq_results.db", "SQ_INSTS_VALU", "duration")       # ns per launch of the solo (1 step in flight) run
fetch = per_kernel(f"{src}/pmc_FETCH_SIZE/p_results.db", "FETCH_SIZE")
write = per_kernel(f"{src}/pmc_WRITE_SIZE/p_results.db", "WRITE_SIZE")
h = hashlib.sha256()
with open(os.path.join(ROOT, "mpc_bulletproof_amd", "libbpgpu.so"), "rb") as f:
    h.update(f.read())
out = {
    "lib_sha256_16": h.hexdigest()[:16],
    "source": f"{tag}_pmc_sq_summary.txt, {tag}_pmc_FETCH_SIZE.csv, {tag}_pmc_WRITE_SIZE.csv (tools/profile_all.sh: solo run, 1 step in flight, batch 1024, c = 20)",
    "valu_wave_instr_per_launch": {k: v[0] for k, v in valu.items()},
    "valu_wave_instr_per_step_1024": sum(v[0] for v in valu.values()),
    "waves_per_launch": {k: v[0] for k, v in waves.items()},
    "solo_us": {k: v[0] / 1e3 for k, v in solo.items()},
    # HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B request)
    "traffic_bytes_per_launch": {k: int((2 * fetch[k][0] + write.get(k, (0, 0))[0]) * 1024) for k in fetch},
}
# the profiled run of the driver's command (tools/profile_all.sh): the dominant kernel's average duration by rocprofv3 and by the
# HIP events of that same run.  Under the tracer the steps in flight overlap less than in an un-profiled run, so a kernel launch
# is SHORTER there (fewer neighbours on the chip): bench.py quotes these beside its own event average.
try:
    import csv
    run = json.loads(open(os.path.join(OUT, f"{os.path.basename(tag)}_bench20_profiled_run.json")).read())
    roof = run["roofline"]
    kern = roof["kernel"].split("<")[0]
    for row in csv.DictReader(open(os.path.join(OUT, f"{os.path.basename(tag)}_bench20_kernel_stats.csv"))):
        if kern in row["Name"]:
            out["profiled_run_bench20"] = {"kernel": roof["kernel"], "rocprofv3_avg_ms": float(row["AverageNs"]) / 1e6, "rocprofv3_calls": int(row["Calls"]),
                                           "hip_events_avg_ms": roof["avg_launch_ms"], "hip_events_launches": roof["launches"],
                                           "source": f"profiles/{os.path.basename(tag)}_bench20_kernel_stats.csv, profiles/{os.path.basename(tag)}_bench20_profiled_run.json"}
            break
except (OSError, KeyError, ValueError) as e:
    print("no profiled-run comparison:", e, file=sys.stderr)
# the prover stream's counters (tools/pmc_prove.sh), when that pass has been run
try:
    out["prover"] = json.loads(open(f"{src}/{os.path.basename(tag)}_prove_pmc.json").read())
except OSError as e:
    print("no prover counters:", e, file=sys.stderr)
path = os.path.join(OUT, "pmc_constants.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
