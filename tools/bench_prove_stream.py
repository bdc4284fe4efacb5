#!/usr/bin/env python3
"""Prover STREAM throughput (BASELINE.json configs[2] shape: 256 provers x (16 x 64-bit range gadgets in one constraint system),
q = 2064 constraints per prover): batches proved by worker threads that each own a context, OsRng blinding factors (blinding
vectors drawn on the device).  Sweeps the number of worker threads, with the constraint systems prebuilt (the reference's bench
definition: only Prover::prove is timed, benches/r1cs.rs:36-55) and with circuit building inside the timed region.
BPH_TIMING=1 prints the phase laps of every prove_batch call."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
nb, nvals, n_bits = int(os.environ.get("NB", "256")), 16, 64
n, q = nvals * n_bits, nvals * (2 * n_bits + 1)
label = b"RangeProofTest"
lab = (C.c_uint8 * len(label)).from_buffer_copy(label)
vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
arr = (C.c_uint64 * len(vals))(*vals)
OS = (1 << 64) - 1


def run(nbatch, threads, prebuild):
    proofs, plen = (C.c_uint8 * (nbatch * nb * 4096))(), C.c_size_t(0)
    com, ms = (C.c_uint8 * (nbatch * nb * nvals * 64))(), (C.c_double * 12)()
    rc = host.bph_range_prove_stream(C.c_size_t(nbatch), C.c_size_t(threads), C.c_int(prebuild), C.c_int(0), C.c_size_t(nb), C.c_size_t(nvals),
                                     C.c_size_t(n_bits), lab, C.c_size_t(len(label)), arr, C.c_uint64(OS), C.c_size_t(n), proofs,
                                     C.byref(plen), com, ms)
    assert rc == 0, rc
    return list(ms)


run(2, 1, 1)      # generator tables, workspaces, pools
for prebuild in (() if os.environ.get("PROFILE_ROUND_MSM") else (1, 0)):
    for threads in (1, 2, 3, 4):
        nbatch = 12
        run(threads, threads, prebuild)   # this thread count's contexts warm
        best = None
        for rep in range(3):
            ms = run(nbatch, threads, prebuild)
            if best is None or ms[0] < best[0]:
                best = ms
        wall, build, prove, drop = best[:4]
        print(f"prebuild={prebuild} threads={threads}: {wall / nbatch:7.2f} ms/batch = {nbatch * nb * q / wall / 1e3:7.2f} M constraints/s "
              f"| per batch: build {build / nbatch:6.2f}  prove_batch {prove / nbatch:6.2f}  drop {drop / nbatch:5.2f} ms", flush=True)


def run_profiled(nbatch, threads):
    proofs, plen = (C.c_uint8 * (nbatch * nb * 4096))(), C.c_size_t(0)
    com, ms = (C.c_uint8 * (nbatch * nb * nvals * 64))(), (C.c_double * 12)()
    rc = host.bph_range_prove_stream(C.c_size_t(nbatch), C.c_size_t(threads), C.c_int(1), C.c_int(1), C.c_size_t(nb), C.c_size_t(nvals),
                                     C.c_size_t(n_bits), lab, C.c_size_t(len(label)), arr, C.c_uint64(OS), C.c_size_t(n), proofs,
                                     C.byref(plen), com, ms)
    assert rc == 0, rc
    return list(ms)


if os.environ.get("PROFILE_ROUND_MSM"):      # HIP-event duration of the round MSM (k_fixed_msm_ipp* + its partial sums), one worker thread
    run_profiled(2, 1)
    ms = run_profiled(4, 1)
    print(f"round MSM (L / R table walks of one IPP round, {2 * nb} MSMs of {n + 1} terms): {ms[5] / max(ms[6], 1):.3f} ms per launch over {int(ms[6])} launches; "
          f"device phases per batch: rounds {ms[7] / 4:.2f}  commitments {ms[8] / 4:.2f}  polys {ms[9] / 4:.2f}  T {ms[10] / 4:.2f}  ipp setup {ms[11] / 4:.2f} ms")
