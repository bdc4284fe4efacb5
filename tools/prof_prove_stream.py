#!/usr/bin/env python3
"""Driver of tools/prof_prove_stream.sh: warm-up, a 0.3 s idle gap (the trace is cut there), then ONE stream of NBATCH batches of
256 provers on THREADS worker threads with the constraint systems prebuilt.  Usage: prof_prove_stream.py [threads] [nbatch]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nbatch = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nb, nvals, n_bits = 256, 16, 64
n, q = nvals * n_bits, nvals * (2 * n_bits + 1)
label = b"RangeProofTest"
lab = (C.c_uint8 * len(label)).from_buffer_copy(label)
vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
arr = (C.c_uint64 * len(vals))(*vals)


def run(nbatch, threads):
    proofs, plen = (C.c_uint8 * (nbatch * nb * 4096))(), C.c_size_t(0)
    com, ms = (C.c_uint8 * (nbatch * nb * nvals * 64))(), (C.c_double * 12)()
    rc = host.bph_range_prove_stream(C.c_size_t(nbatch), C.c_size_t(threads), C.c_int(1), C.c_int(0), C.c_size_t(nb), C.c_size_t(nvals),
                                     C.c_size_t(n_bits), lab, C.c_size_t(len(label)), arr, C.c_uint64((1 << 64) - 1), C.c_size_t(n), proofs,
                                     C.byref(plen), com, ms)
    assert rc == 0, rc
    return list(ms)


run(2, 1)
run(2 * threads, threads)
time.sleep(0.3)
ms = run(nbatch, threads)
print(f"threads={threads} nbatch={nbatch}: wall {ms[0]:.2f} ms = {ms[0] / nbatch:.2f} ms/batch = {nbatch * nb * q / ms[0] / 1e3:.2f} M constraints/s")
