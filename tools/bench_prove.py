#!/usr/bin/env python3
"""Prover throughput (BASELINE.json configs[2] shape): nb provers in lock-step, each range-proving 16 values of
64 bits in one constraint system (n = 1024 multipliers, q = 2064 constraints, m = 16), through the host C++
mirror (Prover::prove_batch) over the C ABI.  Reports proofs/s and R1CS constraints/s (wall clock incl. the host
transcripts and packing), and the CPU oracle's single-thread prove time for the same circuit."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o   # noqa: E402

host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
nvals, n_bits = 16, 64
n = nvals * n_bits
q = nvals * (2 * n_bits + 1)
label = b"RangeProofTest"


def run(nb, seed0=900):
    vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
    arr = (C.c_uint64 * len(vals))(*vals)
    plen = C.c_size_t(0)
    proofs = (C.c_uint8 * (nb * 4096))()
    com = (C.c_uint8 * (nb * nvals * 64))()
    t0 = time.perf_counter()
    rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(nvals), C.c_size_t(n_bits), o._buf(label), C.c_size_t(len(label)),
                                    arr, C.c_uint64(seed0), C.c_size_t(n), proofs, C.byref(plen), com)
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    return dt, vals, bytes(proofs)[:plen.value * nb], bytes(com), plen.value


run(2)   # warm-up: generator tables, workspaces
for nb in (1, 16, 64, 256):
    run(nb)   # grows the workspaces for this batch size
    dt, vals, proofs, com, L = run(nb)
    print(f"nb={nb:4d}: {dt * 1e3:9.1f} ms  {nb / dt:9.1f} proofs/s  {nb * q / dt / 1e6:8.3f} M constraints/s  ({q} constraints, n={n}, proof {L} B)")
# parity + CPU baseline on one prover
t0 = time.perf_counter()
rc, proof_o, com_o = o.r1cs_prove(o.K_RANGE_MULTI, n_bits | (nvals << 16), label, vals[:nvals], 900, n)
tc = time.perf_counter() - t0
assert proofs[:L] == proof_o and com[:nvals * 64] == com_o, "GPU prover disagrees with the oracle"
print(f"cpu oracle (1 thread): {tc * 1e3:.0f} ms per proof = {q / tc / 1e3:.1f} k constraints/s ; GPU proof bytes identical")
