#!/usr/bin/env python3
"""BASELINE.json configs[3]: the k-shuffle R1CS gadget (tests/r1cs.rs:23-164, benches/shuffle.rs) at k = 2^14
-> n = 32 766 multipliers (all phase 2), q = 65 533 ~ 2^16 constraints, m = 32 768 commitments, padded n = 2^15,
98 347-term mega_check MSM.  One proof, prove then verify, through the host C++ mirror over the C ABI.
Reports R1CS constraints/s for prove and for verify; checks the proof bytes against the CPU oracle at a smaller k
and has the oracle verify the big proof (--oracle-verify)."""
import argparse
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o   # noqa: E402

host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
label = b"ShuffleProofTest"
NAMES = ["generators", "tables+commitments", "prover circuit", "prove", "verifier circuit", "verify"]


def shuffle_values(k, seed):
    rnd = random.Random(seed)
    x = [rnd.getrandbits(64) for _ in range(k)]      # benches/shuffle.rs:162-167: uniform u64 inputs
    y = list(x)
    rnd.shuffle(y)
    return x + y


OS_ENTROPY = (1 << 64) - 1     # seed value that selects OsRng in the test harness


def run(k, seed=77):
    vals = shuffle_values(k, 77 if seed == OS_ENTROPY else seed)
    cap = 1
    while cap < 2 * (k - 1):
        cap *= 2
    cap = max(cap, 2)
    arr = (C.c_uint64 * len(vals))(*vals)
    proof = (C.c_uint8 * (64 * 64 + 4096))()
    plen = C.c_size_t(0)
    com = (C.c_uint8 * (2 * k * 64))()
    ms = (C.c_double * 6)()
    t0 = time.perf_counter()
    rc = host.bph_shuffle_prove_verify(C.c_size_t(k), arr, C.c_uint64(seed), C.c_size_t(cap), proof, C.byref(plen), com, ms)
    dt = time.perf_counter() - t0
    assert rc == 0, rc
    return vals, cap, bytes(proof)[:plen.value], bytes(com), list(ms), dt


ap = argparse.ArgumentParser()
ap.add_argument("--log2k", type=int, nargs="*", default=[6, 10, 14])
ap.add_argument("--oracle-verify", action="store_true", help="let the CPU oracle verify the largest proof too")
ap.add_argument("--parity-log2k", type=int, default=8, help="byte-compare the GPU proof with the oracle's at this k")
args = ap.parse_args()

run(4)    # warm-up (context, kernels)
for lg in args.log2k:
    k = 1 << lg
    runs = []
    for rep in range(4):   # first repetition untimed (workspaces grow, code is paged in); then the median of three, stage by stage
        vals, cap, proof, com, ms, dt = run(k, OS_ENTROPY)   # the default RNG (OS-keyed; blinding vectors expanded on the device), as bench.py's leg: fresh blindings, a fresh gadget challenge, no repetition finds its circuit cached
        if rep:
            runs.append(ms)
    ms = [sorted(r[i] for r in runs)[1] for i in range(6)]
    best = [min(r[i] for r in runs) for i in range(6)]
    q = 4 * (k - 1) + 1
    print(f"k=2^{lg}: n={2 * (k - 1)} q={q} m={2 * k} proof {len(proof)} B | " +
          " ".join(f"{n} {t:.1f}" for n, t in zip(NAMES, ms)) + " ms (medians of 3)")
    print(f"    prove  {ms[3]:9.1f} ms = {q / ms[3] / 1e3:8.3f} M constraints/s   (with circuit build {q / (ms[2] + ms[3]) / 1e3:.3f})")
    print(f"    verify {ms[5]:9.1f} ms = {q / ms[5] / 1e3:8.3f} M constraints/s   (with circuit build {q / (ms[4] + ms[5]) / 1e3:.3f})")
    # (the device call after ~80 ms of host-only circuit building sometimes runs 3-5x slower for its first milliseconds -- the
    # GPU's clocks have dropped; the best of the three repetitions shows the call on a warm device)
    print(f"    best of 3: prove {best[3]:.1f} ms, verify {best[5]:.1f} ms")
    sys.stdout.flush()
if args.oracle_verify:
    t0 = time.perf_counter()
    rc = o.r1cs_verify(o.K_SHUFFLE, k, label, [], com, proof, cap)
    tv = time.perf_counter() - t0
    print(f"cpu oracle verify of the k=2^{lg} GPU proof: rc={rc} in {tv:.1f} s = {q / tv / 1e3:.1f} k constraints/s (1 thread)")
    assert rc == 0
k = 1 << args.parity_log2k
vals, cap, proof, com, ms, dt = run(k)
t0 = time.perf_counter()
rc, proof_o, com_o = o.r1cs_prove(o.K_SHUFFLE, k, label, vals, 77, cap)
tc = time.perf_counter() - t0
assert rc == 0 and proof == proof_o and com == com_o, "GPU shuffle proof differs from the oracle's"
print(f"k=2^{args.parity_log2k}: GPU proof bytes identical to the oracle's (oracle prove {tc:.2f} s, 1 thread = {(4 * (k - 1) + 1) / tc / 1e3:.1f} k constraints/s)")
