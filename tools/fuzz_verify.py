#!/usr/bin/env python3
"""Differential fuzz of the whole Verifier::verify: the GPU (transcript replay on the device + verification,
bpgpu_r1cs_verify_batch_fs) against the CPU oracle, proof by proof, on proofs with random byte flips ANYWHERE in the proof or
the commitment -- points knocked off the curve, non-canonical coordinates and scalars, identity points, flipped challenge
material.  Every accept bit must equal the oracle's verdict.  Imports the oracle through tests/oracle_lib.py (checker).
Usage: fuzz_verify.py [proofs per size] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402,F401
import mpc_bulletproof_amd as m   # noqa: E402
import bp_helpers as bh   # noqa: E402
import oracle_lib as o   # noqa: E402
import pymodel as pm   # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
LABEL = b"RangeProofTest"
gpu = m.BpGpu(0)
total = rejected = malformed_like = 0
t0 = time.time()
for n_bits in (8, 16, 32, 64):
    cap = 1 << max(0, (n_bits - 1).bit_length())
    recs = []
    for i in range(nb):
        v = rnd.getrandbits(n_bits)
        rc, proof, com = o.r1cs_prove(o.K_RANGE, n_bits, LABEL, [v], rnd.getrandbits(40), cap)
        assert rc == 0
        proof, com = bytearray(proof), bytearray(com)
        mode = rnd.randrange(8)
        if mode >= 2:                                 # 75 %: one to three flips
            for _ in range(rnd.choice((1, 1, 2, 3))):
                if mode == 7:                          # in the commitment
                    com[rnd.randrange(len(com))] ^= 1 << rnd.randrange(8)
                elif mode == 6:                        # a whole point -> identity
                    slot = rnd.randrange(11)
                    proof[8 + 64 * slot:8 + 64 * slot + 64] = bytes(64)
                elif mode == 5:                        # top bits of a coordinate / scalar (non-canonical values)
                    pos = 8 + 32 * rnd.randrange((len(proof) - 8) // 32) + 31
                    proof[pos] |= 0xF0
                else:                                  # anywhere after the 8-byte header
                    proof[8 + rnd.randrange(len(proof) - 8)] ^= 1 << rnd.randrange(8)
        recs.append((bytes(proof), bytes(com)))
    s0 = o.VerifySession(o.K_RANGE, n_bits, LABEL, [], *reversed(o.r1cs_prove(o.K_RANGE, n_bits, LABEL, [1], 7, cap)[1:]), cap)
    circ = gpu.circuit_create(*s0.csr(), s0.n1 + s0.n2, s0.m)
    g = gpu.gens_create(o.gens("G", cap), o.gens("H", cap), o.generator(), o.generator(), 8)
    init = pm.Transcript(LABEL).state * nb
    pts = sc = b""
    for proof, com in recs:
        k, p, q = bh.verify_inputs(proof, com)
        pts, sc = pts + p, sc + q
    ok, _, _ = gpu.r1cs_verify_batch_fs(g, circ, nb, s0.n1, s0.k, s0.m, init, pts, sc, want_mega=False)
    for i, (proof, com) in enumerate(recs):
        rc = o.r1cs_verify(o.K_RANGE, n_bits, LABEL, [], com, proof, cap)
        want = 1 if rc == 0 else 0
        assert ok[i] == want, f"n_bits={n_bits} proof {i}: GPU accept bit {ok[i]}, oracle rc {rc}"
        total += 1
        rejected += 1 - want
    gpu.gens_destroy(g)
    gpu.circuit_destroy(circ)
    s0.close()
    print(f"n_bits={n_bits:2d}: {nb} proofs, verdicts identical ({time.time() - t0:.0f} s)", flush=True)
print(f"fuzz_verify: {total} proofs, {rejected} rejected by both, {total - rejected} accepted by both; no disagreement")
