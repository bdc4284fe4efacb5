#!/usr/bin/env python3
"""The reference's own criterion benches restated against the GPU path (SURVEY 8d "Micro"):
  * benches/inner_product.rs  "ipp-prover"  n = 2^1 .. 2^16: InnerProductProof::create with random generators and
    factors (the literal schedule: generators given per call, folded every round, transcript on the host);
  * benches/r1cs.rs  "prover" / "verifier"  dummy circuit of n = 2^1 .. 2^10 multiplications (benches/r1cs.rs:24-33);
  * benches/generators.rs  "BulletproofGens::new"  sizes 2 << i, i < 10 (and 2^15 for the shuffle).
Through the host C++ mirror over the C ABI; wall clock per call after one warm-up call.  The CPU column is the oracle's
single-thread restatement on the smaller sizes."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as o   # noqa: E402

host = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
label = b"test"
buf = o._buf

print("== inner_product / ipp-prover (random G, H, Q, factors; generators folded every round)")
base = o.gens("G", 4096) + o.gens("H", 4096)
for lg in range(1, 17):
    n = 1 << lg
    G = (base * ((n + 8191) // 8192))[:64 * n]
    H = (base[64 * 7:] + base[:64 * 7]) * ((n + 8191) // 8192)
    H = H[:64 * n]
    Q = o.point_mul(o.random_scalars(1, 1), o.generator())
    Gf, Hf, a, b = (o.random_scalars(10 * lg + j, n) for j in range(4))
    k = lg
    L, R, ao, bo = (C.c_uint8 * (64 * k))(), (C.c_uint8 * (64 * k))(), (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
    args = (buf(label), C.c_size_t(len(label)), C.c_size_t(n), buf(Q), buf(Gf), buf(Hf), buf(G), buf(H), buf(a), buf(b), L, R, ao, bo)
    assert host.bph_ipp_create(*args) == 0
    reps = 3 if lg <= 12 else 1
    t0 = time.perf_counter()
    for _ in range(reps):
        assert host.bph_ipp_create(*args) == 0
    tg = (time.perf_counter() - t0) / reps
    line = f"n=2^{lg:<2d} gpu {tg * 1e3:9.2f} ms"
    if lg <= 10:
        t0 = time.perf_counter()
        Lo, Ro, a_o, b_o, _ = o.ipp_create(label, n, Q, Gf, Hf, G, H, a, b)
        tc = time.perf_counter() - t0
        assert (bytes(L), bytes(R), bytes(ao), bytes(bo)) == (Lo, Ro, a_o, b_o)
        line += f"   cpu-oracle 1T {tc * 1e3:9.1f} ms  x{tc / tg:6.1f}   (bytes identical)"
    print(line)
    sys.stdout.flush()

print("== r1cs / prover, verifier (dummy circuit of n multiplications)")
for lg in range(1, 11):
    n = 1 << lg
    cap = max(n, 2)
    proof, plen, com, m = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * 64)(), C.c_size_t(0)
    vals = (C.c_uint64 * 1)(0)

    def prove():
        return host.bph_r1cs_prove(3, C.c_size_t(n), buf(label), C.c_size_t(len(label)), vals, C.c_size_t(0), C.c_uint64(77), C.c_size_t(cap),
                                   proof, C.byref(plen), com, C.byref(m))

    def verify():
        return host.bph_r1cs_verify(3, C.c_size_t(n), buf(label), C.c_size_t(len(label)), vals, C.c_size_t(0), com, C.c_size_t(1),
                                    proof, C.c_size_t(plen.value), C.c_size_t(cap), None)
    rp, rv = prove(), verify()
    assert rp == 0 and rv == 0, (lg, rp, rv)
    t0 = time.perf_counter()
    for _ in range(3):
        assert prove() == 0
    tp = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        assert verify() == 0
    tv = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    rc, proof_o, com_o = o.r1cs_prove(o.K_DUMMY, n, label, [], 77, cap)
    tc = time.perf_counter() - t0
    same = bytes(proof)[:plen.value] == proof_o
    print(f"n=2^{lg:<2d} prove {tp * 1e3:8.2f} ms  verify {tv * 1e3:8.2f} ms   cpu-oracle prove 1T {tc * 1e3:8.1f} ms   proof bytes {'identical' if same else 'DIFFER'}")
    assert same
    sys.stdout.flush()

print("== generators / BulletproofGens::new(size, 1)")
for size in [2 << i for i in range(10)] + [1 << 15]:
    out = (C.c_uint8 * (64 * size))()
    assert host.bph_gens(ord("G"), 0, C.c_size_t(size), out) == 0
    t0 = time.perf_counter()
    for _ in range(3):
        assert host.bph_gens(ord("G"), 0, C.c_size_t(size), out) == 0
    tg = (time.perf_counter() - t0) / 3
    line = f"size={size:6d} gpu {tg * 1e3:8.2f} ms (G and H chains of `size` generators each)"
    if size <= 1024:
        t0 = time.perf_counter()
        ref = o.gens("G", size)
        o.gens("H", size)
        tc = time.perf_counter() - t0
        assert bytes(out) == ref
        line += f"   cpu-oracle 1T {tc * 1e3:8.1f} ms  x{tc / tg:6.1f}   (bytes identical)"
    print(line)
