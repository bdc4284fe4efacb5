"""GPU tests of the host-side C++ mirror of the reference API (libbphost.so over libbpgpu.so), driven through the flat
harness tests/host/libbph_capi.so: its proofs are byte-identical to the CPU oracle's on the same seeds, and the reference's
own tests (restated in tests/host/host_tests.cpp) pass."""
import ctypes as C
import os
import subprocess

import pytest

import oracle_lib as o

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = bytes.fromhex


@pytest.fixture(scope="module")
def host():
    lib = C.CDLL(os.path.join(ROOT, "tests", "host", "libbph_capi.so"))
    return lib


def _prove(host, kind, param, label, values, seed, cap):
    vals = (C.c_uint64 * max(len(values), 1))(*values)
    proof = (C.c_uint8 * 8192)()
    plen, m = C.c_size_t(0), C.c_size_t(0)
    com = (C.c_uint8 * (64 * max(1, 2 * param if kind == o.K_SHUFFLE else param >> 16, 5)))()
    rc = host.bph_r1cs_prove(kind, C.c_size_t(param), o._buf(label), C.c_size_t(len(label)), vals, C.c_size_t(len(values)),
                             C.c_uint64(seed), C.c_size_t(cap), proof, C.byref(plen), com, C.byref(m))
    return rc, bytes(proof)[:plen.value], bytes(com)[:64 * m.value]


def _verify(host, kind, param, label, values, com, proof, cap):
    vals = (C.c_uint64 * max(len(values), 1))(*values)
    mega = (C.c_uint8 * 64)()
    rc = host.bph_r1cs_verify(kind, C.c_size_t(param), o._buf(label), C.c_size_t(len(label)), vals, C.c_size_t(len(values)),
                              o._buf(com), C.c_size_t(len(com) // 64), o._buf(proof), C.c_size_t(len(proof)), C.c_size_t(cap), mega)
    return rc, bytes(mega)


def test_generators_match_oracle(host):
    for which in "GH":
        out = (C.c_uint8 * (64 * 40))()
        assert host.bph_gens(ord(which), 0, C.c_size_t(40), out) == 0
        assert bytes(out) == o.gens(which, 40)


def test_generator_cache_on_disk(host, tmp_path, monkeypatch):
    """BPH_GENS_CACHE_DIR: the chains' points are written once, served from the file afterwards (validated by
    the hash-chain state stored with every record), extended when more are asked for and repaired when a record is stale."""
    monkeypatch.setenv("BPH_GENS_CACHE_DIR", str(tmp_path))
    want = o.gens("G", 48), o.gens("H", 48)

    def get(which, n):
        out = (C.c_uint8 * (64 * n))()
        assert host.bph_gens(ord(which), 0, C.c_size_t(n), out) == 0
        return bytes(out)

    assert get("G", 20) == want[0][:64 * 20]
    f = tmp_path / "gens_G0.bin"
    assert f.stat().st_size == 96 * 20 and (tmp_path / "gens_H0.bin").stat().st_size == 96 * 20
    recs = f.read_bytes()
    assert all(recs[96 * i + 32:96 * i + 96] == want[0][64 * i:64 * i + 64] for i in range(20))
    assert get("G", 20) == want[0][:64 * 20] and get("H", 48) == want[1]            # served; extended
    assert f.stat().st_size == 96 * 48
    bad = bytearray(f.read_bytes())
    bad[96 * 5 + 40] ^= 1                 # a corrupted POINT behind a valid state would be served: states are the check ...
    bad[96 * 9] ^= 1                      # ... so corrupt a state: everything from record 9 on is recomputed and rewritten
    f.write_bytes(bytes(bad[:96 * 5 + 40]) + bytes([bad[96 * 5 + 40] ^ 1]) + bytes(bad[96 * 5 + 41:]))
    assert get("G", 48) == want[0]
    assert f.read_bytes()[96 * 9:96 * 10 + 96] == recs[96 * 9:96 * 10 + 96]


def test_prover_bytes_identical_to_oracle(host, golden_r1cs):
    """Prover::prove on the GPU reproduces the oracle's proof bytes (same transcript, same RNG)."""
    for rec in golden_r1cs["range"]:
        rc, proof, com = _prove(host, o.K_RANGE, rec["n_bits"], H(rec["label"]), [rec["v"]], rec["seed"], 16)
        assert rc == 0 and com == b"".join(map(H, rec["commitments"])) and proof == H(rec["proof"])
    for rec in golden_r1cs["shuffle"]:
        rc, proof, com = _prove(host, o.K_SHUFFLE, rec["k"], H(rec["label"]), rec["values"], rec["seed"], 16)
        assert rc == 0 and proof == H(rec["proof"])
    for rec in golden_r1cs["example"]:
        rc, proof, com = _prove(host, o.K_EXAMPLE, 0, H(rec["label"]), rec["values"], rec["seed"], 16)
        assert rc == 0 and proof == H(rec["proof"])


def test_prover_bytes_identical_to_oracle_on_random_circuits(host):
    """The same byte equality away from the golden sizes: range gadgets of random widths (incl. widths that are not powers of two:
    padded n), several values in one constraint system, k-shuffles of random k, dummy circuits, random values, seeds and labels,
    in both blinding modes; the GPU verifier accepts each proof and rejects it after a one-bit flip, as the oracle does."""
    import random
    rnd = random.Random(20261004)
    cases = [(o.K_RANGE, nb_, [rnd.getrandbits(nb_)]) for nb_ in (1, 3, 7, 13, 24, 33, 47, 64)]
    cases += [(o.K_RANGE_MULTI, nb_ | (nv << 16), [rnd.getrandbits(nb_) for _ in range(nv)]) for nb_, nv in ((8, 3), (16, 5), (5, 2), (64, 2))]
    for k in (2, 3, 5, 12):
        xs = [rnd.getrandbits(64) for _ in range(k)]
        ys = list(xs)
        rnd.shuffle(ys)
        cases.append((o.K_SHUFFLE, k, xs + ys))
    cases += [(o.K_DUMMY, n, []) for n in (3, 20)]
    for vkeys in (0, 1):
        host.bph_set_seeded_vector_keys(vkeys)
        try:
            for kind, param, values in cases:
                label = bytes(rnd.getrandbits(8) | 1 for _ in range(rnd.randrange(1, 30)))
                seed, cap = rnd.getrandbits(48), 128
                rc_o, proof_o, com_o = o.r1cs_prove(kind, param, label, values, seed, cap, vector_keys=bool(vkeys))
                rc, proof, com = _prove(host, kind, param, label, values, seed, cap)
                assert rc == rc_o == 0 and proof == proof_o and com == com_o, (kind, param, vkeys)
                vals_v = values if kind == o.K_DUMMY else []
                assert _verify(host, kind, param, label, vals_v, com, proof, cap)[0] == 0
                bad = bytearray(proof)
                bad[8 + 11 * 64 + rnd.randrange(96)] ^= 1 << rnd.randrange(7)         # t_x / t_x_blinding / e_blinding
                rc_bad = _verify(host, kind, param, label, vals_v, com, bytes(bad), cap)[0]
                assert rc_bad != 0 and (o.r1cs_verify(kind, param, label, vals_v, com, bytes(bad), cap) != 0)
        finally:
            host.bph_set_seeded_vector_keys(0)


def test_verifier_matches_oracle(host, golden_r1cs):
    for rec in golden_r1cs["range"]:
        com = b"".join(map(H, rec["commitments"]))
        rc, mega = _verify(host, o.K_RANGE, rec["n_bits"], H(rec["label"]), [], com, H(rec["proof"]), 16)
        assert (rc == 0) == rec["ok"] and mega == H(rec["mega_check"])
    for rec in golden_r1cs["shuffle"]:
        com = b"".join(map(H, rec["commitments"]))
        rc, mega = _verify(host, o.K_SHUFFLE, rec["k"], H(rec["label"]), [], com, H(rec["proof"]), 16)
        assert (rc == 0) == rec["ok"] and mega == H(rec["mega_check"])


def test_prover_blinding_randomness_comes_from_the_os_by_default(host):
    """seed = all ones selects OsRng (getrandom(2)-keyed keccak DRBG re-keyed with the transcript state and the witness
    blindings, as prover.rs:435-445): two proofs of the same statement share no blinded element, both verify (GPU and
    oracle).  A 64-bit seed -- the SeededRng of the parity tests -- replays byte for byte."""
    OS = 2**64 - 1
    rc1, p1, c1 = _prove(host, o.K_RANGE, 16, b"RangeProofTest", [40000], OS, 16)
    rc2, p2, c2 = _prove(host, o.K_RANGE, 16, b"RangeProofTest", [40000], OS, 16)
    assert rc1 == 0 and rc2 == 0 and len(p1) == len(p2)
    assert c1 != c2                                     # Pedersen commitments hide the same value differently
    k = int.from_bytes(p1[:4], "little")
    chunks = [8 + 64 * i for i in range(11)] + [8 + 11 * 64 + 32 * i for i in range(3)]
    for off in chunks:                                  # A_I1 A_O1 S1 ... T_6, t_x t_x_blinding e_blinding
        w = 64 if off < 8 + 11 * 64 else 32
        a, b = p1[off:off + w], p2[off:off + w]
        assert a != b or a == bytes(w), off           # (identity phase-2 commitments of a 1-phase proof are equal)
    for proof, com in ((p1, c1), (p2, c2)):
        assert _verify(host, o.K_RANGE, 16, b"RangeProofTest", [], com, proof, 16)[0] == 0
        assert o.r1cs_verify(o.K_RANGE, 16, b"RangeProofTest", [], com, proof, 16) == 0
    assert _prove(host, o.K_RANGE, 16, b"RangeProofTest", [40000], 5, 16) == _prove(host, o.K_RANGE, 16, b"RangeProofTest", [40000], 5, 16)
    assert k == 4


def test_prove_verify_64bit_and_errors(host):
    rc, proof, com = _prove(host, o.K_RANGE, 64, b"RangeProofTest", [2**64 - 5], 77, 64)
    assert rc == 0
    rc_o, proof_o, com_o = o.r1cs_prove(o.K_RANGE, 64, b"RangeProofTest", [2**64 - 5], 77, 64)
    assert (proof, com) == (proof_o, com_o)
    assert _verify(host, o.K_RANGE, 64, b"RangeProofTest", [], com, proof, 64)[0] == 0
    assert o.r1cs_verify(o.K_RANGE, 64, b"RangeProofTest", [], com, proof, 64) == 0
    # InvalidGeneratorsLength both sides (prover.rs:450-452, verifier.rs:421-423)
    assert _prove(host, o.K_RANGE, 64, b"RangeProofTest", [1], 1, 32)[0] == -2
    assert _verify(host, o.K_RANGE, 64, b"RangeProofTest", [], com, proof, 32)[0] == -2
    # identity A_I1 -> VerificationError from the transcript check (transcript.rs:101-113)
    bad = bytearray(proof)
    bad[8:72] = bytes(64)
    assert _verify(host, o.K_RANGE, 64, b"RangeProofTest", [], com, bytes(bad), 64)[0] == -1
    # dummy bench circuit (benches/r1cs.rs:24-33,83): verifier with a different public input rejects
    rc, proof, com = _prove(host, o.K_DUMMY, 8, b"test", [], 3, 8)
    assert rc == 0 and (proof, com) == o.r1cs_prove(o.K_DUMMY, 8, b"test", [], 3, 8)[1:]
    assert _verify(host, o.K_DUMMY, 8, b"test", [], com, proof, 8)[0] == 0
    assert _verify(host, o.K_DUMMY, 8, b"test", [], o.generator(), proof, 8)[0] == -1


@pytest.mark.parametrize("lg", range(1, 11))
def test_dummy_circuit_single_proof_prove_and_verify(host, lg):
    """The reference's r1cs benches (benches/r1cs.rs:24-55,57-108): ONE prover / verifier of the dummy circuit with 2^lg
    multipliers, lg = 1..10 -- the single-proof shapes that hit the round-2 scratch overrun.  Proof bytes and commitment as
    the oracle's; accepted by the GPU verifier and the oracle; a tampered proof is rejected with the oracle's mega_check."""
    n = 1 << lg
    rc, proof, com = _prove(host, o.K_DUMMY, n, b"test", [], 60 + lg, n)
    assert rc == 0
    rc_o, proof_o, com_o = o.r1cs_prove(o.K_DUMMY, n, b"test", [], 60 + lg, n)
    assert rc_o == 0 and (proof, com) == (proof_o, com_o)
    rc, mega = _verify(host, o.K_DUMMY, n, b"test", [], com, proof, n)
    assert rc == 0 and mega == bytes(64)
    assert o.r1cs_verify(o.K_DUMMY, n, b"test", [], com, proof, n) == 0
    bad = bytearray(proof)
    bad[-40] ^= 2                                             # ipp a
    s = o.VerifySession(o.K_DUMMY, n, b"test", [], com, bytes(bad), n)
    rc, mega = _verify(host, o.K_DUMMY, n, b"test", [], com, bytes(bad), n)
    assert rc == -1 and s.rc != 0 and mega == s.mega_check()
    s.close()


def test_prove_batch_two_provers_of_4096_multipliers(host):
    """Two lock-step provers whose blinding vectors are >= 4096 scalars each: Rng::scalars spreads its reductions over the
    thread pool from INSIDE prove_batch's per-prover loop (a nested parallel_for, which deadlocked the pool before round 3).
    Proofs as the oracle's."""
    nb, nvals, n_bits = 2, 64, 64
    n = nvals * n_bits
    label = b"RangeProofTest"
    vals = [((0x9E3779B97F4A7C15 * (i + 3 + 7 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
    arr = (C.c_uint64 * len(vals))(*vals)
    proofs = (C.c_uint8 * (nb * 4096))()
    plen = C.c_size_t(0)
    com = (C.c_uint8 * (nb * nvals * 64))()
    rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(nvals), C.c_size_t(n_bits), o._buf(label), C.c_size_t(len(label)),
                                    arr, C.c_uint64(31), C.c_size_t(n), proofs, C.byref(plen), com)
    assert rc == 0
    L = plen.value
    param = n_bits | (nvals << 16)
    rc_o, proof_o, com_o = o.r1cs_prove(o.K_RANGE_MULTI, param, label, vals[nvals:], 32, n)
    assert rc_o == 0 and bytes(proofs)[L:2 * L] == proof_o and bytes(com)[nvals * 64:] == com_o
    assert _verify(host, o.K_RANGE_MULTI, param, label, [], bytes(com)[:nvals * 64], bytes(proofs)[:L], n)[0] == 0


def test_prover_vector_key_mode_matches_oracle(host):
    """Rng::vector_keys (OsRng's default): the blinding vectors s_L, s_R are drawn on the device from one 32-byte key per prover
    and phase (BlindVec v1) instead of scalar by scalar on the host.  With the replayable test RNG in that mode the GPU prover's
    proofs equal the oracle's vector-key mode byte for byte: 1-phase (range, example, dummy) and 2-phase (shuffle) circuits."""
    host.bph_set_seeded_vector_keys(1)
    try:
        for kind, param, label, vals, cap in ((o.K_RANGE, 8, b"RangeProofTest", [77], 16), (o.K_RANGE, 64, b"RangeProofTest", [2**63 + 5], 64),
                                              (o.K_SHUFFLE, 4, b"ShuffleProofTest", [5, 9, 2, 7, 2, 5, 7, 9], 16),
                                              (o.K_SHUFFLE, 5, b"ShuffleProofTest", [1, 2, 3, 4, 5, 5, 4, 3, 2, 1], 16),
                                              (o.K_EXAMPLE, 0, b"R1CSExampleGadget", [3, 4, 6, 1, 40, 9], 16), (o.K_DUMMY, 16, b"test", [], 16)):
            rc, proof, com = _prove(host, kind, param, label, vals, 21, cap)
            rc_o, proof_o, com_o = o.r1cs_prove(kind, param, label, vals, 21, cap, vector_keys=True)
            assert rc == 0 and rc_o == 0 and (proof, com) == (proof_o, com_o), (kind, param)
            assert proof != o.r1cs_prove(kind, param, label, vals, 21, cap)[1]
            assert _verify(host, kind, param, label, vals[-1:] if kind == o.K_EXAMPLE else [], com, proof, cap)[0] == 0
    finally:
        host.bph_set_seeded_vector_keys(0)


@pytest.mark.parametrize("prebuild", [0, 1])
@pytest.mark.parametrize("vkeys", [0, 1])
def test_prove_stream_over_worker_threads(host, prebuild, vkeys):
    """bph_range_prove_stream: batches of lock-step provers proved by two worker threads, each on a Device (context) of its own,
    sharing the generator tables and the cached circuit: every proof of every batch equals the oracle's for its seed."""
    nbatch, threads, nb, nvals, n_bits = 3, 2, 3, 4, 8
    n, cap, label = nvals * n_bits, 32, b"RangeProofTest"
    vals = [(17 * (i + 1) + 101 * p) % (1 << n_bits) for p in range(nb) for i in range(nvals)]
    arr = (C.c_uint64 * len(vals))(*vals)
    proofs = (C.c_uint8 * (nbatch * nb * 8192))()
    plen = C.c_size_t(0)
    com = (C.c_uint8 * (nbatch * nb * nvals * 64))()
    ms = (C.c_double * 12)()
    host.bph_set_seeded_vector_keys(vkeys)
    try:
        rc = host.bph_range_prove_stream(C.c_size_t(nbatch), C.c_size_t(threads), C.c_int(prebuild), C.c_int(0), C.c_size_t(nb), C.c_size_t(nvals),
                                         C.c_size_t(n_bits), o._buf(label), C.c_size_t(len(label)), arr, C.c_uint64(700), C.c_size_t(cap),
                                         proofs, C.byref(plen), com, ms)
    finally:
        host.bph_set_seeded_vector_keys(0)
    assert rc == 0 and ms[0] > 0
    L = plen.value
    param = n_bits | (nvals << 16)
    for bi in range(nbatch):
        for p in range(nb):
            rc_o, proof_o, com_o = o.r1cs_prove(o.K_RANGE_MULTI, param, label, vals[p * nvals:(p + 1) * nvals], 700 + bi * nb + p, cap,
                                                vector_keys=bool(vkeys))
            i = bi * nb + p
            assert rc_o == 0 and bytes(proofs)[i * L:(i + 1) * L] == proof_o and bytes(com)[i * nvals * 64:(i + 1) * nvals * 64] == com_o, (bi, p)
    assert n == 32


def test_reference_tests_restated_in_cpp():
    exe = os.path.join(ROOT, "tests", "host", "host_tests")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all passed" in r.stdout


@pytest.mark.parametrize("nb,nvals,n_bits,sample", [(3, 4, 8, (0, 1, 2)), (70, 1, 64, (0, 33, 64, 69))])
def test_prove_batch_lockstep_matches_oracle(host, nb, nvals, n_bits, sample):
    """Prover::prove_batch (config 3 shape: several range-proved values per constraint system, provers in
    lock-step on the GPU) reproduces the oracle's proofs prover by prover, and they verify.  70 provers of 64 multipliers: the
    A_I / A_O / S commitments take the MSM-per-lane walk (k_fixed_msm_m: a full wave and one with six live lanes per class)."""
    n = nvals * n_bits
    cap = n
    vals = [(17 * (i + 1) + 101 * p) % (1 << n_bits) for p in range(nb) for i in range(nvals)]
    arr = (C.c_uint64 * len(vals))(*vals)
    proofs = (C.c_uint8 * (nb * 8192))()
    plen = C.c_size_t(0)
    com = (C.c_uint8 * (nb * nvals * 64))()
    rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(nvals), C.c_size_t(n_bits), o._buf(b"RangeProofTest"),
                                    C.c_size_t(14), arr, C.c_uint64(500), C.c_size_t(cap), proofs, C.byref(plen), com)
    assert rc == 0
    L = plen.value
    param = n_bits | (nvals << 16)
    for p in sample:
        rc_o, proof_o, com_o = o.r1cs_prove(o.K_RANGE_MULTI, param, b"RangeProofTest", vals[p * nvals:(p + 1) * nvals], 500 + p, cap)
        assert rc_o == 0
        assert bytes(proofs)[p * L:(p + 1) * L] == proof_o
        assert bytes(com)[p * nvals * 64:(p + 1) * nvals * 64] == com_o
        assert o.r1cs_verify(o.K_RANGE_MULTI, param, b"RangeProofTest", [], com_o, proof_o, cap) == 0
        assert _verify(host, o.K_RANGE_MULTI, param, b"RangeProofTest", [], com_o, proof_o, cap)[0] == 0


# ------------------------------------------------------------------ BASELINE full sizes (configs[2], configs[3])
def test_config2_full_size_256_provers_16x64bit(host):
    """BASELINE.json configs[2]: 256 provers in lock-step, each range-proving 16 values of 64 bits in one constraint
    system (n = 1024 multipliers, q = 2064 constraints, m = 16).  Checked at full size through properties that do not
    need the oracle for every proof: a sample of proofs verifies (GPU verifier), a tampered proof and a proof checked
    against another prover's commitments are rejected; the first and the last prover's proofs are byte-identical to
    the CPU oracle's (1 s each)."""
    nb, nvals, n_bits = 256, 16, 64
    n = nvals * n_bits
    label = b"RangeProofTest"
    vals = [((0x9E3779B97F4A7C15 * (i + 1 + 31 * p)) & ((1 << 64) - 1)) for p in range(nb) for i in range(nvals)]
    arr = (C.c_uint64 * len(vals))(*vals)
    proofs = (C.c_uint8 * (nb * 4096))()
    plen = C.c_size_t(0)
    com = (C.c_uint8 * (nb * nvals * 64))()
    rc = host.bph_range_prove_batch(C.c_size_t(nb), C.c_size_t(nvals), C.c_size_t(n_bits), o._buf(label), C.c_size_t(len(label)),
                                    arr, C.c_uint64(900), C.c_size_t(n), proofs, C.byref(plen), com)
    assert rc == 0
    L = plen.value
    assert L == 8 + 11 * 64 + 3 * 32 + 2 * 10 * 64 + 2 * 32      # 10 IPP rounds for n = 1024
    pb, cb = bytes(proofs), bytes(com)
    param = n_bits | (nvals << 16)
    proof_of = lambda p: pb[p * L:(p + 1) * L]                      # noqa: E731
    com_of = lambda p: cb[p * nvals * 64:(p + 1) * nvals * 64]      # noqa: E731
    assert len({proof_of(p) for p in range(nb)}) == nb              # 256 distinct proofs
    for p in (0, 1, 77, 128, 254, 255):
        assert _verify(host, o.K_RANGE_MULTI, param, label, [], com_of(p), proof_of(p), n)[0] == 0, p
    bad = bytearray(proof_of(77))
    bad[L // 2] ^= 1
    assert _verify(host, o.K_RANGE_MULTI, param, label, [], com_of(77), bytes(bad), n)[0] != 0
    assert _verify(host, o.K_RANGE_MULTI, param, label, [], com_of(78), proof_of(77), n)[0] != 0
    for p in (0, nb - 1):
        rc_o, proof_o, com_o = o.r1cs_prove(o.K_RANGE_MULTI, param, label, vals[p * nvals:(p + 1) * nvals], 900 + p, n)
        assert rc_o == 0 and proof_of(p) == proof_o and com_of(p) == com_o, p


def test_config3_full_size_shuffle_2e14(host):
    """BASELINE.json configs[3]: the k-shuffle gadget at k = 2^14 (n = 32 766 multipliers, q = 65 533 constraints,
    m = 32 768 commitments, a 98 347-term mega_check): prove and verify on the GPU through the host mirror (two-phase
    circuit: the randomized constraints are built on the host from the transcript challenge); a proof of a
    non-permutation must not verify.  (The oracle proves this size in minutes: byte parity is pinned at k = 2^8 in
    tools/bench_shuffle.py and at the golden sizes in test_prover_bytes_identical_to_oracle.)"""
    k = 1 << 14
    cap = 1 << 15
    xs = [((0x9E3779B97F4A7C15 * (i + 1)) & ((1 << 64) - 1)) for i in range(k)]
    ys = xs[1:] + xs[:1]                                    # a rotation: a valid shuffle
    ms = (C.c_double * 6)()

    def run(values):
        arr = (C.c_uint64 * (2 * k))(*values)
        proof = (C.c_uint8 * 8192)()
        plen = C.c_size_t(0)
        com = (C.c_uint8 * (2 * k * 64))()
        rc = host.bph_shuffle_prove_verify(C.c_size_t(k), arr, C.c_uint64(4242), C.c_size_t(cap), proof, C.byref(plen), com, ms)
        return rc, plen.value

    rc, L = run(xs + ys)
    assert rc == 0                                          # proved and accepted
    assert L == 8 + 11 * 64 + 3 * 32 + 2 * 15 * 64 + 2 * 32      # flat proof: 11 points, 3 scalars, 15 IPP rounds, a, b
    ys_bad = list(ys)
    ys_bad[5] ^= 1                                          # no longer a permutation of xs
    rc_bad, _ = run(xs + ys_bad)
    assert rc_bad != 0


@pytest.mark.parametrize("lgk", [1, 3, 6, 14])
def test_shuffle_verified_against_a_parametric_circuit(host, lgk):
    """Verifier::verify(proof, gens, ParametricCircuit) -- the k-shuffle's randomized constraints (tests/r1cs.rs:23-62) handed to the
    device ONCE as coefficients affine in the gadget challenge, the verifier no longer executing the gadget per proof
    (verifier.rs:366-385) -- gives the verdicts of the ordinary Verifier::verify: valid proofs accepted (k = 2 ... 2^14, the last
    one BASELINE.json configs[3]'s full size), a proof of a non-permutation, a tampered proof byte and a swapped commitment rejected
    with the same error code."""
    k = 1 << lgk
    cap = max(2, 2 * k)
    xs = [((0x9E3779B97F4A7C15 * (i + 1)) & ((1 << 64) - 1)) for i in range(k)]
    ys = xs[1:] + xs[:1]
    ms, ms3 = (C.c_double * 6)(), (C.c_double * 3)()

    def prove(values):
        arr = (C.c_uint64 * (2 * k))(*values)
        proof, plen, com = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * (2 * k * 64))()
        rc = host.bph_shuffle_prove_verify(C.c_size_t(k), arr, C.c_uint64(777 + lgk), C.c_size_t(cap), proof, C.byref(plen), com, ms)
        return rc, bytes(proof)[:plen.value], bytes(com)

    def verify_param(proof, com, reps=2):
        return host.bph_shuffle_verify_param(C.c_size_t(k), (C.c_uint8 * len(com)).from_buffer_copy(com),
                                             (C.c_uint8 * len(proof)).from_buffer_copy(proof), C.c_size_t(len(proof)), C.c_size_t(cap),
                                             C.c_size_t(reps), ms3)

    rc, proof, com = prove(xs + ys)
    assert rc == 0
    assert verify_param(proof, com) == 0
    # the PROVER bound to the same circuit (Prover::use_circuit: gadget run for its witness only, no rows built or uploaded;
    # bpgpu_r1cs_prover_session_polys_param): byte for byte the ordinary prover's proof under the same seeded randomness
    arr = (C.c_uint64 * (2 * k))(*(xs + ys))
    proof2, plen2, com2 = (C.c_uint8 * 8192)(), C.c_size_t(0), (C.c_uint8 * (2 * k * 64))()
    assert host.bph_shuffle_prove_param(C.c_size_t(k), arr, C.c_uint64(777 + lgk), C.c_size_t(cap), proof2, C.byref(plen2), com2, ms3) == 0
    assert bytes(proof2)[:plen2.value] == proof and bytes(com2) == com
    if k > 1:
        ys_bad = list(ys)
        ys_bad[0] ^= 1
        rc_bad, proof_bad, com_bad = prove(xs + ys_bad)          # (the prover runs; the mirror's own verify rejects: rc != 0)
        assert rc_bad != 0 and verify_param(proof_bad, com_bad) == rc_bad
        t = bytearray(proof)
        t[8 + 11 * 64 + 3] ^= 4                                     # t_x
        assert verify_param(bytes(t), com) == rc_bad                # VerificationError either way
        swapped = com[64:128] + com[:64] + com[128:]
        assert verify_param(proof, swapped) == rc_bad
