"""Helpers shared by the parity tests, smoke() and bench.py: assemble the byte arrays of
include/bpgpu.h from oracle-generated proofs.  Test infrastructure."""
import oracle_lib as o


def parse_flat_proof(proof):
    k = int.from_bytes(proof[:4], "little")
    off = 8
    pts11 = proof[off:off + 11 * 64]
    off += 11 * 64
    sc3 = proof[off:off + 96]
    off += 96
    L = proof[off:off + 64 * k]
    off += 64 * k
    R = proof[off:off + 64 * k]
    off += 64 * k
    ab = proof[off:off + 64]
    return k, pts11, sc3, L, R, ab


def verify_inputs(proof, commitments):
    """-> (k, points[(11+m+2k)*64], scalars[5*32]) in the layout of bpgpu_r1cs_verify_batch."""
    k, pts11, sc3, L, R, ab = parse_flat_proof(proof)
    points = pts11[:6 * 64] + commitments + pts11[6 * 64:] + L + R
    return k, points, sc3 + ab


def csr_of(session):
    rp, kind, idx, coeff = session.csr()
    return rp, kind, idx, coeff


def make_range_batch(n_bits, nb, seed0=1000, label=b"RangeProofTest", tamper=()):
    """nb proofs of the n_bits range gadget (tests/r1cs.rs:620-652) from the CPU oracle.
    Returns dict with per-proof byte arrays and one VerifySession per proof (for expectations)."""
    cap = 1 << max(0, (n_bits - 1).bit_length())
    recs = []
    for i in range(nb):
        v = (0x9E3779B97F4A7C15 * (i + 1) + seed0) % (1 << n_bits)
        rc, proof, com = o.r1cs_prove(o.K_RANGE, n_bits, label, [v], seed0 + i, cap)
        assert rc == 0
        if i in tamper:
            bad = bytearray(proof)
            bad[8 + 11 * 64 + (i % 3) * 32] ^= 1 + (i % 7)   # flip a bit of t_x / t_x_blinding / e_blinding
            proof = bytes(bad)
        recs.append((proof, com))
    return recs, cap
