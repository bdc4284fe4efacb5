"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ORACLE_DIR, "liboracle.so")

K_RANGE, K_SHUFFLE, K_EXAMPLE, K_DUMMY, K_RANGE_MULTI = 0, 1, 2, 3, 4
N = 0x0800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F
P = 2**251 + 17 * 2**192 + 1


def _load():
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "liboracle.so"])
    lib = C.CDLL(_SO)
    lib.bpo_verify_open.restype = C.c_void_p
    lib.bpo_proof_flat_size.restype = C.c_size_t
    return lib


lib = _load()
u8p = C.POINTER(C.c_uint8)


def _buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(b) if len(b) else (C.c_uint8 * 1)()


def _out(n):
    return (C.c_uint8 * max(n, 1))()


def s2b(x):
    return (x % N).to_bytes(32, "little")


def b2s(b):
    return int.from_bytes(bytes(b), "little")


def scalars(xs):
    return b"".join(s2b(x) for x in xs)


def unscalars(b):
    b = bytes(b)
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def sc_binop(op, a, b):
    n = len(a) // 32
    o = _out(32 * n)
    assert lib.bpo_sc_binop(op, _buf(a), _buf(b), C.c_size_t(n), o) == 0
    return bytes(o)[:32 * n]


def sc_inv(a):
    n = len(a) // 32
    o = _out(32 * n)
    assert lib.bpo_sc_inv(_buf(a), C.c_size_t(n), o) == 0
    return bytes(o)[:32 * n]


def batch_inverse(a):
    n = len(a) // 32
    o = _buf(a)
    assert lib.bpo_sc_batch_inverse(o, C.c_size_t(n)) == 0
    return bytes(o)[:32 * n]


def inner_product(a, b):
    o = _out(32)
    assert lib.bpo_inner_product(_buf(a), _buf(b), C.c_size_t(len(a) // 32), o) == 0
    return bytes(o)


def hash_to_scalar(low):
    o = _out(32)
    lib.bpo_hash_to_scalar(_buf(low), o)
    return bytes(o)


def keccak256(data):
    o = _out(32)
    lib.bpo_keccak256(_buf(data), C.c_size_t(len(data)), o)
    return bytes(o)


def exp_iter(x, n):
    o = _out(32 * n)
    assert lib.bpo_exp_iter(_buf(x), C.c_size_t(n), o) == 0
    return bytes(o)[:32 * n]


def sum_of_powers(x, n, slow=False):
    o = _out(32)
    assert lib.bpo_sum_of_powers(_buf(x), C.c_size_t(n), int(slow), o) == 0
    return bytes(o)


def random_scalars(seed, n):
    o = _out(32 * n)
    lib.bpo_random_scalars(C.c_uint64(seed), C.c_size_t(n), o)
    return bytes(o)[:32 * n]


def point_add(a, b):
    o = _out(64)
    assert lib.bpo_point_add(_buf(a), _buf(b), o) == 0
    return bytes(o)


def point_mul(s, p):
    o = _out(64)
    assert lib.bpo_point_mul(_buf(s), _buf(p), o) == 0
    return bytes(o)


def msm(sc, pts, algo=0):
    n = len(sc) // 32
    assert len(pts) == 64 * n
    o = _out(64)
    assert lib.bpo_msm(_buf(sc), _buf(pts), C.c_size_t(n), algo, o) == 0
    return bytes(o)


def msm_batch(sc, pts, nb, n):
    o = _out(64 * nb)
    assert lib.bpo_msm_batch(_buf(sc), _buf(pts), C.c_size_t(nb), C.c_size_t(n), o) == 0
    return bytes(o)[:64 * nb]


def gens(which, n, party=0, dlogs=False):
    o = _out(64 * n)
    d = _out(32 * n)
    lib.bpo_gens(ord(which), C.c_uint32(party), C.c_size_t(n), o, d)
    return (bytes(o)[:64 * n], bytes(d)[:32 * n]) if dlogs else bytes(o)[:64 * n]


def generator():
    o = _out(64)
    lib.bpo_generator(o)
    return bytes(o)


def fold_witness(n, u, u_inv, a, b, G, H):
    ao, bo, Go, Ho = _out(32 * n), _out(32 * n), _out(64 * n), _out(64 * n)
    assert lib.bpo_fold_witness(C.c_size_t(n), _buf(u), _buf(u_inv), _buf(a), _buf(b), _buf(G), _buf(H),
                                ao, bo, Go, Ho) == 0
    return bytes(ao)[:32 * n], bytes(bo)[:32 * n], bytes(Go)[:64 * n], bytes(Ho)[:64 * n]


def verification_scalars(ch, n):
    k = len(ch) // 32
    a, b, s = _out(32 * k), _out(32 * k), _out(32 * n)
    assert lib.bpo_verification_scalars(_buf(ch), C.c_size_t(k), C.c_size_t(n), a, b, s) == 0
    return bytes(a)[:32 * k], bytes(b)[:32 * k], bytes(s)[:32 * n]


def ipp_create(label, n, Q, Gf, Hf, G, H, a, b):
    k = n.bit_length() - 1
    L, R, ao, bo, ch = _out(64 * k), _out(64 * k), _out(32), _out(32), _out(32 * k)
    assert lib.bpo_ipp_create(_buf(label), C.c_size_t(len(label)), C.c_size_t(n), _buf(Q), _buf(Gf), _buf(Hf),
                              _buf(G), _buf(H), _buf(a), _buf(b), L, R, ao, bo, ch) == 0
    return bytes(L)[:64 * k], bytes(R)[:64 * k], bytes(ao), bytes(bo), bytes(ch)[:32 * k]


def ipp_verify(label, n, Gf, Hf, Pp, Q, G, H, L, R, a, b):
    k = len(L) // 64
    return lib.bpo_ipp_verify(_buf(label), C.c_size_t(len(label)), C.c_size_t(n), _buf(Gf), _buf(Hf), _buf(Pp),
                              _buf(Q), _buf(G), _buf(H), _buf(L), _buf(R), C.c_size_t(k), _buf(a), _buf(b))


def blind_vector(key, v, count):
    """BlindVec v1 (oracle/bpo.h): the prover's device-drawn blinding vector s_L (v = 0) / s_R (v = 1), count x 32 B"""
    out = _out(32 * count)
    lib.bpo_blind_vector(_buf(key), C.c_int(v), C.c_size_t(count), out)
    return bytes(out)[:32 * count]


def r1cs_prove(kind, param, label, values, seed, gens_capacity, vector_keys=False):
    """-> (rc, proof_bytes, commitments_bytes).  vector_keys: the blinding vectors as BlindVec v1 keys drawn from the stream"""
    lib.bpo_set_vector_keys(1 if vector_keys else 0)
    try:
        return _r1cs_prove(kind, param, label, values, seed, gens_capacity)
    finally:
        lib.bpo_set_vector_keys(0)


def _r1cs_prove(kind, param, label, values, seed, gens_capacity):
    vals = (C.c_uint64 * max(len(values), 1))(*values)
    proof = _out(lib.bpo_proof_flat_size(C.c_size_t(32)))
    plen, m = C.c_size_t(0), C.c_size_t(0)
    mmax = {K_RANGE: 1, K_SHUFFLE: 2 * param, K_EXAMPLE: 5, K_DUMMY: 1, K_RANGE_MULTI: param >> 16}[kind]
    com = _out(64 * mmax)
    rc = lib.bpo_r1cs_prove(kind, C.c_size_t(param), _buf(label), C.c_size_t(len(label)), vals,
                            C.c_size_t(len(values)), C.c_uint64(seed), C.c_size_t(gens_capacity), proof,
                            C.byref(plen), com, C.byref(m))
    return rc, bytes(proof)[:plen.value], bytes(com)[:64 * m.value]


class VerifySession:
    def __init__(self, kind, param, label, values, commitments, proof, gens_capacity):
        vals = (C.c_uint64 * max(len(values), 1))(*values)
        m = len(commitments) // 64
        self.h = C.c_void_p(lib.bpo_verify_open(kind, C.c_size_t(param), _buf(label), C.c_size_t(len(label)), vals,
                                                C.c_size_t(len(values)), _buf(commitments), C.c_size_t(m),
                                                _buf(proof), C.c_size_t(len(proof)), C.c_size_t(gens_capacity)))
        self.rc = lib.bpo_verify_rc(self.h)
        d = (C.c_uint64 * 8)()
        lib.bpo_verify_dims(self.h, d)
        (self.n1, self.n2, self.padded_n, self.k, self.m, self.nterms, self.q, self.nnz) = [int(x) for x in d]

    def challenges(self):
        o = _out(32 * (6 + self.k))
        lib.bpo_verify_challenges(self.h, o)
        return bytes(o)[:32 * (6 + self.k)]

    def msm_terms(self):
        s, p = _out(32 * self.nterms), _out(64 * self.nterms)
        lib.bpo_verify_msm(self.h, s, p)
        return bytes(s)[:32 * self.nterms], bytes(p)[:64 * self.nterms]

    def mega_check(self):
        o = _out(64)
        lib.bpo_verify_mega(self.h, o)
        return bytes(o)

    def csr(self):
        rp = (C.c_uint32 * (self.q + 1))()
        kind = (C.c_uint32 * max(self.nnz, 1))()
        idx = (C.c_uint32 * max(self.nnz, 1))()
        coeff = _out(32 * self.nnz)
        lib.bpo_verify_csr(self.h, rp, kind, idx, coeff)
        return list(rp), list(kind)[:self.nnz], list(idx)[:self.nnz], bytes(coeff)[:32 * self.nnz]

    def flatten(self, z):
        n, m = self.n1 + self.n2, self.m
        wL, wR, wO, wV, wc = _out(32 * n), _out(32 * n), _out(32 * n), _out(32 * m), _out(32)
        lib.bpo_verify_flatten(self.h, _buf(z), wL, wR, wO, wV, wc)
        return bytes(wL)[:32 * n], bytes(wR)[:32 * n], bytes(wO)[:32 * n], bytes(wV)[:32 * m], bytes(wc)

    def close(self):
        if self.h:
            lib.bpo_verify_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def r1cs_verify(kind, param, label, values, commitments, proof, gens_capacity):
    s = VerifySession(kind, param, label, values, commitments, proof, gens_capacity)
    rc = s.rc
    s.close()
    return rc


def r1cs_verify_many(kind, param, label, values, commitments, m, proofs, proof_len, nb, gens_capacity):
    vals = (C.c_uint64 * max(len(values), 1))(*values)
    ok = (C.c_int32 * max(nb, 1))()
    rc = lib.bpo_r1cs_verify_many(kind, C.c_size_t(param), _buf(label), C.c_size_t(len(label)), vals,
                                  C.c_size_t(len(values)), _buf(commitments), C.c_size_t(m), _buf(proofs),
                                  C.c_size_t(proof_len), C.c_size_t(nb), C.c_size_t(gens_capacity), ok)
    assert rc == 0, rc
    return list(ok)[:nb]
