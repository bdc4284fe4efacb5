"""Pure-Python big-integer model of the Bulletproofs hot path over the Stark curve.

TEST INFRASTRUCTURE ONLY.  This file is the *second* independent restatement
(the first is the C oracle in this directory).  It exists to pin the C oracle:
`oracle/gen_golden.py` runs it in the build container and commits hex fixtures
under `tests/golden/`.  Nothing in the product path imports it.

Parity status: the arithmetic the reference delegates to `mpc-stark 0.2`
(crates.io, Cargo.toml:21, no Cargo.lock) and to the un-pinned
`renegade-fi/merlin` git fork (Cargo.toml:36) is NOT in /root/reference, so
  * field / curve / MSM / IPP / R1CS algebra: restated from the published curve
    parameters and the reference's call sites; pinned by the
    reference's own KATs (inner_product == 40, powers of 2, sums of powers of
    10, EXAMPLE_GADGET_WEIGHTS) and by algebraic identities;
  * transcript bytes: **parity unpinned** -- `HashChainTranscript` internals are
    not specified by any file in the reference.  The transcript below is *a*
    deterministic keccak hash chain with the same labels and call order as the
    reference; challenges are therefore stored explicitly in every fixture.

Each function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

# --------------------------------------------------------------------------- constants
# Stark curve (Cargo.toml:14 "Bulletproofs over the Stark curve").
P = 2**251 + 17 * 2**192 + 1
N = 0x0800000000000010FFFFFFFFFFFFFFFFB781126DCAE7B2321E66A241ADC64D2F
CURVE_A = 1
CURVE_B = 0x06F21413EFBE40DE150E596D72F7A8C5609AD26C15C915C1F4CDFCB99CEE9E89
GX = 0x01EF15C18599971B7BECED415A40F0C7DEACFD9B0D1819E03D723D8BC943CFCA
GY = 0x005668060AA49730B7BE4801DF46EC62DE53ECD11ABE43A32873000C36E8DC1F
G = (GX, GY)
INF = None  # identity

# --------------------------------------------------------------------------- keccak256
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61],
        [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, r):
    r %= 64
    return ((x << r) | (x >> (64 - r))) & _M64 if r else x


def _keccak_f(A):
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)]
             for x in range(5)]
        A[0][0] ^= rc
    return A


def keccak256(data: bytes) -> bytes:
    """Original Keccak-256 (pad 0x01, not SHA3's 0x06); merlin fork `keccak256`
    as used at src/util.rs:255, src/generators.rs:86,96,114."""
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        A = _keccak_f(A)
    out = b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


def pad_label(label: bytes) -> bytes:
    """merlin fork `pad_label` (src/transcript.rs:65, src/generators.rs:84): source
    absent -> defined here as zero-padding on the right to a multiple of 32 bytes
    (at least 32).  Parity unpinned."""
    k = max(32, (len(label) + 31) // 32 * 32)
    return label + b"\0" * (k - len(label))


# --------------------------------------------------------------------------- scalars
def hash_to_scalar(low: bytes) -> int:
    """src/util.rs:252-267: high = keccak256(low); int_LE(low || high) mod n."""
    high = keccak256(low)
    return int.from_bytes(low + high, "little") % N


def s2b(s: int) -> bytes:
    """32-byte little-endian canonical scalar (boundary encoding)."""
    return (s % N).to_bytes(32, "little")


def b2s(b: bytes) -> int:
    return int.from_bytes(b, "little")


def inv(s: int) -> int:
    return pow(s, -1, N)


def inner_product(a, b) -> int:
    """src/inner_product_proof.rs:463-472."""
    if len(a) != len(b):
        raise ValueError("inner_product(a,b): lengths of vectors do not match")
    out = 0
    for x, y in zip(a, b):
        out = (out + x * y) % N
    return out


def exp_iter(x: int, n: int):
    """src/util.rs:73-76 (first n powers)."""
    out, cur = [], 1
    for _ in range(n):
        out.append(cur)
        cur = cur * x % N
    return out


def sum_of_powers_slow(x: int, n: int) -> int:
    """src/util.rs:237-239."""
    return sum(exp_iter(x, n)) % N


def sum_of_powers(x: int, n: int) -> int:
    """src/util.rs:218-234."""
    if n & (n - 1):
        return sum_of_powers_slow(x, n)
    if n in (0, 1):
        return n
    m, result, factor = n, (1 + x) % N, x
    while m > 2:
        factor = factor * factor % N
        result = (result + factor * result) % N
        m //= 2
    return result


def batch_inverse(v):
    """Scalar::batch_inverse (mpc-stark; call site inner_product_proof.rs:283):
    Montgomery's trick; inputs must be non-zero."""
    pref, acc = [], 1
    for x in v:
        pref.append(acc)
        acc = acc * x % N
    accinv = inv(acc)
    out = [0] * len(v)
    for i in range(len(v) - 1, -1, -1):
        out[i] = accinv * pref[i] % N
        accinv = accinv * v[i] % N
    return out


# --------------------------------------------------------------------------- curve
def on_curve(Pt) -> bool:
    if Pt is INF:
        return True
    x, y = Pt
    return (y * y - (x * x * x + CURVE_A * x + CURVE_B)) % P == 0


def pt_neg(Pt):
    return INF if Pt is INF else (Pt[0], (-Pt[1]) % P)


def pt_add(A, B):
    """Complete affine addition on y^2 = x^3 + x + b."""
    if A is INF:
        return B
    if B is INF:
        return A
    x1, y1 = A
    x2, y2 = B
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1 + CURVE_A) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return (x3, (lam * (x1 - x3) - y1) % P)


def pt_mul(k: int, Pt):
    k %= N
    acc = INF
    while k:
        if k & 1:
            acc = pt_add(acc, Pt)
        Pt = pt_add(Pt, Pt)
        k >>= 1
    return acc


def msm(scalars, points):
    """StarkPoint::msm / msm_iter (mpc-stark; call sites SURVEY K1): sum s_i * P_i."""
    if len(scalars) != len(points):
        raise ValueError("msm: length mismatch")
    acc = INF
    for s, Pt in zip(scalars, points):
        acc = pt_add(acc, pt_mul(s, Pt))
    return acc


def p2b(Pt) -> bytes:
    """src/util.rs:274-289: affine x||y, 32-byte LE each; identity = 64 zero bytes."""
    if Pt is INF:
        return b"\0" * 64
    return Pt[0].to_bytes(32, "little") + Pt[1].to_bytes(32, "little")


def b2p(b: bytes):
    if b == b"\0" * 64:
        return INF
    return (int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little"))


# --------------------------------------------------------------------------- transcript
class Transcript:
    """Stand-in for merlin's `HashChainTranscript` (source absent; parity unpinned).

    state_0            = keccak256(pad_label("bp-hashchain-v1") || pad_label(label))
    append_message     : state = keccak256(state || 0x00 || pad_label(l) || u32le(len) || msg)
                         (v1: a 4-byte length as merlin frames it -- a 64-byte point then makes a 133-byte message, ONE rate
                         block; with v0's 8-byte length it was 137 bytes, two permutations per appended point)
    challenge_bytes(32): state = keccak256(state || 0x01 || pad_label(l)); output = state

    Protocol layer (labels, order, encodings) follows src/transcript.rs:63-121.
    """

    def __init__(self, label: bytes):
        self.state = keccak256(pad_label(b"bp-hashchain-v1") + pad_label(label))

    def append_message(self, label: bytes, msg: bytes):
        self.state = keccak256(self.state + b"\0" + pad_label(label)
                               + len(msg).to_bytes(4, "little") + msg)

    def append_u64(self, label: bytes, x: int):
        self.append_message(label, x.to_bytes(8, "little"))

    def challenge_bytes(self, label: bytes) -> bytes:
        self.state = keccak256(self.state + b"\x01" + pad_label(label))
        return self.state

    # --- TranscriptProtocol, src/transcript.rs:63-121
    def innerproduct_domain_sep(self, n):
        self.append_message(b"dom-sep", pad_label(b"ipp v1"))
        self.append_u64(b"n", n)

    def r1cs_domain_sep(self):
        self.append_message(b"dom-sep", pad_label(b"r1cs v1"))

    def r1cs_1phase_domain_sep(self):
        self.append_message(b"dom-sep", pad_label(b"r1cs-1phase"))

    def r1cs_2phase_domain_sep(self):
        self.append_message(b"dom-sep", pad_label(b"r1cs-2phase"))

    def append_scalar(self, label, s):
        self.append_message(label, s2b(s))  # LE, src/transcript.rs:87-92

    def append_point(self, label, Pt):
        self.append_message(label, p2b(Pt))

    def validate_and_append_point(self, label, Pt):
        if Pt is INF:
            raise VerificationError()
        self.append_message(label, p2b(Pt))

    def challenge_scalar(self, label) -> int:
        return hash_to_scalar(self.challenge_bytes(label))


class VerificationError(Exception):
    pass


class InvalidGeneratorsLength(Exception):
    pass


# --------------------------------------------------------------------------- generators
class PedersenGens:
    """src/generators.rs:61-70: B = B_blinding = curve generator."""

    def __init__(self):
        self.B = G
        self.B_blinding = G

    def commit(self, value, blinding):
        return pt_add(pt_mul(value, self.B), pt_mul(blinding, self.B_blinding))


def generators_chain(label: bytes, count: int, skip: int = 0):
    """src/generators.rs:76-129.  Returns (scalars k_i, points k_i*G)."""
    state = keccak256(pad_label(b"GeneratorsChain" + label))
    for _ in range(skip):
        state = keccak256(state)
    ks, pts = [], []
    for _ in range(count):
        state = keccak256(state)
        k = hash_to_scalar(state)
        ks.append(k)
        pts.append(pt_mul(k, G))
    return ks, pts


class BulletproofGens:
    """src/generators.rs:158-235 (party share 0 only is used by the R1CS path)."""

    def __init__(self, gens_capacity: int, party_capacity: int = 1):
        self.gens_capacity = gens_capacity
        self.party_capacity = party_capacity
        self.G_vec, self.H_vec, self.G_dlog, self.H_dlog = [], [], [], []
        for i in range(party_capacity):
            lab = i.to_bytes(4, "little")
            kg, g = generators_chain(b"G" + lab, gens_capacity)
            kh, h = generators_chain(b"H" + lab, gens_capacity)
            self.G_vec.append(g)
            self.H_vec.append(h)
            self.G_dlog.append(kg)
            self.H_dlog.append(kh)

    def G(self, n, share=0):
        return self.G_vec[share][:n]

    def H(self, n, share=0):
        return self.H_vec[share][:n]


# --------------------------------------------------------------------------- RNG (injectable)
class SplitMix64:
    """Deterministic replacement for the reference's `thread_rng()`
    (src/r1cs/prover.rs:435-445): blinding factors are *inputs* of the hot path."""

    def __init__(self, seed: int):
        self.s = seed & _M64

    def next_u64(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def scalar(self) -> int:
        v = 0
        for i in range(4):
            v |= self.next_u64() << (64 * i)
        return v % N


# --------------------------------------------------------------------------- inner product proof
def fold_witness(u, u_inv, a_L, a_R, b_L, b_R, G_L, G_R, H_L, H_R):
    """src/inner_product_proof.rs:202-248."""
    n = len(a_L)
    a = [(a_L[i] * u + u_inv * a_R[i]) % N for i in range(n)]
    b = [(b_L[i] * u_inv + u * b_R[i]) % N for i in range(n)]
    Gn = [msm([u_inv, u], [G_L[i], G_R[i]]) for i in range(n)]
    Hn = [msm([u, u_inv], [H_L[i], H_R[i]]) for i in range(n)]
    return a, b, Gn, Hn


def ipp_create(transcript: Transcript, Q, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec,
               trace=None):
    """src/inner_product_proof.rs:49-193.  Returns (L_vec, R_vec, a, b).
    `trace`, if a list, receives (u, u_inv) per round."""
    n = len(G_vec)
    assert len(H_vec) == n and len(a_vec) == n and len(b_vec) == n
    assert len(G_factors) == n and len(H_factors) == n
    assert n and (n & (n - 1)) == 0
    transcript.innerproduct_domain_sep(n)
    L_vec, R_vec = [], []
    first = True
    while n != 1:
        n //= 2
        a_L, a_R = a_vec[:n], a_vec[n:]
        b_L, b_R = b_vec[:n], b_vec[n:]
        G_L, G_R = G_vec[:n], G_vec[n:]
        H_L, H_R = H_vec[:n], H_vec[n:]
        c_L = inner_product(a_L, b_R)
        c_R = inner_product(a_R, b_L)
        if first:
            L = msm([a_L[i] * G_factors[n + i] % N for i in range(n)]
                    + [b_R[i] * H_factors[i] % N for i in range(n)] + [c_L],
                    G_R + H_L + [Q])
            R = msm([a_R[i] * G_factors[i] % N for i in range(n)]
                    + [b_L[i] * H_factors[n + i] % N for i in range(n)] + [c_R],
                    G_L + H_R + [Q])
        else:
            L = msm(a_L + b_R + [c_L], G_R + H_L + [Q])
            R = msm(a_R + b_L + [c_R], G_L + H_R + [Q])
        L_vec.append(L)
        R_vec.append(R)
        transcript.append_point(b"L", L)
        transcript.append_point(b"R", R)
        u = transcript.challenge_scalar(b"u")
        u_inv = inv(u)
        if trace is not None:
            trace.append((u, u_inv))
        if first:
            Gs = [pt_mul(G_factors[i], G_vec[i]) for i in range(2 * n)]
            Hs = [pt_mul(H_factors[i], H_vec[i]) for i in range(2 * n)]
            G_L, G_R, H_L, H_R = Gs[:n], Gs[n:], Hs[:n], Hs[n:]
            first = False
        a_vec, b_vec, G_vec, H_vec = fold_witness(u, u_inv, a_L, a_R, b_L, b_R,
                                                  G_L, G_R, H_L, H_R)
    return L_vec, R_vec, a_vec[0], b_vec[0]


def ipp_challenges(L_vec, R_vec, n, transcript: Transcript):
    """Transcript replay part of verification_scalars, inner_product_proof.rs:259-278."""
    lg_n = len(L_vec)
    if lg_n >= 32 or n != (1 << lg_n):
        raise VerificationError()
    transcript.innerproduct_domain_sep(n)
    ch = []
    for L, R in zip(L_vec, R_vec):
        transcript.validate_and_append_point(b"L", L)
        transcript.validate_and_append_point(b"R", R)
        ch.append(transcript.challenge_scalar(b"u"))
    return ch


def verification_scalars_from_challenges(challenges, n):
    """Arithmetic part of verification_scalars, inner_product_proof.rs:280-309."""
    lg_n = len(challenges)
    ch_inv = batch_inverse(challenges)
    allinv = 1
    for c in ch_inv:
        allinv = allinv * c % N
    u_sq = [c * c % N for c in challenges]
    u_inv_sq = [c * c % N for c in ch_inv]
    s = [allinv]
    for i in range(1, n):
        lg_i = i.bit_length() - 1
        k = 1 << lg_i
        s.append(s[i - k] * u_sq[(lg_n - 1) - lg_i] % N)
    return u_sq, u_inv_sq, s


def ipp_verify(L_vec, R_vec, a, b, n, transcript, G_factors, H_factors, Pp, Q, Gv, Hv):
    """src/inner_product_proof.rs:317-372.  Returns expect_P == P."""
    ch = ipp_challenges(L_vec, R_vec, n, transcript)
    u_sq, u_inv_sq, s = verification_scalars_from_challenges(ch, n)
    gs = [a * s[i] % N * G_factors[i] % N for i in range(n)]
    hs = [b * s[n - 1 - i] % N * H_factors[i] % N for i in range(n)]
    expect = msm([a * b % N] + gs + hs + [(-x) % N for x in u_sq] + [(-x) % N for x in u_inv_sq],
                 [Q] + list(Gv) + list(Hv) + list(L_vec) + list(R_vec))
    return expect == Pp


# --------------------------------------------------------------------------- R1CS
# Variables: ('L', i) MultiplierLeft, ('R', i), ('O', i), ('V', i) Committed, ('1',) One
# (src/r1cs/linear_combination.rs:15-28).  LinearCombination = dict var -> coeff.
ONE = ("1",)


def lc(*terms):
    out = {}
    for var, c in terms:
        out[var] = (out.get(var, 0) + c) % N
    return out


def lc_add(a, b):
    out = dict(a)
    for v, c in b.items():
        out[v] = (out.get(v, 0) + c) % N
    return out


def lc_neg(a):
    return {v: (-c) % N for v, c in a.items()}


def lc_sub(a, b):
    return lc_add(a, lc_neg(b))


def lc_var(v):
    return {v: 1}


def lc_const(c):
    return {ONE: c % N}


def extract_weights(l):
    """src/r1cs/linear_combination.rs:140-192: (w_l, w_r, w_o, w_v, c) rows; w_v and c negated."""
    rows = {"L": [], "R": [], "O": [], "V": []}
    c = None
    for var, coeff in sorted(((v, k) for v, k in l.items() if k % N), key=lambda t: t[0][1:] or (1 << 62,)):
        if var[0] in "LRO":
            rows[var[0]].append((var[1], coeff))
        elif var[0] == "V":
            rows["V"].append((var[1], (-coeff) % N))
        elif var == ONE:
            c = (-coeff) % N
    return rows["L"], rows["R"], rows["O"], rows["V"], c


class _CS:
    """Shared circuit-builder state (src/r1cs/prover.rs:27-50, verifier.rs:27-60)."""

    def __init__(self, transcript: Transcript, pc_gens: PedersenGens):
        self.transcript = transcript
        self.pc_gens = pc_gens
        self.constraints = []
        self.deferred = []
        self.pending_multiplier = None
        transcript.r1cs_domain_sep()

    def constrain(self, l):
        self.constraints.append(dict(l))

    def specify_randomized_constraints(self, cb):
        self.deferred.append(cb)

    def challenge_scalar(self, label):
        return self.transcript.challenge_scalar(label)

    def _create_randomized_constraints(self):
        """prover.rs:383-402 / verifier.rs:366-385."""
        self.pending_multiplier = None
        if not self.deferred:
            self.transcript.r1cs_1phase_domain_sep()
        else:
            self.transcript.r1cs_2phase_domain_sep()
            cbs, self.deferred = self.deferred, []
            for cb in cbs:
                cb(self)

    def get_weights(self):
        """prover.rs:76-97."""
        w = {"w_l": [], "w_r": [], "w_o": [], "w_v": [], "c": []}
        for i, l in enumerate(self.constraints):
            a, b, c, d, e = extract_weights(l)
            w["w_l"].append(a)
            w["w_r"].append(b)
            w["w_o"].append(c)
            w["w_v"].append(d)
            if e is not None:
                w["c"].append((i, e))
        return w


class Prover(_CS):
    """src/r1cs/prover.rs."""

    def __init__(self, pc_gens, transcript):
        super().__init__(transcript, pc_gens)
        self.a_L, self.a_R, self.a_O, self.v, self.v_blinding = [], [], [], [], []

    def commit(self, v, v_blinding):
        """prover.rs:319-329."""
        i = len(self.v)
        self.v.append(v % N)
        self.v_blinding.append(v_blinding % N)
        V = self.pc_gens.commit(v, v_blinding)
        self.transcript.append_point(b"V", V)
        return V, ("V", i)

    def commit_public(self, v):
        return self.commit(v, 1)[1]

    def eval(self, l):
        """prover.rs:179-194."""
        tot = 0
        for var, c in l.items():
            val = {"L": self.a_L, "R": self.a_R, "O": self.a_O, "V": self.v}.get(var[0])
            tot += c * (1 if var == ONE else val[var[1]])
        return tot % N

    def multiply(self, left, right):
        """prover.rs:99-125."""
        l, r = self.eval(left), self.eval(right)
        i = len(self.a_L)
        self.a_L.append(l)
        self.a_R.append(r)
        self.a_O.append(l * r % N)
        left = lc_add(left, {("L", i): N - 1})
        right = lc_add(right, {("R", i): N - 1})
        self.constrain(left)
        self.constrain(right)
        return ("L", i), ("R", i), ("O", i)

    def allocate(self, assignment):
        """prover.rs:127-146."""
        if self.pending_multiplier is None:
            i = len(self.a_L)
            self.pending_multiplier = i
            self.a_L.append(assignment % N)
            self.a_R.append(0)
            self.a_O.append(0)
            return ("L", i)
        i, self.pending_multiplier = self.pending_multiplier, None
        self.a_R[i] = assignment % N
        self.a_O[i] = self.a_L[i] * self.a_R[i] % N
        return ("R", i)

    def allocate_multiplier(self, assignment):
        """prover.rs:148-165."""
        l, r = assignment
        i = len(self.a_L)
        self.a_L.append(l % N)
        self.a_R.append(r % N)
        self.a_O.append(l * r % N)
        return ("L", i), ("R", i), ("O", i)

    def flattened_constraints(self, z):
        """prover.rs:342-379."""
        n, m = len(self.a_L), len(self.v)
        wL, wR, wO, wV = [0] * n, [0] * n, [0] * n, [0] * m
        exp_z = z
        for l in self.constraints:
            for var, c in l.items():
                if var[0] == "L":
                    wL[var[1]] = (wL[var[1]] + exp_z * c) % N
                elif var[0] == "R":
                    wR[var[1]] = (wR[var[1]] + exp_z * c) % N
                elif var[0] == "O":
                    wO[var[1]] = (wO[var[1]] + exp_z * c) % N
                elif var[0] == "V":
                    wV[var[1]] = (wV[var[1]] - exp_z * c) % N
            exp_z = exp_z * z % N
        return wL, wR, wO, wV

    def prove(self, bp_gens: BulletproofGens, rng: SplitMix64, trace=None):
        """prover.rs:412-727 with the RNG injected (see SplitMix64)."""
        tr = self.transcript
        tr.append_u64(b"m", len(self.v))
        n1 = len(self.a_L)
        if bp_gens.gens_capacity < n1:
            raise InvalidGeneratorsLength()
        Bb = self.pc_gens.B_blinding
        i_b1, o_b1, s_b1 = rng.scalar(), rng.scalar(), rng.scalar()
        s_L1 = [rng.scalar() for _ in range(n1)]
        s_R1 = [rng.scalar() for _ in range(n1)]
        A_I1 = msm([i_b1] + self.a_L + self.a_R, [Bb] + bp_gens.G(n1) + bp_gens.H(n1))
        A_O1 = msm([o_b1] + self.a_O, [Bb] + bp_gens.G(n1))
        S1 = msm([s_b1] + s_L1 + s_R1, [Bb] + bp_gens.G(n1) + bp_gens.H(n1))
        tr.append_point(b"A_I1", A_I1)
        tr.append_point(b"A_O1", A_O1)
        tr.append_point(b"S1", S1)
        self._create_randomized_constraints()
        n = len(self.a_L)
        n2 = n - n1
        padded_n = 1 if n == 0 else 1 << (n - 1).bit_length()
        pad = padded_n - n
        if bp_gens.gens_capacity < padded_n:
            raise InvalidGeneratorsLength()
        if n2 > 0:
            i_b2, o_b2, s_b2 = rng.scalar(), rng.scalar(), rng.scalar()
        else:
            i_b2 = o_b2 = s_b2 = 0
        s_L2 = [rng.scalar() for _ in range(n2)]
        s_R2 = [rng.scalar() for _ in range(n2)]
        if n2 > 0:
            Gs, Hs = bp_gens.G(n)[n1:], bp_gens.H(n)[n1:]
            A_I2 = msm([i_b2] + self.a_L[n1:] + self.a_R[n1:], [Bb] + Gs + Hs)
            A_O2 = msm([o_b2] + self.a_O[n1:], [Bb] + Gs)
            S2 = msm([s_b2] + s_L2 + s_R2, [Bb] + Gs + Hs)
        else:
            A_I2 = A_O2 = S2 = INF
        tr.append_point(b"A_I2", A_I2)
        tr.append_point(b"A_O2", A_O2)
        tr.append_point(b"S2", S2)
        y = tr.challenge_scalar(b"y")
        z = tr.challenge_scalar(b"z")
        wL, wR, wO, wV = self.flattened_constraints(z)
        y_inv = inv(y)
        exp_y_inv = exp_iter(y_inv, padded_n)
        sL, sR = s_L1 + s_L2, s_R1 + s_R2
        l1, l2, l3 = [0] * n, [0] * n, [0] * n
        r0, r1, r3 = [0] * n, [0] * n, [0] * n
        exp_y = 1
        for i in range(n):
            l1[i] = (self.a_L[i] + exp_y_inv[i] * wR[i]) % N
            l2[i] = self.a_O[i]
            l3[i] = sL[i]
            r0[i] = (wO[i] - exp_y) % N
            r1[i] = (exp_y * self.a_R[i] + wL[i]) % N
            r3[i] = exp_y * sR[i] % N
            exp_y = exp_y * y % N
        # util.rs:152-170 special_inner_product (l0 = 0, r2 = 0)
        ip = inner_product
        t1 = ip(l1, r0)
        t2 = (ip(l1, r1) + ip(l2, r0)) % N
        t3 = (ip(l2, r1) + ip(l3, r0)) % N
        t4 = (ip(l1, r3) + ip(l3, r1)) % N
        t5 = ip(l2, r3)
        t6 = ip(l3, r3)
        tb1, tb3, tb4, tb5, tb6 = (rng.scalar() for _ in range(5))
        pc = self.pc_gens
        T_1, T_3, T_4, T_5, T_6 = (pc.commit(t1, tb1), pc.commit(t3, tb3), pc.commit(t4, tb4),
                                   pc.commit(t5, tb5), pc.commit(t6, tb6))
        for lab, T in ((b"T_1", T_1), (b"T_3", T_3), (b"T_4", T_4), (b"T_5", T_5), (b"T_6", T_6)):
            tr.append_point(lab, T)
        u = tr.challenge_scalar(b"u")
        x = tr.challenge_scalar(b"x")
        tb2 = sum(c * vb for c, vb in zip(wV, self.v_blinding)) % N

        def poly6(c1, c2, c3, c4, c5, c6):  # util.rs:192-194
            return x * (c1 + x * (c2 + x * (c3 + x * (c4 + x * (c5 + x * c6))))) % N

        t_x = poly6(t1, t2, t3, t4, t5, t6)
        t_x_blinding = poly6(tb1, tb2, tb3, tb4, tb5, tb6)
        l_vec = [(x * (l1[i] + x * (l2[i] + x * l3[i]))) % N for i in range(n)] + [0] * pad
        r_vec = [(r0[i] + x * (r1[i] + x * (x * r3[i]))) % N for i in range(n)] + [0] * pad
        for i in range(n, padded_n):
            r_vec[i] = (-exp_y) % N
            exp_y = exp_y * y % N
        i_b = (i_b1 + u * i_b2) % N
        o_b = (o_b1 + u * o_b2) % N
        s_b = (s_b1 + u * s_b2) % N
        e_blinding = x * (i_b + x * (o_b + x * s_b)) % N
        tr.append_scalar(b"t_x", t_x)
        tr.append_scalar(b"t_x_blinding", t_x_blinding)
        tr.append_scalar(b"e_blinding", e_blinding)
        w = tr.challenge_scalar(b"w")
        Q = pt_mul(w, pc.B)
        G_factors = [1] * n1 + [u] * (n2 + pad)
        H_factors = [exp_y_inv[i] * G_factors[i] % N for i in range(padded_n)]
        ipp_trace = [] if trace is not None else None
        L_vec, R_vec, a, b = ipp_create(tr, Q, G_factors, H_factors, bp_gens.G(padded_n),
                                        bp_gens.H(padded_n), l_vec, r_vec, ipp_trace)
        if trace is not None:
            trace.update(dict(y=y, z=z, u=u, x=x, w=w, ipp=ipp_trace, l_vec=l_vec, r_vec=r_vec,
                              wL=wL, wR=wR, wO=wO, wV=wV, t=[t1, t2, t3, t4, t5, t6], Q=Q))
        return dict(A_I1=A_I1, A_O1=A_O1, S1=S1, A_I2=A_I2, A_O2=A_O2, S2=S2, T_1=T_1, T_3=T_3,
                    T_4=T_4, T_5=T_5, T_6=T_6, t_x=t_x, t_x_blinding=t_x_blinding,
                    e_blinding=e_blinding, L_vec=L_vec, R_vec=R_vec, a=a, b=b)


class Verifier(_CS):
    """src/r1cs/verifier.rs."""

    def __init__(self, pc_gens, transcript):
        super().__init__(transcript, pc_gens)
        self.num_vars = 0
        self.V = []

    def commit(self, V):
        """verifier.rs:298-306."""
        i = len(self.V)
        self.V.append(V)
        self.transcript.append_point(b"V", V)
        return ("V", i)

    def multiply(self, left, right):
        """verifier.rs:76-99 (same constraints as the prover, no assignments)."""
        i = self.num_vars
        self.num_vars += 1
        self.constrain(lc_add(left, {("L", i): N - 1}))
        self.constrain(lc_add(right, {("R", i): N - 1}))
        return ("L", i), ("R", i), ("O", i)

    def allocate_multiplier(self, assignment=None):
        """verifier.rs:137-151."""
        i = self.num_vars
        self.num_vars += 1
        return ("L", i), ("R", i), ("O", i)

    def allocate(self, assignment=None):
        """verifier.rs:122-135."""
        if self.pending_multiplier is None:
            i = self.num_vars
            self.num_vars += 1
            self.pending_multiplier = i
            return ("L", i)
        i, self.pending_multiplier = self.pending_multiplier, None
        return ("R", i)

    def commit_public(self, value):
        """verifier.rs:153-160: blinding factor fixed to one."""
        return self.commit(self.pc_gens.commit(value, 1))

    def flattened_constraints(self, z):
        """verifier.rs:323-362."""
        n, m = self.num_vars, len(self.V)
        wL, wR, wO, wV, wc = [0] * n, [0] * n, [0] * n, [0] * m, 0
        exp_z = z
        for l in self.constraints:
            for var, c in l.items():
                if var[0] == "L":
                    wL[var[1]] = (wL[var[1]] + exp_z * c) % N
                elif var[0] == "R":
                    wR[var[1]] = (wR[var[1]] + exp_z * c) % N
                elif var[0] == "O":
                    wO[var[1]] = (wO[var[1]] + exp_z * c) % N
                elif var[0] == "V":
                    wV[var[1]] = (wV[var[1]] - exp_z * c) % N
                elif var == ONE:
                    wc = (wc - exp_z * c) % N
            exp_z = exp_z * z % N
        return wL, wR, wO, wV, wc

    def verification_msm(self, proof, bp_gens: BulletproofGens, trace=None):
        """verifier.rs:393-547: returns (scalars, points) of the `mega_check` MSM in the
        exact order of verifier.rs:517-546.  Raises VerificationError on the
        identity-point checks (transcript.rs:101-113)."""
        tr = self.transcript
        tr.append_u64(b"m", len(self.V))
        n1 = self.num_vars
        tr.validate_and_append_point(b"A_I1", proof["A_I1"])
        tr.validate_and_append_point(b"A_O1", proof["A_O1"])
        tr.validate_and_append_point(b"S1", proof["S1"])
        self._create_randomized_constraints()
        n = self.num_vars
        n2 = n - n1
        padded_n = 1 if n == 0 else 1 << (n - 1).bit_length()
        pad = padded_n - n
        if bp_gens.gens_capacity < padded_n:
            raise InvalidGeneratorsLength()
        tr.append_point(b"A_I2", proof["A_I2"])
        tr.append_point(b"A_O2", proof["A_O2"])
        tr.append_point(b"S2", proof["S2"])
        y = tr.challenge_scalar(b"y")
        z = tr.challenge_scalar(b"z")
        for lab in ("T_1", "T_3", "T_4", "T_5", "T_6"):
            tr.validate_and_append_point(lab.encode(), proof[lab])
        u = tr.challenge_scalar(b"u")
        x = tr.challenge_scalar(b"x")
        tr.append_scalar(b"t_x", proof["t_x"])
        tr.append_scalar(b"t_x_blinding", proof["t_x_blinding"])
        tr.append_scalar(b"e_blinding", proof["e_blinding"])
        w = tr.challenge_scalar(b"w")
        wL, wR, wO, wV, wc = self.flattened_constraints(z)
        ch = ipp_challenges(proof["L_vec"], proof["R_vec"], padded_n, tr)
        u_sq, u_inv_sq, s = verification_scalars_from_challenges(ch, padded_n)
        a, b = proof["a"], proof["b"]
        y_inv = inv(y)
        y_inv_vec = exp_iter(y_inv, padded_n)
        yneg_wR = [wR[i] * y_inv_vec[i] % N for i in range(n)] + [0] * pad
        delta = inner_product(yneg_wR[:n], wL)
        u_for_g = [1] * n1 + [u] * (n2 + pad)
        g_scalars = [u_for_g[i] * (x * yneg_wR[i] - a * s[i]) % N for i in range(padded_n)]
        wLp, wOp = wL + [0] * pad, wO + [0] * pad
        h_scalars = [u_for_g[i] * (y_inv_vec[i] * (x * wLp[i] + wOp[i] - b * s[padded_n - 1 - i]) - 1) % N
                     for i in range(padded_n)]
        r = tr.challenge_scalar(b"r")
        xx = x * x % N
        rxx = r * xx % N
        xxx = x * xx % N
        T_scalars = [r * x % N, rxx * x % N, rxx * xx % N, rxx * xxx % N, rxx * xx % N * xx % N]
        scalars = ([x, xx, xxx, u * x % N, u * xx % N, u * xxx % N]
                   + [wVi * rxx % N for wVi in wV] + T_scalars
                   + [(w * (proof["t_x"] - a * b) + r * (xx * (wc + delta) - proof["t_x"])) % N]
                   + [(-proof["e_blinding"] - r * proof["t_x_blinding"]) % N]
                   + g_scalars + h_scalars + u_sq + u_inv_sq)
        points = ([proof["A_I1"], proof["A_O1"], proof["S1"], proof["A_I2"], proof["A_O2"], proof["S2"]]
                  + self.V + [proof[k] for k in ("T_1", "T_3", "T_4", "T_5", "T_6")]
                  + [self.pc_gens.B, self.pc_gens.B_blinding]
                  + bp_gens.G(padded_n) + bp_gens.H(padded_n) + proof["L_vec"] + proof["R_vec"])
        if trace is not None:
            trace.update(dict(y=y, z=z, u=u, x=x, w=w, r=r, ipp_u=ch, n1=n1, n2=n2, padded_n=padded_n,
                              wL=wL, wR=wR, wO=wO, wV=wV, wc=wc, s=s, u_sq=u_sq, u_inv_sq=u_inv_sq,
                              delta=delta))
        return scalars, points

    def verify(self, proof, bp_gens, trace=None):
        """verifier.rs:393-554: True iff mega_check is the identity."""
        try:
            scalars, points = self.verification_msm(proof, bp_gens, trace)
        except VerificationError:
            return False
        mega = msm(scalars, points)
        if trace is not None:
            trace["mega_check"] = mega
        return mega is INF


# --------------------------------------------------------------------------- gadgets (tests/r1cs.rs)
def range_proof_gadget(cs, v_lc, v_assignment, n_bits):
    """tests/r1cs.rs:620-652."""
    exp_2 = 1
    v = dict(v_lc)
    for i in range(n_bits):
        if v_assignment is not None:
            bit = (v_assignment >> i) & 1
            a, b, o = cs.allocate_multiplier((1 - bit, bit))
        else:
            a, b, o = cs.allocate_multiplier(None)
        cs.constrain(lc_var(o))
        cs.constrain(lc_add(lc_var(a), lc_sub(lc_var(b), lc_const(1))))
        v = lc_sub(v, {b: exp_2})
        exp_2 = (exp_2 + exp_2) % N
    cs.constrain(v)


def shuffle_gadget(cs, x, y):
    """tests/r1cs.rs:23-62."""
    assert len(x) == len(y)
    k = len(x)
    if k == 1:
        cs.constrain(lc_sub(lc_var(y[0]), lc_var(x[0])))
        return

    def cb(cs):
        z = cs.challenge_scalar(b"shuffle challenge")
        mz = lc_const(-z)
        _, _, last = cs.multiply(lc_add(lc_var(x[k - 1]), mz), lc_add(lc_var(x[k - 2]), mz))
        for i in range(k - 3, -1, -1):
            _, _, last = cs.multiply(lc_var(last), lc_add(lc_var(x[i]), mz))
        first_x = last
        _, _, last = cs.multiply(lc_add(lc_var(y[k - 1]), mz), lc_add(lc_var(y[k - 2]), mz))
        for i in range(k - 3, -1, -1):
            _, _, last = cs.multiply(lc_var(last), lc_add(lc_var(y[i]), mz))
        cs.constrain(lc_sub(lc_var(first_x), lc_var(last)))

    cs.specify_randomized_constraints(cb)


def example_gadget(cs, a1, a2, b1, b2, c1, c2):
    """tests/r1cs.rs:217-228."""
    _, _, c_var = cs.multiply(lc_add(a1, a2), lc_add(b1, b2))
    cs.constrain(lc_sub(lc_add(c1, c2), lc_var(c_var)))


# --------------------------------------------------------------------------- wire codec
# R1CSProof::{to,from}_bytes (src/r1cs/proof.rs:82-207), InnerProductProof::{to,from}_bytes
# (src/inner_product_proof.rs:379-455).  Scalars: 32 bytes big-endian, read back mod n
# (`from_be_bytes_mod_order`).  Points: StarkPoint::to_bytes() is defined in the absent crate mpc-stark = "0.2";
# restated here as arkworks' compressed short-Weierstrass encoding (ark-serialize 0.4, which mpc-stark builds
# on): x as 32 little-endian bytes with two flag bits in the top of the last byte -- bit 7: y is the
# lexicographically larger of (y, -y); bit 6: point at infinity.  PARITY UNPINNED for the point bytes (no file
# or test of the reference fixes them); everything else follows the reference line by line.
ONE_PHASE_COMMITMENTS = 0
TWO_PHASE_COMMITMENTS = 1


class FormatError(Exception):
    """ProofError::FormatError / R1CSError::FormatError"""


def fp_sqrt(a: int):
    """Tonelli-Shanks in F_p (p - 1 = 2^192 * (2^59 + 17)); None for a non-residue."""
    a %= P
    if a == 0:
        return 0
    if pow(a, (P - 1) // 2, P) != 1:
        return None
    s, t = 192, (P - 1) >> 192
    z = pow(3, t, P)                      # 3 is the smallest quadratic non-residue
    x, b, m = pow(a, (t + 1) // 2, P), pow(a, t, P), s
    while b != 1:
        i, bb = 0, b
        while bb != 1:
            bb = bb * bb % P
            i += 1
        g = pow(z, 1 << (m - i - 1), P)
        x, z, b, m = x * g % P, g * g % P, b * g * g % P, i
    assert x * x % P == a
    return x


def point_compress(Pt) -> bytes:
    if Pt is INF:
        return bytes(31) + b"\x40"
    x, y = Pt
    b = bytearray(x.to_bytes(32, "little"))
    if y > P - y:
        b[31] |= 0x80
    return bytes(b)


def point_decompress(b: bytes):
    if len(b) != 32:
        raise FormatError("length")
    flags = b[31] >> 6
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
    if flags == 3 or x >= P:
        raise FormatError("flags / non-canonical x")
    if flags == 1:
        return INF
    y = fp_sqrt((x * x * x + CURVE_A * x + CURVE_B) % P)
    if y is None:
        raise FormatError("x is not on the curve")
    if (y > P - y) != (flags == 2):
        y = P - y
    return (x, y)


def scalar_to_bytes_be(s: int) -> bytes:
    return (s % N).to_bytes(32, "big")


def ipp_to_bytes(L_vec, R_vec, a, b) -> bytes:
    out = b""
    for l, r in zip(L_vec, R_vec):
        out += point_compress(l) + point_compress(r)
    return out + scalar_to_bytes_be(a) + scalar_to_bytes_be(b)


def ipp_from_bytes(s: bytes):
    """inner_product_proof.rs:419-455"""
    if len(s) < 64 or len(s) % 32:
        raise FormatError("length")
    num_points = (len(s) - 64) // 32
    if num_points % 2:
        raise FormatError("odd number of points")
    lg_n = num_points // 2
    if lg_n >= 32:
        raise FormatError("too big")
    L = [point_decompress(s[64 * i:64 * i + 32]) for i in range(lg_n)]
    R = [point_decompress(s[64 * i + 32:64 * i + 64]) for i in range(lg_n)]
    pos = 64 * lg_n
    return L, R, int.from_bytes(s[pos:pos + 32], "big") % N, int.from_bytes(s[pos + 32:pos + 64], "big") % N


_PROOF_POINTS_1 = ("A_I1", "A_O1", "S1")
_PROOF_POINTS_2 = ("A_I2", "A_O2", "S2")
_PROOF_POINTS_T = ("T_1", "T_3", "T_4", "T_5", "T_6")


def r1cs_proof_to_bytes(p: dict) -> bytes:
    """r1cs/proof.rs:82-123; p has the keys of Prover.prove()'s result"""
    one_phase = all(p[k] is INF for k in _PROOF_POINTS_2)
    out = bytes([ONE_PHASE_COMMITMENTS if one_phase else TWO_PHASE_COMMITMENTS])
    for k in _PROOF_POINTS_1 + (() if one_phase else _PROOF_POINTS_2) + _PROOF_POINTS_T:
        out += point_compress(p[k])
    for k in ("t_x", "t_x_blinding", "e_blinding"):
        out += scalar_to_bytes_be(p[k])
    return out + ipp_to_bytes(p["L_vec"], p["R_vec"], p["a"], p["b"])


def r1cs_proof_from_bytes(s: bytes) -> dict:
    """r1cs/proof.rs:128-207"""
    if not s:
        raise FormatError("empty")
    version, s = s[0], s[1:]
    if len(s) % 32:
        raise FormatError("length")
    if version not in (ONE_PHASE_COMMITMENTS, TWO_PHASE_COMMITMENTS):
        raise FormatError("version")
    if len(s) < (11 if version == ONE_PHASE_COMMITMENTS else 14) * 32:
        raise FormatError("short")
    p, pos = {}, 0

    def rd_point():
        nonlocal pos
        v = point_decompress(s[pos:pos + 32])
        pos += 32
        return v
    for k in _PROOF_POINTS_1:
        p[k] = rd_point()
    for k in _PROOF_POINTS_2:
        p[k] = INF if version == ONE_PHASE_COMMITMENTS else rd_point()
    for k in _PROOF_POINTS_T:
        p[k] = rd_point()
    for k in ("t_x", "t_x_blinding", "e_blinding"):
        p[k] = int.from_bytes(s[pos:pos + 32], "big") % N
        pos += 32
    p["L_vec"], p["R_vec"], p["a"], p["b"] = ipp_from_bytes(s[pos:])
    return p
