"""The DEVICE code path of the arithmetic headers (csrc/fe29.cuh: asm MAD chains; csrc/ec29.cuh: select-resolved
identity operands, out-of-line exact branches) run on the GPU one element per lane (tests/csrc/fe29_gpu_test.hip,
built by __graft_entry__.build()) and compared with Python big integers / the Python model -- the GPU twin of
test_device_arith_cpu.py, plus products of adversarial RAW limb vectors (the representation bounds the
multiplication promises to accept)."""
import ctypes as C
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pymodel as pm   # noqa: E402

pytestmark = pytest.mark.gpu
P, N = pm.P, pm.N
R = 1 << 261
NL, LB = 9, 29


@pytest.fixture(scope="module")
def g():
    so = os.path.join(HERE, "csrc", "libfe29_gpu.so")
    assert os.path.exists(so), "tests/csrc/libfe29_gpu.so missing: run __graft_entry__.build()"
    return C.CDLL(so)


def le(x):
    return x.to_bytes(32, "little")


def field_cases(m, rnd, extra):
    edge = [0, 1, 2, m - 1, m - 2, (m + 1) // 2, 2**128 - 1, 2**250]
    return [(a % m, b % m) for a in edge for b in edge] + [(rnd.randrange(m), rnd.randrange(m)) for _ in range(extra)]


@pytest.mark.parametrize("field,m", [(0, P), (1, N)])
def test_field_ops_on_device(g, field, m):
    rnd = random.Random(129 + field)
    want = {0: lambda a, b: a + b, 1: lambda a, b: a - b, 2: lambda a, b: a * b, 3: lambda a, b: a * a,
            4: lambda a, b: pow(a, m - 2, m), 5: lambda a, b: -a, 6: lambda a, b: 8 * a,
            7: lambda a, b: a * a - b * b, 8: lambda a, b: (a - 2 * b) ** 2, 9: lambda a, b: 3 * a * (b - a),
            10: lambda a, b: pow(a, m - 2, m)}
    cases = field_cases(m, rnd, 2000)
    n = len(cases)
    a = (C.c_uint8 * (32 * n)).from_buffer_copy(b"".join(le(x) for x, _ in cases))
    b = (C.c_uint8 * (32 * n)).from_buffer_copy(b"".join(le(y) for _, y in cases))
    out, rc = (C.c_uint8 * (32 * n))(), (C.c_int * n)()
    for op, fn in want.items():
        assert g.g29_field(field, op, a, b, C.c_size_t(n), out, rc) == 0, "no HIP device"
        assert not any(rc)
        ob = bytes(out)
        for i, (x, y) in enumerate(cases):
            assert int.from_bytes(ob[32 * i:32 * i + 32], "little") == fn(x, y) % m, (field, op, hex(x), hex(y))
    # non-canonical input is rejected
    bad = (C.c_uint8 * 32).from_buffer_copy(le(m))
    one = (C.c_uint8 * 32).from_buffer_copy(le(1))
    assert g.g29_field(field, 2, bad, one, C.c_size_t(1), out, rc) == 0 and rc[0] == -1


def _limbs_value(v):
    return sum(x << (LB * j) for j, x in enumerate(v))


def _raw_cases(rnd, count):
    """limb vectors within the documented input bounds of mul / sqr: lower limbs in T' = [-8, 2^29 + 8) or up to
    1.5 * 2^29 in magnitude (either sign) against a T' partner, top limb small signed, |value| < 2^256"""
    lo_t, hi_t = -8, (1 << LB) + 7
    big = 3 << (LB - 1)           # 1.5 * 2^29
    top = (1 << 23)

    def tprime():
        return [rnd.choice([lo_t, hi_t, 0, rnd.randint(lo_t, hi_t)]) for _ in range(NL - 1)] + [rnd.randint(-top, top)]

    def wide():
        return [rnd.choice([-big, big, rnd.randint(-big, big)]) for _ in range(NL - 1)] + [rnd.randint(-top, top)]

    cases = [([hi_t] * 8 + [top], [hi_t] * 8 + [top]), ([lo_t] * 8 + [-top], [hi_t] * 8 + [top]),
             ([big] * 8 + [top], [hi_t] * 8 + [top]), ([-big] * 8 + [-top], [hi_t] * 8 + [-top]),
             ([0] * 9, [hi_t] * 8 + [top]), ([1] + [0] * 8, [1] + [0] * 8)]
    for _ in range(count):
        cases.append((tprime(), tprime()))
        cases.append((wide(), tprime()))
    return cases


@pytest.mark.parametrize("field,m", [(0, P), (1, N)])
def test_raw_limb_products_at_the_representation_bounds(g, field, m):
    rnd = random.Random(229 + field)
    cases = _raw_cases(rnd, 1500)
    n = len(cases)
    rinv = pow(R, -1, m)
    a = (C.c_int32 * (NL * n))(*[x for u, _ in cases for x in u])
    b = (C.c_int32 * (NL * n))(*[x for _, v in cases for x in v])
    out = (C.c_uint8 * (32 * n))()
    assert g.g29_rawmul(field, 0, a, b, C.c_size_t(n), out) == 0, "no HIP device"
    ob = bytes(out)
    for i, (u, v) in enumerate(cases):
        assert int.from_bytes(ob[32 * i:32 * i + 32], "little") == _limbs_value(u) * _limbs_value(v) * rinv % m, (field, u, v)
    # squares: T' inputs only (a square never sees doubled limbs)
    sq = [(u, u) for u, v in cases[:6] if max(abs(x) for x in u[:8]) <= (1 << LB) + 8] + [c for c in cases[6::2]]
    sq = [(u, u) for u, _ in sq]
    n = len(sq)
    a = (C.c_int32 * (NL * n))(*[x for u, _ in sq for x in u])
    assert g.g29_rawmul(field, 1, a, a, C.c_size_t(n), out) == 0
    ob = bytes(out)
    for i, (u, _) in enumerate(sq):
        assert int.from_bytes(ob[32 * i:32 * i + 32], "little") == _limbs_value(u) ** 2 * rinv % m, (field, u)


def test_point_ops_including_exceptional_cases_on_device(g):
    rnd = random.Random(131)
    pts = [pm.INF, pm.G, pm.pt_neg(pm.G), pm.pt_mul(2, pm.G), pm.pt_mul(N - 2, pm.G)] + \
          [pm.pt_mul(rnd.randrange(1, N), pm.G) for _ in range(11)]
    pairs = [(a, b) for a in pts for b in pts]          # includes P + P, P + (-P), identity on either side
    n = len(pairs)
    a = (C.c_uint8 * (64 * n)).from_buffer_copy(b"".join(pm.p2b(x) for x, _ in pairs))
    b = (C.c_uint8 * (64 * n)).from_buffer_copy(b"".join(pm.p2b(y) for _, y in pairs))
    out, rc = (C.c_uint8 * (64 * n))(), (C.c_int * n)()
    for op in (0, 1, 2, 3, 4):       # 3, 4: the extended-Jacobian accumulator's mixed addition (fixed-base walks, window sums)
        assert g.g29_point(op, a, b, C.c_size_t(n), out, rc) == 0, "no HIP device"
        ob = bytes(out)
        for i, (x, y) in enumerate(pairs):
            if op == 4 and y is pm.INF:
                assert rc[i] == -3       # xyzz_madd_nzq is for non-identity addends only
                continue
            assert rc[i] == 0
            want = pm.pt_add(x, x) if op == 2 else pm.pt_add(x, y)
            assert pm.b2p(ob[64 * i:64 * i + 64]) == want, (op, x, y)
    # off-curve input is rejected
    bad = bytearray(pm.p2b(pm.G))
    bad[0] ^= 1
    ba = (C.c_uint8 * 64).from_buffer_copy(bytes(bad))
    gg = (C.c_uint8 * 64).from_buffer_copy(pm.p2b(pm.G))
    assert g.g29_point(0, ba, gg, C.c_size_t(1), out, rc) == 0 and rc[0] == -1


def test_quad_cooperative_group_law_on_device(g):
    """ec29_quad.cuh (four lanes of a DPP quad per point, modified Jacobian coordinates with the halved doubling): sums,
    doublings and a Horner stretch of 68 doublings and 3 additions equal the Python model for every pair of a point set that
    includes the identity on either side, P + P and P + (-P); T = Z^4 holds on exit."""
    rnd = random.Random(137)
    pts = [pm.INF, pm.G, pm.pt_neg(pm.G), pm.pt_mul(2, pm.G), pm.pt_mul(N - 2, pm.G)] + \
          [pm.pt_mul(rnd.randrange(1, N), pm.G) for _ in range(11)]
    pairs = [(a, b) for a in pts for b in pts]
    pairs += [(pm.pt_mul(k, pm.G), pm.pt_mul(k << 32, pm.G)) for k in (1, 5)]     # the chain meets its own operand
    while len(pairs) % 16:
        pairs.append((pm.G, pm.G))
    n = len(pairs)
    a = (C.c_uint8 * (64 * n)).from_buffer_copy(b"".join(pm.p2b(x) for x, _ in pairs))
    b = (C.c_uint8 * (64 * n)).from_buffer_copy(b"".join(pm.p2b(y) for _, y in pairs))
    out, rc = (C.c_uint8 * (64 * n))(), (C.c_int * n)()
    for op in (0, 1, 2):
        assert g.g29_q4(op, a, b, C.c_size_t(n), out, rc) == 0, "no HIP device"
        assert not any(rc), (op, list(rc))
        ob = bytes(out)
        for i, (x, y) in enumerate(pairs):
            if op == 0:
                want = pm.pt_add(x, y)
            elif op == 1:
                want = pm.pt_add(x, x)
            else:
                acc = x
                for _ in range(2):
                    acc = pm.pt_add(pm.pt_mul(1 << 32, acc), y)
                want = pm.pt_add(pm.pt_mul(16, acc), x)
            assert pm.b2p(ob[64 * i:64 * i + 64]) == want, (op, i, x, y)


# ------------------------------------------------------------------ row-distributed arithmetic (csrc/ec29_row.cuh)
def test_row_distributed_products_at_the_representation_bounds(g):
    """rmul (one multiplication spread over the 16 lanes of a DPP row, carry-free two-block Montgomery reduction): raw limb
    vectors at the documented input bounds, result = value(a) * value(b) / 2^261 mod p; lanes 9..15 stay zero."""
    rnd = random.Random(331)
    cases = _raw_cases(rnd, 1500)
    # the row form also takes differences of products on both sides: limbs up to +-(2^29 + 2^25) against +-(2^30 + 2^25)
    lim_a, lim_b = (1 << 29) + (1 << 25), (1 << 30) + (1 << 25)
    for _ in range(500):
        cases.append(([rnd.choice([-lim_a, lim_a, rnd.randint(-lim_a, lim_a)]) for _ in range(8)] + [rnd.randint(-(1 << 23), 1 << 23)],
                      [rnd.choice([-lim_b, lim_b, rnd.randint(-lim_b, lim_b)]) for _ in range(8)] + [rnd.randint(-(1 << 23), 1 << 23)]))
    n = len(cases)
    rinv = pow(R, -1, P)
    a = (C.c_int32 * (NL * n))(*[x for u, _ in cases for x in u])
    b = (C.c_int32 * (NL * n))(*[x for _, v in cases for x in v])
    out = (C.c_uint8 * (32 * n))()
    assert g.g29_rowmul(a, b, C.c_size_t(n), out) == 0, "no HIP device"
    ob = bytes(out)
    for i, (u, v) in enumerate(cases):
        assert int.from_bytes(ob[32 * i:32 * i + 32], "little") == _limbs_value(u) * _limbs_value(v) * rinv % P, (i, u, v)


def test_row_moves_and_helpers(g):
    """rbc4 (v_permlane16_swap / v_permlane32_swap: row r of a register in every row), rgather / rscatter, the distributed
    halving and parallel carry"""
    rnd = random.Random(337)
    els = []
    for r in range(4):
        els.append([rnd.randint(-(1 << 30), 1 << 30) for _ in range(8)] + [rnd.randint(-(1 << 22), 1 << 22)])
    a = (C.c_int32 * 36)(*[x for e in els for x in e])
    out = (C.c_int32 * 448)()
    assert g.g29_rowprobe(a, out) == 0, "no HIP device"
    o = list(out)
    for r in range(4):
        assert o[64 * r:64 * r + 64] == [16 * r + (t & 15) for t in range(64)], ("rbc4", r)
    assert all(o[256:320])
    for r in range(4):
        v = _limbs_value(els[r])
        half = o[320 + 16 * r:320 + 16 * r + 16]
        assert half[9:] == [0] * 7 and (2 * _limbs_value(half[:9]) - v) % P == 0, ("half", r)
        assert max(abs(x) for x in half[:8]) < (1 << 29) + (1 << 28) + (1 << 23)
        nm = o[384 + 16 * r:384 + 16 * r + 16]
        assert nm[9:] == [0] * 7 and _limbs_value(nm[:9]) == v and all(-4 <= x < (1 << 29) + 4 for x in nm[:8]), ("norm", r)


def test_row_distributed_group_law_on_device(g):
    """ec29_row.cuh (one point per wave: a row per product of a level): sums, doublings and a Horner stretch of 68 doublings and
    3 additions equal the Python model for every pair of a point set with the identity on either side, P + P and P + (-P);
    Th = -Z^4 / 2 holds on exit, the four rows agree and lanes 9..15 hold zero."""
    rnd = random.Random(139)
    pts = [pm.INF, pm.G, pm.pt_neg(pm.G), pm.pt_mul(2, pm.G), pm.pt_mul(N - 2, pm.G)] + \
          [pm.pt_mul(rnd.randrange(1, N), pm.G) for _ in range(11)]
    pairs = [(a, b) for a in pts for b in pts]
    pairs += [(pm.pt_mul(k, pm.G), pm.pt_mul(k << 32, pm.G)) for k in (1, 5)]     # the chain meets its own operand
    n = len(pairs)
    a = (C.c_uint8 * (64 * n)).from_buffer_copy(b"".join(pm.p2b(x) for x, _ in pairs))
    b = (C.c_uint8 * (64 * n)).from_buffer_copy(b"".join(pm.p2b(y) for _, y in pairs))
    out, rc = (C.c_uint8 * (64 * n))(), (C.c_int * n)()
    for op in (0, 1, 2, 3):
        assert g.g29_rowpoint(op, a, b, C.c_size_t(n), out, rc) == 0, "no HIP device"
        assert not any(rc), (op, list(rc))
        ob = bytes(out)
        for i, (x, y) in enumerate(pairs):
            if op == 0:
                want = pm.pt_add(x, y)
            elif op == 1:
                want = pm.pt_add(x, x)
            elif op == 3:        # identity accumulator (empty top windows) meeting addends that carry no Th
                want = pm.pt_add(pm.pt_mul(32, y), x)
            else:
                acc = x
                for _ in range(2):
                    acc = pm.pt_add(pm.pt_mul(1 << 32, acc), y)
                want = pm.pt_add(pm.pt_mul(16, acc), x)
            assert pm.b2p(ob[64 * i:64 * i + 64]) == want, (op, i, x, y)


def test_doubling_chain_latency_row_against_quad(g, capsys):
    """252 dependent doublings (+ 7 additions) on ONE wave: both forms give 2^252-ish multiples of G equal to the model; the row
    form must be the faster chain (printed: microseconds per form)."""
    g.g29_chain.restype = C.c_double
    a = (C.c_uint8 * 64).from_buffer_copy(pm.p2b(pm.G))
    out = (C.c_uint8 * 64)()
    nd = 252
    want = pm.G
    for d in range(nd):
        want = pm.pt_add(want, want)
        if d & 31 == 31:
            want = pm.pt_add(want, pm.G)
    us = {}
    for form in (0, 1):
        t = g.g29_chain(form, nd, a, out)
        assert t > 0, "no HIP device"
        assert pm.b2p(bytes(out)) == want, form
        us[form] = t
    with capsys.disabled():
        print(f"\n[chain] 252 doublings + 7 additions on one wave: quad form {us[0]:.1f} us, row form {us[1]:.1f} us")
    assert us[1] < us[0]
