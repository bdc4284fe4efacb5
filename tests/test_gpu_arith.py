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
