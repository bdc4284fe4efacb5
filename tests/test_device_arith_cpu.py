"""The device arithmetic headers (csrc/fe29.cuh, ec29.cuh, fe29_sqrt.cuh) compiled for the CPU with
-fsanitize=undefined (tests/csrc/fe29_host_test.cpp) and checked against Python big integers / the Python
model: the exact source the HIP kernels run, unit-tested without a GPU."""
import ctypes as C
import os
import random
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pymodel as pm   # noqa: E402

P, N = pm.P, pm.N


@pytest.fixture(scope="module")
def h():
    out = os.path.join(HERE, "csrc", "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libfe29_host.so")
    src = os.path.join(HERE, "csrc", "fe29_host_test.cpp")
    hdrs = [os.path.join(ROOT, "mpc_bulletproof_amd", "csrc", f) for f in ("fe29.cuh", "ec29.cuh", "fe29_sqrt.cuh", "fe29_consts.h")]
    if not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-fsanitize=undefined",
                               "-fno-sanitize-recover=undefined", "-Wno-unknown-pragmas", "-o", so, src])
    return C.CDLL(so)


def le(x):
    return x.to_bytes(32, "little")


def buf(b):
    return (C.c_uint8 * len(b)).from_buffer_copy(b)


def field_cases(m, rnd):
    edge = [0, 1, 2, m - 1, m - 2, (m + 1) // 2, 2**128 - 1, 2**250]
    return [(a % m, b % m) for a in edge for b in (edge[1], edge[3], edge[5])] + \
           [(rnd.randrange(m), rnd.randrange(m)) for _ in range(60)]


@pytest.mark.parametrize("name,m", [("h29_fp", P), ("h29_fn", N)])
def test_field_ops(h, name, m):
    rnd = random.Random(29)
    f = getattr(h, name)
    want = {0: lambda a, b: a + b, 1: lambda a, b: a - b, 2: lambda a, b: a * b, 3: lambda a, b: a * a,
            4: lambda a, b: pow(a, m - 2, m), 5: lambda a, b: -a, 6: lambda a, b: 8 * a,
            7: lambda a, b: a * a - b * b, 8: lambda a, b: (a - 2 * b) ** 2, 9: lambda a, b: 3 * a * (b - a),
            10: lambda a, b: pow(a, m - 2, m), 11: lambda a, b: pow(a, m - 2, m),
            12: lambda a, b: 0 if (a - b) % m == 0 else 1}
    out = (C.c_uint8 * 32)()
    for a, b in field_cases(m, rnd):
        for op, fn in want.items():
            assert f(op, buf(le(a)), buf(le(b)), out) == 0
            assert int.from_bytes(bytes(out), "little") == fn(a, b) % m, (name, op, hex(a), hex(b))
    assert f(2, buf(le(m)), buf(le(1)), out) == -1      # non-canonical input is rejected


def test_point_ops_including_exceptional_cases(h):
    rnd = random.Random(31)
    pts = [pm.INF, pm.G, pm.pt_neg(pm.G), pm.pt_mul(2, pm.G)] + [pm.pt_mul(rnd.randrange(1, N), pm.G) for _ in range(6)]
    out = (C.c_uint8 * 64)()
    for a in pts:
        for b in pts:
            for op in (0, 1, 3, 4):      # 3, 4: the extended-Jacobian (X : Y : ZZ : ZZZ) accumulator's mixed addition
                if op == 4 and b is pm.INF:
                    continue
                assert h.h29_point(op, buf(pm.p2b(a)), buf(pm.p2b(b)), out) == 0
                assert pm.b2p(bytes(out)) == pm.pt_add(a, b), (op, a, b)
        assert h.h29_point(2, buf(pm.p2b(a)), buf(pm.p2b(a)), out) == 0
        assert pm.b2p(bytes(out)) == pm.pt_add(a, a)


@pytest.mark.parametrize("c", [2, 4, 5, 7])
def test_windowed_scalar_mul(h, c):
    rnd = random.Random(37 + c)
    out = (C.c_uint8 * 64)()
    pt = pm.pt_mul(rnd.randrange(1, N), pm.G)
    for s in [0, 1, 2, N - 1, N - 2, 2**251, 2**252 - 1 - (2**252 - 1) % 1] + [rnd.randrange(N) for _ in range(6)]:
        s %= N
        assert h.h29_scalar_mul(c, buf(le(s)), buf(pm.p2b(pt)), out) == 0
        assert pm.b2p(bytes(out)) == pm.pt_mul(s, pt), (c, hex(s))


@pytest.mark.parametrize("c", [4, 8, 12, 13, 16])
def test_signed_window_recoding(h, c):
    rnd = random.Random(41 + c)
    digits = (C.c_int * 80)()
    for s in [0, 1, N - 1, 2**251, (1 << 252) - 1] + [rnd.randrange(N) for _ in range(40)]:
        s %= N
        n = h.h29_recode(c, buf(le(s)), digits)
        assert n == 252 // c + 1
        ds = list(digits)[:n]
        assert all(-(1 << (c - 1)) <= d <= (1 << (c - 1)) - 1 or d == (1 << (c - 1)) for d in ds)
        assert sum(d << (c * w) for w, d in enumerate(ds)) == s


def test_fp_sqrt_against_model(h):
    rnd = random.Random(43)
    out = (C.c_uint8 * 32)()
    n_sq = n_non = 0
    for a in [0, 1, 4, P - 1, 3, 2**192, 2**251] + [rnd.randrange(P) for _ in range(300)]:
        a %= P
        rc = h.h29_sqrt(buf(le(a)), out)
        want = pm.fp_sqrt(a)
        if want is None:
            assert rc == 0, hex(a)
            n_non += 1
        else:
            r = int.from_bytes(bytes(out), "little")
            assert rc == 1 and r in (want, P - want) and r * r % P == a, hex(a)
            n_sq += 1
    assert n_sq > 100 and n_non > 100
    # elements of small 2-power order exercise the zero digits of the discrete logarithm
    c = pow(3, (P - 1) >> 192, P)
    for k in (1, 2, 8, 9, 100, 184, 190, 191):
        a = pow(c, 1 << k, P)
        assert h.h29_sqrt(buf(le(a)), out) == 1
        r = int.from_bytes(bytes(out), "little")
        assert r * r % P == a
    assert h.h29_sqrt(buf(le(c)), out) == 0        # the Sylow generator itself is a non-residue
