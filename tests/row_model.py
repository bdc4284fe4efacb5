"""Lane-level model of the row-distributed Montgomery multiplication of mpc_bulletproof_amd/csrc/ec29_row.cuh (16 lanes of a DPP
row, limb j in lane j): every DPP move (row_shr / row_shl with zero fill, row_newbcast), every 64-bit accumulator and every
32-bit intermediate of `rmul` restated on Python integers with range assertions -- the algorithm and its overflow bounds are
checked here without a GPU (tests/test_row_model.py); the device code is checked against big integers in tests/test_gpu_arith.py.
Test infrastructure."""
import random
P = 2**251 + 17*2**192 + 1
LB, M = 29, (1 << 29) - 1
R = 1 << 261
P6, P8 = 17 << 18, 1 << 19
NLANE = 16

def shr(v, n): return [v[l - n] if l - n >= 0 else 0 for l in range(NLANE)]
def shl(v, n): return [v[l + n] if l + n < NLANE else 0 for l in range(NLANE)]
def bc(v, n): return [v[n]] * NLANE
def s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x
def s32(x):
    x &= (1 << 32) - 1
    return x - (1 << 32) if x >> 31 else x
def chk64(x): assert -(1 << 63) <= x < (1 << 63), x; return x
def chk32(x): assert -(1 << 31) <= x < (1 << 31), x; return x

def to_lanes(x):  # tight limbs
    v = [(x >> (LB * j)) & M for j in range(8)] + [x >> (LB * 8)] + [0] * 7
    return v
def val(v): return sum(v[j] << (LB * j) for j in range(9))

def rmul(a, b, stats=None):
    LO = [0] * NLANE; HI = [0] * NLANE
    for i in range(9):
        s = bc(a, i)
        if i <= 5:
            t = shr(b, i)
            LO = [chk64(LO[l] + s[l] * t[l]) for l in range(NLANE)]
        t = shl(b, 6 - i) if i < 6 else (b if i == 6 else shr(b, i - 6))
        HI = [chk64(HI[l] + s[l] * t[l]) for l in range(NLANE)]
    le5 = [l <= 5 for l in range(NLANE)]
    l = [(LO[k] & M) if le5[k] else 0 for k in range(NLANE)]
    h = [((LO[k] >> 29) & M) if le5[k] else 0 for k in range(NLANE)]
    g = [(LO[k] >> 58) if le5[k] else 0 for k in range(NLANE)]
    h1, g2 = shr(h, 1), shr(g, 2)
    d = [l[k] + h1[k] + g2[k] for k in range(NLANE)]
    m1 = [d[k] if le5[k] else 0 for k in range(NLANE)]
    h5, g4 = shl(h, 5), shl(g, 4)
    cin = [h5[k] + g4[k] for k in range(NLANE)]
    m1s = shr(m1, 2)
    HI = [chk64(HI[k] + cin[k] - P6 * m1[k] - P8 * m1s[k]) for k in range(NLANE)]
    assert all(HI[k] == 0 for k in range(11, NLANE))
    lp = [HI[k] & M for k in range(NLANE)]
    hp = [((HI[k] >> 29) & M) if k < 10 else s32(HI[k] >> 29) for k in range(NLANE)]
    if stats is not None: assert -(1 << 31) <= (HI[10] >> 29) < (1 << 31)
    gp = [HI[k] >> 58 for k in range(NLANE)]
    h1, g2 = shr(hp, 1), shr(gp, 2)
    dp = [chk32(lp[k] + h1[k] + g2[k]) for k in range(NLANE)]
    m2 = [dp[k] if k <= 2 else 0 for k in range(NLANE)]
    dm = [dp[k] - m2[k] for k in range(NLANE)]
    m6, m8 = shr(m2, 6), shr(m2, 8)
    U = [chk64(dm[k] - P6 * m6[k] - P8 * m8[k]) for k in range(NLANE)]
    lo = [(U[k] & M) if 3 <= k <= 10 else (s32(U[k]) if k == 11 else 0) for k in range(NLANE)]
    if stats is not None: assert -(1 << 31) <= U[11] < (1 << 31)
    hc = [chk32(U[k] >> 29) if 3 <= k <= 10 else 0 for k in range(NLANE)]
    l3, h2 = shl(lo, 3), shl(hc, 2)
    r = [chk32(l3[k] + h2[k]) for k in range(NLANE)]
    assert all(r[k] == 0 for k in range(9, NLANE))
    if stats is not None:
        stats['maxlimb'] = max(stats.get('maxlimb', 0), max(abs(x) for x in r[:8]))
        stats['maxtop'] = max(stats.get('maxtop', 0), abs(r[8]))
    return r

def self_test(count=20000, seed=1):
    rnd = random.Random(seed)
    Rinv = pow(R, -1, P)
    st = {}
    for it in range(count):
        if it % 4 == 0: x, y = rnd.randrange(P), rnd.randrange(P)
        elif it % 4 == 1: x, y = P - 1 - rnd.randrange(3), P - 1 - rnd.randrange(3)
        elif it % 4 == 2: x, y = rnd.randrange(1 << 256), rnd.randrange(1 << 256)
        else: x, y = rnd.randrange(4), rnd.randrange(P)
        a, b = to_lanes(x), to_lanes(y)
        r = rmul(a, b, st)
        v = val(r)
        assert (v - x * y * Rinv) % P == 0, it
        assert -2.2 * P < v < 1.1 * P, (it, v / P)
    # signed lazy inputs: differences of two tight values (limbs in (-2^29, 2^29))
    for it in range(count):
        x1, x2, y1, y2 = (rnd.randrange(P) for _ in range(4))
        a = [p - q for p, q in zip(to_lanes(x1), to_lanes(x2))]
        b = [p - q for p, q in zip(to_lanes(y1), to_lanes(y2))]
        r = rmul(a, b, st)
        assert (val(r) - (x1 - x2) * (y1 - y2) * Rinv) % P == 0
    # one side up to 2^30.6 (sum of three) against tight
    for it in range(count):
        x1, x2, x3, y1 = (rnd.randrange(1 << 252) for _ in range(4))
        a = [p + q + s for p, q, s in zip(to_lanes(x1), to_lanes(x2), to_lanes(x3))]
        b = to_lanes(y1)
        r = rmul(a, b, st)
        assert (val(r) - (x1 + x2 + x3) * y1 * Rinv) % P == 0
    return st


if __name__ == "__main__":
    print("ok", self_test())
