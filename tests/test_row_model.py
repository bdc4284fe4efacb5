"""The carry-free two-block Montgomery reduction of the row-distributed multiplication (csrc/ec29_row.cuh), on the lane-level
Python model: results = a * b / 2^261 mod p for tight, signed-lazy and wide operands, every 64-bit / 32-bit intermediate inside
its register, output limbs and values inside the bounds the header promises."""
import row_model as rm


def test_row_multiplication_model_and_bounds():
    st = rm.self_test(count=3000, seed=5)
    assert st["maxlimb"] < (1 << 29) + (1 << 24) and st["maxtop"] < (1 << 22)


def test_row_multiplication_extreme_limbs():
    """all limbs at the documented extremes: +-(2^29 + 2^25) against +-(2^30 + 2^25), top limbs +-2^23"""
    la, lb, top = (1 << 29) + (1 << 25), (1 << 30) + (1 << 25), 1 << 23
    rinv = pow(rm.R, -1, rm.P)
    for sa in (1, -1):
        for sb in (1, -1):
            a = [sa * la] * 8 + [sa * top] + [0] * 7
            b = [sb * lb] * 8 + [-sb * top] + [0] * 7
            r = rm.rmul(a, b, {})
            assert (rm.val(r) - rm.val(a) * rm.val(b) * rinv) % rm.P == 0
            alt = [(-1) ** j * sa * la for j in range(8)] + [top] + [0] * 7
            r = rm.rmul(alt, b, {})
            assert (rm.val(r) - rm.val(alt) * rm.val(b) * rinv) % rm.P == 0
