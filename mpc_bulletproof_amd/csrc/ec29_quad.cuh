// ec29_quad.cuh -- QUAD-COOPERATIVE point arithmetic: the four lanes of a DPP quad work on ONE point.
//
// Why: every variable-base MSM ends in a chain of ~252 dependent doublings (Horner over the windows) that no
// arrangement of the work can shorten -- 2^251 * P needs 251 sequential group operations -- and a lane executes a
// doubling as 9 dependent field multiplications (1 200 instructions, 2.5 us at one wave per SIMD): 0.6 ms of pure
// latency per batch / per MSM, with the chip idle.  Here the independent multiplications of one group operation run in
// different lanes of a quad (v_mov_b32_dpp quad_perm moves an operand between them in one instruction per limb), so
// a doubling is 3 multiplications deep instead of 9 and a general addition 5 deep instead of 16.
//
// Coordinates: modified Jacobian (X : Y : Z : Th) with Th = -Z^4 / 2 (a = 1), the point is (X / Z^2, Y / Z^3); the identity
// has all limbs of Z and Th literally zero.  Doubling returns the representative scaled by lambda = 1/2, which removes
// every small constant of dbl-2007-bl:
//     E = (3 X^2 + Z^4) / 2 = X (3X/2) - Th,  X3 = E^2 - 2 X Y^2,  Y3 = E (X Y^2 - X3) - Y^4,  Z3 = Y Z,  Th3 = Y^4 Th
// (= (X3'/4, Y3'/8, Z3'/2) of the textbook result, the same point).  Levels: {X (3X/2), Y^2, Y Z} -> {E^2, X Y^2, Y^4,
// (2X) Y^2} -> {E (..), Y^4 Th}.  Every sum between two levels is a DIFFERENCE of two products (limbs in (-2^29, 2^29),
// signed limbs are what fe29.cuh multiplies) or feeds one side of a product only (|limb| < 2^30.6 against a tight
// operand keeps the 9-term column sums below 2^63): no carry pass inside a doubling.  3X/2 = X + X/2 is the one halving,
// taken BEFORE the product (E^2 could not absorb a sum of three terms); 2 X Y^2 is its own product in the lane that
// idles at level 2, so that X3 is a plain difference.
//
// Convention: on entry and exit of every q4_* function the whole state is REPLICATED in the four lanes of the quad;
// `role` = lane & 3.  All four lanes of a quad must be active together (they take every branch together: the branch
// conditions below are computed from replicated values).  The curve has prime order, so there is no point with Y = 0.
// Device only (DPP); tests/csrc/fe29_gpu_test.hip compares every function with the one-lane group law of ec29.cuh.
#pragma once
#include "ec29.cuh"

#if defined(__HIPCC__)
namespace bp {

struct JacT { Fp X, Y, Z, Th; };   // Th = -Z^4 / 2; X, Y: signed limbs in (-2^29, 2^29); Z, Th: tight

// quad_perm move.  bound_ctrl stays 0 ON PURPOSE: with bound_ctrl = 1 the compiler folds the move into the consuming
// subtraction, and for `a - dpp(b)` it emits v_subrev_u32_dpp, which this toolchain (ROCm 7.2) assembles so that the
// permutation lands on the OTHER operand (measured on gfx950: v_subrev_u32_dpp d, x, y quad_perm:[3,3,3,3] returns
// y[lane 3] - x, not y - x[lane 3]; v_sub_u32_dpp is fine) -- every difference below has the moved operand on the right.
template <int CTRL> __device__ __forceinline__ Fp fp_dpp(const Fp &a) {
  Fp r;
#pragma unroll
  for (int j = 0; j < NL; j++) {
#if defined(__HIP_DEVICE_COMPILE__)
    r.v[j] = __builtin_amdgcn_update_dpp(a.v[j], a.v[j], CTRL, 0xf, 0xf, false);
#else
    r.v[j] = a.v[j];   // (the host pass only parses device functions)
#endif
  }
  return r;
}
constexpr int QBC0 = 0x00, QBC1 = 0x55, QBC2 = 0xAA, QBC3 = 0xFF;   // quad_perm broadcast of lane k: k * 0b01010101
__device__ __forceinline__ Fp fp_sel(bool c, const Fp &a, const Fp &b) {
  Fp r;
#pragma unroll
  for (int j = 0; j < NL; j++) r.v[j] = c ? a.v[j] : b.v[j];
  return r;
}
// v / 2 mod p for limbs |v_j| <= 2^30 (value even: exact halving; odd: (v + p) / 2).  Output limbs < 2^29 + 2^28 + 2^22,
// to be normalised by the caller.  p = [1, 0,0,0,0,0, 17 << 18, 0, 1 << 19].
__device__ __forceinline__ Fp fp_half_nr(const Fp &x) {
  Fp s = x;
  const int32_t odd = -(x.v[0] & 1);          // all ones when the value is odd (its parity is that of limb 0)
  s.v[0] += odd & 1;
  s.v[6] += odd & (17 << 18);
  s.v[8] += odd & (1 << 19);
  Fp r;
#pragma unroll
  for (int j = 0; j < NL - 1; j++) r.v[j] = (s.v[j] >> 1) + ((s.v[j + 1] & 1) << (LB - 1));
  r.v[NL - 1] = s.v[NL - 1] >> 1;
  return r;
}

__device__ __forceinline__ JacT jact_inf() {
  JacT r;
  r.X = fe_one<FP>(); r.Y = fe_one<FP>(); r.Z = fe_zero<FP>(); r.Th = fe_zero<FP>();
  return r;
}
__device__ __forceinline__ bool jact_is_inf(const JacT &p) { return is_zero_limbs(p.Z); }
// -w / 2 for a tight w, limbs in (-2^28 - 2^22, 2^28 + 2^22): one side of a product
__device__ __forceinline__ Fp fp_neg_half_nr(const Fp &w) { return fp_half_nr(sub_nr(fe_zero<FP>(), w)); }
// one-lane helpers (every lane of the quad does the same): Jacobian <-> modified Jacobian
__device__ __forceinline__ JacT jact_from_jac(const Jac &p) {
  JacT r;
  r.X = p.X; r.Y = p.Y; r.Z = p.Z;
  const Fp zz = sqr(p.Z);
  r.Th = mul(zz, fp_neg_half_nr(zz));       // zero limbs in, zero limbs out
  return r;
}
__device__ __forceinline__ Jac jact_to_jac(const JacT &p) { Jac r; r.X = norm(p.X); r.Y = norm(p.Y); r.Z = p.Z; return r; }
__device__ __forceinline__ JacT jact_select(bool c, const JacT &a, const JacT &b) {
  JacT r;
  r.X = fp_sel(c, a.X, b.X); r.Y = fp_sel(c, a.Y, b.Y); r.Z = fp_sel(c, a.Z, b.Z); r.Th = fp_sel(c, a.Th, b.Th);
  return r;
}

// 2 P.  In: X, Y limbs in (-2^29, 2^29) (tight included); Z, Th tight.  Out: the same.
__device__ __forceinline__ JacT q4_dbl(const JacT &p, int role) {
  const bool is0 = role == 0, is2 = role == 2, is3 = role == 3;
  const Fp X15 = add_nr(p.X, fp_half_nr(p.X));          // 3X/2, |limb| < 2^30
  const Fp X2 = add_nr(p.X, p.X);                       // |limb| < 2^30
  // level 1 -- lane 0: X (3X/2), lanes 1 and 2: Y Y, lane 3: Y Z
  Fp A = fp_sel(is0, p.X, p.Y);
  Fp B = fp_sel(is0, X15, fp_sel(is3, p.Z, p.Y));
  const Fp P = mul(A, B);
  const Fp E = sub_nr(P, p.Th);                         // lane 0: (3 XX + Z^4) / 2, limbs in (-2^29, 2^29)
  // level 2 -- lane 0: E E, lane 1: X YY, lane 2: YY YY, lane 3: (2X) YY
  A = fp_sel(is0, E, fp_sel(is2, P, fp_sel(is3, X2, p.X)));
  B = fp_sel(is0, E, fp_dpp<0xA4>(P));                  // quad_perm [0, 1, 2, 2]: lane 3 <- YY of lane 2
  const Fp Q = mul(A, B);
  const Fp X3 = sub_nr(Q, fp_dpp<QBC3>(Q));             // lane 0: EE - 2 XYY
  // level 3 -- lane 0: E (XYY - X3), lane 2: Y4 Th
  A = fp_sel(is0, E, Q);
  B = fp_sel(is0, sub_nr(fp_dpp<QBC1>(Q), X3), p.Th);   // lane 0: |limb| < 2^30 against E
  const Fp R = mul(A, B);
  const Fp Y3 = sub_nr(R, fp_dpp<QBC2>(Q));             // lane 0
  JacT r;
  r.X = fp_dpp<QBC0>(X3);
  r.Y = fp_dpp<QBC0>(Y3);
  r.Z = fp_dpp<QBC3>(P);
  r.Th = fp_dpp<QBC2>(R);
  return r;
}

// P1 + P2, complete (identity operands, P1 = P2, P1 = -P2)
__device__ __forceinline__ JacT q4_add(const JacT &p1, const JacT &p2, int role) {
  const bool is0 = role == 0, is1 = role == 1, is2 = role == 2, lo = role < 2;
  // level 1 -- lane 0 (and 3): Z1 Z1, lane 1: Z2 Z2, lane 2: Z1 Z2
  Fp A = fp_sel(is1, p2.Z, p1.Z);
  Fp B = fp_sel(is1 || is2, p2.Z, p1.Z);
  const Fp P = mul(A, B);
  const Fp Z1Z1 = fp_dpp<QBC0>(P), Z2Z2 = fp_dpp<QBC1>(P);
  // level 2 -- lane 0: U1 = X1 Z2Z2, lane 1: U2 = X2 Z1Z1, lane 2: Z2 Z2Z2, lane 3: Z1 Z1Z1
  A = fp_sel(lo, fp_sel(is0, p1.X, p2.X), fp_sel(is2, p2.Z, p1.Z));
  B = fp_sel(is0 || is2, Z2Z2, Z1Z1);
  const Fp Q = mul(A, B);
  const Fp U1 = fp_dpp<QBC0>(Q), U2 = fp_dpp<QBC1>(Q);
  const Fp G = fp_dpp<0xEE>(Q);                         // quad_perm [2, 3, 2, 3]: lane 0 <- Z2^3, lane 1 <- Z1^3
  const Fp H = sub_nr(U2, U1);                          // products on both sides: limbs in (-2^29, 2^29), fine for one product
  // level 3 -- lane 0: S1 = Y1 Z2^3, lane 1: S2 = Y2 Z1^3, lane 2: Z3 = Z1Z2 H, lane 3: HH = H H
  A = fp_sel(lo, fp_sel(is0, p1.Y, p2.Y), fp_sel(is2, P, H));
  B = fp_sel(lo, G, H);
  const Fp R = mul(A, B);
  const Fp S1 = fp_dpp<QBC0>(R), S2 = fp_dpp<QBC1>(R), Z3 = fp_dpp<QBC2>(R), HH = fp_dpp<QBC3>(R);
  const Fp rr = sub_nr(S2, S1);
  const bool inf1 = jact_is_inf(p1), inf2 = jact_is_inf(p2);
  if (fp_maybe_zero(H) && !inf1 && !inf2) {             // H = 0 (mod p) needs U1 = U2: same x.  Replicated values: quad-uniform.
    if (is_zero_exact(H)) {
      if (is_zero_exact(rr)) return q4_dbl(p1, role);   // P1 = P2
      return jact_inf();                                // P1 = -P2
    }
  }
  // level 4 -- lane 0: HHH = H HH, lane 1: V = U1 HH, lane 2: rr rr, lane 3: Z3 Z3
  A = fp_sel(lo, fp_sel(is0, H, U1), fp_sel(is2, rr, Z3));
  B = fp_sel(lo, HH, A);
  const Fp W = mul(A, B);
  const Fp HHH = fp_dpp<QBC0>(W), V = fp_dpp<QBC1>(W), RR = fp_dpp<QBC2>(W);
  const Fp X3 = norm(sub_nr(sub_nr(sub_nr(RR, HHH), V), V));
  // level 5 -- lane 0: rr (V - X3), lane 1: S1 HHH, lane 3: Th3 = (Z3 Z3) (-(Z3 Z3) / 2)
  A = fp_sel(is0, rr, fp_sel(is1, S1, W));
  B = fp_sel(is0, sub_nr(V, X3), fp_sel(is1, HHH, fp_neg_half_nr(W)));
  const Fp F = mul(A, B);
  JacT r;
  r.X = X3;
  r.Y = sub(fp_dpp<QBC0>(F), fp_dpp<QBC1>(F));
  r.Z = Z3;
  r.Th = fp_dpp<QBC3>(F);
  r = jact_select(inf1, p2, r);
  return jact_select(inf2, p1, r);
}

}  // namespace bp
#endif
