// fe29.cuh -- F_p / F_n arithmetic for gfx950: 9 signed limbs of 29 bits, lazy Montgomery (R = 2^261).
//
// Why this shape (CDNA4 has no carry-chained multiply-add; v_mad_i64_i32 gives a full 32x32+64):
//   * limbs |x_j| <~ 2^29 in int32 registers; a 9x9 schoolbook product is exactly 81 v_mad_i64_i32 whose
//     64-bit column sums need NO carry handling (9 * 2^58 < 2^63);
//   * the product is taken column by column, each column one chain of MADs that starts from the carry of the
//     column before it (no 64-bit additions at all; F_p multiply = 134 VALU instructions, 99 of them MADs);
//   * add / sub are 9 independent VALU ops (no v_addc chains) + a 3-op/limb parallel carry;
//   * R = 2^261 is 2^10 larger than the 252-bit moduli, so a product of two values < 2^256
//     reduces to within one modulus of zero without any conditional subtraction: values stay lazy
//     (small signed multiples of m) until `canon()` at an output / equality test;
//   * p = 2^251 + 17*2^192 + 1 has limbs [1,0,0,0,0,0,17<<18,0,1<<19] and p = 1 (mod 2^29):
//     one Montgomery step for F_p is a mask and two MADs.
//
// Representation invariants ("T" = tight): lower limbs in [0, 2^29), top limb small signed.
// "T'" (after norm()): lower limbs in [-8, 2^29 + 8).  mul/sqr accept T' x T' and limbs up to
// 1.5 * 2^29 against T'; they return T.  Values: every mul input must satisfy |v| < 2^256.
//
// Replaces (device side) mpc-stark's Scalar / base-field arithmetic used at the call sites listed in
// SURVEY.md K1-K10; parity is established against oracle/ (tests/).
#pragma once
#include <stdint.h>
#include "fe29_consts.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BP_HD __host__ __device__ __forceinline__
#else
#define BP_HD inline __attribute__((always_inline))
#endif

namespace bp {

constexpr int NL = 9;
constexpr int LB = 29;
constexpr int32_t LMASK = (1 << LB) - 1;

struct FP { static constexpr bool sparse = true; };   // base field of the Stark curve
struct FN { static constexpr bool sparse = false; };  // scalar field (group order)

template <class F> struct Fe { int32_t v[NL]; };
typedef Fe<FP> Fp;
typedef Fe<FN> Fn;

template <class F> BP_HD Fe<F> fe_zero() {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL; j++) r.v[j] = 0;
  return r;
}
template <class F> BP_HD Fe<F> fe_one() {  // Montgomery form of 1
  Fe<F> r;
  if constexpr (F::sparse) { constexpr int32_t C[NL] = FP_ONE; for (int j = 0; j < NL; j++) r.v[j] = C[j]; }
  else { constexpr int32_t C[NL] = FN_ONE; for (int j = 0; j < NL; j++) r.v[j] = C[j]; }
  return r;
}
template <class F> BP_HD Fe<F> fe_r2() {
  Fe<F> r;
  if constexpr (F::sparse) { constexpr int32_t C[NL] = FP_R2; for (int j = 0; j < NL; j++) r.v[j] = C[j]; }
  else { constexpr int32_t C[NL] = FN_R2; for (int j = 0; j < NL; j++) r.v[j] = C[j]; }
  return r;
}

// parallel (non-rippling) carry: any limbs with |x_j| < 2^31 -> T'
template <class F> BP_HD Fe<F> norm(const Fe<F> &x) {
  Fe<F> r;
  r.v[0] = x.v[0] & LMASK;
#pragma unroll
  for (int j = 1; j < NL - 1; j++) r.v[j] = (x.v[j] & LMASK) + (x.v[j - 1] >> LB);
  r.v[NL - 1] = x.v[NL - 1] + (x.v[NL - 2] >> LB);
  return r;
}
// no-reduce forms: chain at most 3 tight terms before norm()
template <class F> BP_HD Fe<F> add_nr(const Fe<F> &a, const Fe<F> &b) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL; j++) r.v[j] = a.v[j] + b.v[j];
  return r;
}
template <class F> BP_HD Fe<F> sub_nr(const Fe<F> &a, const Fe<F> &b) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL; j++) r.v[j] = a.v[j] - b.v[j];
  return r;
}
template <class F> BP_HD Fe<F> add(const Fe<F> &a, const Fe<F> &b) { return norm(add_nr(a, b)); }
template <class F> BP_HD Fe<F> sub(const Fe<F> &a, const Fe<F> &b) { return norm(sub_nr(a, b)); }
template <class F> BP_HD Fe<F> neg(const Fe<F> &a) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL; j++) r.v[j] = -a.v[j];
  return norm(r);
}
// k * a for small k (|k| <= 16), through int64 so that tight limbs cannot overflow int32
template <int K, class F> BP_HD Fe<F> mul_small(const Fe<F> &a) {
  int64_t t[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) t[j] = (int64_t)a.v[j] * K;
  Fe<F> r;
  r.v[0] = (int32_t)(t[0] & LMASK);
#pragma unroll
  for (int j = 1; j < NL - 1; j++) r.v[j] = (int32_t)((t[j] & LMASK) + (t[j - 1] >> LB));
  r.v[NL - 1] = (int32_t)(t[NL - 1] + (t[NL - 2] >> LB));
  return r;
}

// one Montgomery step on column i (column i is complete when this runs)
template <class F> BP_HD void mont_step_rows(int64_t *c, int i) {
  if constexpr (F::sparse) {
    uint32_t m = (0u - (uint32_t)c[i]) & (uint32_t)LMASK;   // -p^-1 = -1 (mod 2^29)
    c[i] += (int64_t)m;                                       // p[0] = 1
    c[i + 6] += (int64_t)m * 4456448;                         // p[6] = 17 << 18
    c[i + 8] += (int64_t)m * 524288;                          // p[8] = 1 << 19
  } else {
    constexpr int32_t MOD[NL] = FN_MOD;
    uint32_t m = ((uint32_t)c[i] * FN_N0) & (uint32_t)LMASK;
#pragma unroll
    for (int j = 0; j < NL; j++) c[i + j] += (int64_t)m * MOD[j];
  }
  c[i + 1] += c[i] >> LB;
}
template <class F> BP_HD Fe<F> mont_finish_rows(int64_t *c) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL - 1; j++) {
    r.v[j] = (int32_t)(c[NL + j] & LMASK);
    c[NL + j + 1] += c[NL + j] >> LB;
  }
  r.v[NL - 1] = (int32_t)c[2 * NL - 1];
  return r;
}
// a * b / R, row by row (operand scanning): 81 MADs + reduction
template <class F> BP_HD Fe<F> mul_rows(const Fe<F> &a, const Fe<F> &b) {
  int64_t c[2 * NL];
#pragma unroll
  for (int j = 0; j < 2 * NL; j++) c[j] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
#pragma unroll
    for (int j = 0; j < NL; j++) c[i + j] += (int64_t)a.v[j] * (int64_t)b.v[i];
    mont_step_rows<F>(c, i);
  }
  return mont_finish_rows<F>(c);
}
// a * a / R, row by row: 45 MADs + reduction
template <class F> BP_HD Fe<F> sqr_rows(const Fe<F> &a) {
  int64_t c[2 * NL];
  int32_t a2[NL];
#pragma unroll
  for (int j = 0; j < 2 * NL; j++) c[j] = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) a2[j] = 2 * a.v[j];
#pragma unroll
  for (int i = 0; i < NL; i++) {
    c[2 * i] += (int64_t)a.v[i] * (int64_t)a.v[i];
#pragma unroll
    for (int j = i + 1; j < NL; j++) c[i + j] += (int64_t)a.v[i] * (int64_t)a2[j];
    mont_step_rows<F>(c, i);
  }
  return mont_finish_rows<F>(c);
}

// ---- Montgomery multiplication, column by column (product scanning) ---------------------------------
// Column k of a*b (all a_j*b_{k-j}) is accumulated by a chain of v_mad_i64_i32 whose first addend is the carry of
// column k-1, so no separate 64-bit addition is ever issued; the reduction terms of the earlier columns join the same
// chain.  One column costs its MADs + one mask + one 64-bit shift.
//   F_p (sparse): p = 1 (mod 2^29), so the SUBTRACTIVE form T - M*p needs no multiplication for m_k and no addition
//   to clear the column: m_k = c_k & MASK, and (c_k - m_k) >> 29 == c_k >> 29 (floor).  -m_k*p[6], -m_k*p[8] are the
//   only other terms.  The result is (T - M*p)/R in (T/R - p, T/R]: congruent to a*b/R, lazy like every other value
//   here (the additive form gives the same value + p).
//   F_n (dense): additive, m_k = c_k * (-n^-1) mod 2^29, then + m_k * n.
// opaque(): keeps a power-of-two constant out of the optimizer's sight so that m * 2^19 stays ONE v_mad instead of a
// 64-bit shift + a 64-bit add.
BP_HD void chain(int64_t &acc) {   // keeps the MAD chain in source order (the reassociation pass would split it)
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(acc));
#endif
}
BP_HD void fence() {   // one multiplication = one scheduling region: interleaving independent MAD chains only costs registers
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}
BP_HD int32_t opaque(int32_t k) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(k));
#endif
  return k;
}
// A whole column as ONE asm statement of N chained v_mad_i64_i32 (device, F_p).  Written as separate C++ MADs with an
// empty asm between them (chain(), below) the order is kept too, but every VALU instruction that reads a register
// "defined" by an inline asm in the slot before it gets an s_nop from the hazard recognizer: ~100 per multiplication.
#if defined(__HIP_DEVICE_COMPILE__)
BP_HD void madchain1(int64_t &acc, int32_t x0, int32_t y0) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x0), "v"(y0) : "vcc");
}
BP_HD void madchain2(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
BP_HD void madchain3(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2) : "vcc");
}
BP_HD void madchain4(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3) : "vcc");
}
BP_HD void madchain5(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4) : "vcc");
}
BP_HD void madchain6(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4, int32_t x5, int32_t y5) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5) : "vcc");
}
BP_HD void madchain7(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4, int32_t x5, int32_t y5, int32_t x6, int32_t y6) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6) : "vcc");
}
BP_HD void madchain8(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4, int32_t x5, int32_t y5, int32_t x6, int32_t y6, int32_t x7, int32_t y7) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7) : "vcc");
}
BP_HD void madchain9(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4, int32_t x5, int32_t y5, int32_t x6, int32_t y6, int32_t x7, int32_t y7, int32_t x8, int32_t y8) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0\n\tv_mad_i64_i32 %0, vcc, %17, %18, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7), "v"(x8), "v"(y8) : "vcc");
}
BP_HD void madchain10(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4, int32_t x5, int32_t y5, int32_t x6, int32_t y6, int32_t x7, int32_t y7, int32_t x8, int32_t y8, int32_t x9, int32_t y9) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0\n\tv_mad_i64_i32 %0, vcc, %17, %18, %0\n\tv_mad_i64_i32 %0, vcc, %19, %20, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7), "v"(x8), "v"(y8), "v"(x9), "v"(y9) : "vcc");
}
BP_HD void madchain11(int64_t &acc, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2, int32_t x3, int32_t y3, int32_t x4, int32_t y4, int32_t x5, int32_t y5, int32_t x6, int32_t y6, int32_t x7, int32_t y7, int32_t x8, int32_t y8, int32_t x9, int32_t y9, int32_t x10, int32_t y10) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %3, %4, %0\n\tv_mad_i64_i32 %0, vcc, %5, %6, %0\n\tv_mad_i64_i32 %0, vcc, %7, %8, %0\n\tv_mad_i64_i32 %0, vcc, %9, %10, %0\n\tv_mad_i64_i32 %0, vcc, %11, %12, %0\n\tv_mad_i64_i32 %0, vcc, %13, %14, %0\n\tv_mad_i64_i32 %0, vcc, %15, %16, %0\n\tv_mad_i64_i32 %0, vcc, %17, %18, %0\n\tv_mad_i64_i32 %0, vcc, %19, %20, %0\n\tv_mad_i64_i32 %0, vcc, %21, %22, %0" : "+v"(acc) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3), "v"(x4), "v"(y4), "v"(x5), "v"(y5), "v"(x6), "v"(y6), "v"(x7), "v"(y7), "v"(x8), "v"(y8), "v"(x9), "v"(y9), "v"(x10), "v"(y10) : "vcc");
}
BP_HD void madchain(int64_t &acc, int n, const int32_t *x, const int32_t *y) {
  switch (n) {
    case 1: madchain1(acc, x[0], y[0]); break;
    case 2: madchain2(acc, x[0], y[0], x[1], y[1]); break;
    case 3: madchain3(acc, x[0], y[0], x[1], y[1], x[2], y[2]); break;
    case 4: madchain4(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3]); break;
    case 5: madchain5(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4]); break;
    case 6: madchain6(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5]); break;
    case 7: madchain7(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5], x[6], y[6]); break;
    case 8: madchain8(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5], x[6], y[6], x[7], y[7]); break;
    case 9: madchain9(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5], x[6], y[6], x[7], y[7], x[8], y[8]); break;
    case 10: madchain10(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5], x[6], y[6], x[7], y[7], x[8], y[8], x[9], y[9]); break;
    case 11: madchain11(acc, x[0], y[0], x[1], y[1], x[2], y[2], x[3], y[3], x[4], y[4], x[5], y[5], x[6], y[6], x[7], y[7], x[8], y[8], x[9], y[9], x[10], y[10]); break;
    default: break;
  }
}
#endif
template <class F> BP_HD void mont_column(int64_t &acc, int32_t *m, int k) {
  if constexpr (F::sparse) {
    if (k >= 6 && k - 6 < NL) { acc += (int64_t)m[k - 6] * (int64_t)(-4456448); chain(acc); }   // -p[6] = -(17 << 18)
    if (k >= 8 && k - 8 < NL) { acc += (int64_t)m[k - 8] * (int64_t)opaque(-524288); chain(acc); }   // -p[8] = -(1 << 19)
    if (k < NL) m[k] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
  } else {
    constexpr int32_t MOD[NL] = FN_MOD;
#pragma unroll
    for (int i = (k >= NL ? k - NL + 1 : 0); i < k && i < NL; i++) { acc += (int64_t)m[i] * (int64_t)(k - i == NL - 1 ? opaque(MOD[NL - 1]) : MOD[k - i]); chain(acc); }
    if (k < NL) {
      m[k] = (int32_t)(((uint32_t)acc * FN_N0) & (uint32_t)LMASK);
      acc += (int64_t)m[k] * (int64_t)MOD[0]; chain(acc);
    }
  }
}
// a * b / R  (81 MADs + reduction)
template <class F> BP_HD Fe<F> mul_cols(const Fe<F> &a, const Fe<F> &b) {
  int32_t m[NL];
  Fe<F> r;
  int64_t acc = 0;
  fence();
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (F::sparse) {
      int32_t x[11], y[11];
      int n = 0;
#pragma unroll
      for (int j = (k >= NL ? k - NL + 1 : 0); j <= k && j < NL; j++) { x[n] = a.v[j]; y[n] = b.v[k - j]; n++; }
      if (k >= 6 && k - 6 < NL) { x[n] = m[k - 6]; y[n] = -4456448; n++; }   // -p[6] = -(17 << 18)
      if (k >= 8 && k - 8 < NL) { x[n] = m[k - 8]; y[n] = -524288; n++; }    // -p[8] = -(1 << 19)
      madchain(acc, n, x, y);
      if (k < NL) m[k] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
      else r.v[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
      acc >>= LB;
      continue;
    }
    if constexpr (!F::sparse) {
      constexpr int32_t MOD[NL] = FN_MOD;
      int32_t x[11], y[11];
      int n = 0;
#pragma unroll
      for (int j = (k >= NL ? k - NL + 1 : 0); j <= k && j < NL; j++) { x[n] = a.v[j]; y[n] = b.v[k - j]; n++; }
      madchain(acc, n, x, y);
      n = 0;
#pragma unroll
      for (int i = (k >= NL ? k - NL + 1 : 0); i < k && i < NL; i++) if (MOD[k - i] != 0) { x[n] = m[i]; y[n] = MOD[k - i]; n++; }
      madchain(acc, n, x, y);
      if (k < NL) {
        m[k] = (int32_t)(((uint32_t)acc * FN_N0) & (uint32_t)LMASK);
        madchain1(acc, m[k], MOD[0]);
      } else r.v[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
      acc >>= LB;
      continue;
    }
#endif
#pragma unroll
    for (int j = (k >= NL ? k - NL + 1 : 0); j <= k && j < NL; j++) { acc += (int64_t)a.v[j] * (int64_t)b.v[k - j]; chain(acc); }
    mont_column<F>(acc, m, k);
    if (k >= NL) r.v[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
    acc >>= LB;
  }
  r.v[NL - 1] = (int32_t)acc;
  fence();
  return r;
}
// a * a / R  (45 MADs + reduction)
template <class F> BP_HD Fe<F> sqr_cols(const Fe<F> &a) {
  int32_t m[NL], a2[NL];
  Fe<F> r;
  int64_t acc = 0;
  fence();
#pragma unroll
  for (int j = 0; j < NL; j++) a2[j] = 2 * a.v[j];
#pragma unroll
  for (int k = 0; k < 2 * NL - 1; k++) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (F::sparse) {
      int32_t x[11], y[11];
      int n = 0;
#pragma unroll
      for (int j = (k >= NL ? k - NL + 1 : 0); 2 * j < k; j++) { x[n] = a.v[j]; y[n] = a2[k - j]; n++; }
      if ((k & 1) == 0) { x[n] = a.v[k / 2]; y[n] = a.v[k / 2]; n++; }
      if (k >= 6 && k - 6 < NL) { x[n] = m[k - 6]; y[n] = -4456448; n++; }
      if (k >= 8 && k - 8 < NL) { x[n] = m[k - 8]; y[n] = -524288; n++; }
      madchain(acc, n, x, y);
      if (k < NL) m[k] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
      else r.v[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
      acc >>= LB;
      continue;
    }
    if constexpr (!F::sparse) {
      constexpr int32_t MOD[NL] = FN_MOD;
      int32_t x[11], y[11];
      int n = 0;
#pragma unroll
      for (int j = (k >= NL ? k - NL + 1 : 0); 2 * j < k; j++) { x[n] = a.v[j]; y[n] = a2[k - j]; n++; }
      if ((k & 1) == 0) { x[n] = a.v[k / 2]; y[n] = a.v[k / 2]; n++; }
      madchain(acc, n, x, y);
      n = 0;
#pragma unroll
      for (int i = (k >= NL ? k - NL + 1 : 0); i < k && i < NL; i++) if (MOD[k - i] != 0) { x[n] = m[i]; y[n] = MOD[k - i]; n++; }
      madchain(acc, n, x, y);
      if (k < NL) {
        m[k] = (int32_t)(((uint32_t)acc * FN_N0) & (uint32_t)LMASK);
        madchain1(acc, m[k], MOD[0]);
      } else r.v[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
      acc >>= LB;
      continue;
    }
#endif
#pragma unroll
    for (int j = (k >= NL ? k - NL + 1 : 0); 2 * j < k; j++) { acc += (int64_t)a.v[j] * (int64_t)a2[k - j]; chain(acc); }
    if ((k & 1) == 0) { acc += (int64_t)a.v[k / 2] * (int64_t)a.v[k / 2]; chain(acc); }
    mont_column<F>(acc, m, k);
    if (k >= NL) r.v[k - NL] = (int32_t)((uint32_t)acc & (uint32_t)LMASK);
    acc >>= LB;
  }
  r.v[NL - 1] = (int32_t)acc;
  fence();
  return r;
}

// Device code takes the column form for both fields (asm MAD chains).  On the host (CPU tests of these headers) F_n
// keeps the row form, which doubles as an independent statement of the same product for the tests.
template <class F> BP_HD Fe<F> mul(const Fe<F> &a, const Fe<F> &b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return mul_cols(a, b);
#else
  if constexpr (F::sparse) return mul_cols(a, b);
  else return mul_rows(a, b);
#endif
}
template <class F> BP_HD Fe<F> sqr(const Fe<F> &a) {
#if defined(__HIP_DEVICE_COMPILE__)
  return sqr_cols(a);
#else
  if constexpr (F::sparse) return sqr_cols(a);
  else return sqr_rows(a);
#endif
}

// unique representative in [0, m): all limbs in [0, 2^29).  Accepts |value| < 16m.
template <class F> BP_HD Fe<F> canon(const Fe<F> &x) {
  int32_t MOD[NL];
  if constexpr (F::sparse) { constexpr int32_t C[NL] = FP_MOD; for (int j = 0; j < NL; j++) MOD[j] = C[j]; }
  else { constexpr int32_t C[NL] = FN_MOD; for (int j = 0; j < NL; j++) MOD[j] = C[j]; }
  Fe<F> r;
  int64_t carry = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {   // x + 16m >= 0, rippling carry
    int64_t t = (int64_t)x.v[j] + 16 * (int64_t)MOD[j] + carry;
    if (j < NL - 1) { r.v[j] = (int32_t)(t & LMASK); carry = t >> LB; }
    else r.v[j] = (int32_t)t;
  }
#pragma unroll
  for (int k = 16; k >= 1; k >>= 1) {   // value in [0, 32m): subtract 16m, 8m, 4m, 2m, m when possible
    Fe<F> t;
    int64_t br = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      int64_t d = (int64_t)r.v[j] - (int64_t)k * MOD[j] + br;
      if (j < NL - 1) { t.v[j] = (int32_t)(d & LMASK); br = d >> LB; }
      else t.v[j] = (int32_t)d;
    }
    bool ge = t.v[NL - 1] >= 0;
#pragma unroll
    for (int j = 0; j < NL; j++) r.v[j] = ge ? t.v[j] : r.v[j];
  }
  return r;
}
template <class F> BP_HD bool is_zero_exact(const Fe<F> &x) {
  Fe<F> c = canon(x);
  int32_t o = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) o |= c.v[j];
  return o == 0;
}
// all limbs literally zero (the encoding of "Z = 0" for the point at infinity; producers guarantee it)
template <class F> BP_HD bool is_zero_limbs(const Fe<F> &x) {
  int32_t o = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) o |= x.v[j];
  return o == 0;
}
// Cheap necessary condition for x == 0 (mod p) when |value| <= 16p: k*p = k (mod 2^29).  F_p only.
BP_HD bool fp_maybe_zero(const Fp &x) { return (uint32_t)((x.v[0] + 16) & LMASK) <= 32u; }

// 8 x u32 little-endian words (a 256-bit integer < 2^253) <-> limbs
template <class F> BP_HD Fe<F> unpack(const uint32_t w[8]) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const int bit = LB * j, k = bit >> 5, s = bit & 31;
    uint64_t two = (uint64_t)w[k] | (k + 1 < 8 ? (uint64_t)w[k + 1] << 32 : 0);
    r.v[j] = (int32_t)((two >> s) & (j < NL - 1 ? (uint32_t)LMASK : 0xFFFFFFFFu));
  }
  return r;
}
// requires canonical limbs (canon() output)
template <class F> BP_HD void pack(uint32_t w[8], const Fe<F> &x) {
  uint64_t acc = 0;
  int fill = 0, k = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    acc |= (uint64_t)(uint32_t)x.v[j] << fill;
    fill += LB;
    if (fill >= 32) { w[k++] = (uint32_t)acc; acc >>= 32; fill -= 32; }
  }
  if (k < 8) w[k] = (uint32_t)acc;
}
template <class F> BP_HD Fe<F> to_mont(const Fe<F> &plain) { return mul(plain, fe_r2<F>()); }
template <class F> BP_HD Fe<F> from_mont(const Fe<F> &m) {   // -> canonical plain integer limbs
  Fe<F> one = fe_zero<F>();
  one.v[0] = 1;
  return canon(mul(m, one));
}
// is the 256-bit integer < modulus ?
template <class F> BP_HD bool words_lt_mod(const uint32_t w[8]) {
  uint32_t M[8];
  if constexpr (F::sparse) { constexpr uint32_t C[8] = FP_MOD_W; for (int j = 0; j < 8; j++) M[j] = C[j]; }
  else { constexpr uint32_t C[8] = FN_MOD_W; for (int j = 0; j < 8; j++) M[j] = C[j]; }
  bool lt = false, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; j--) {
    bool l = w[j] < M[j], g = w[j] > M[j];
    lt = decided ? lt : l;
    decided = decided || l || g;
  }
  return lt;
}

// x^e for a 256-bit exponent (little-endian words), left-to-right binary
template <class F> BP_HD Fe<F> fe_pow(const Fe<F> &x, const uint32_t e[8]) {
  Fe<F> acc = fe_one<F>();
  for (int i = 255; i >= 0; i--) {
    acc = sqr(acc);
    if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, x);
  }
  return acc;
}
// Field inversion by Fermat (0 -> 0).  F_p: p - 2 = (2^59 + 16) * 2^192 + (2^192 - 1).  The cross-check of inv() below.
template <class F> BP_HD Fe<F> inv_fermat(const Fe<F> &x) {
  if constexpr (F::sparse) {
    Fe<F> x2 = sqr(x), x3 = mul(x2, x), x6 = sqr(x3), x7 = mul(x6, x), x14 = sqr(x7), x15 = mul(x14, x);
    Fe<F> t = x;
    for (int i = 0; i < 55; i++) t = sqr(t);
    t = mul(t, x);                          // x^(2^55 + 1)
    for (int i = 0; i < 4; i++) t = sqr(t); // x^(2^59 + 16)
    for (int i = 0; i < 48; i++) {          // 192 one-bits, 4 at a time
      t = sqr(t); t = sqr(t); t = sqr(t); t = sqr(t);
      t = mul(t, x15);
    }
    return t;
  } else {
    constexpr uint32_t E[8] = FN_EXP_INV_W;
    uint32_t e[8];
    for (int j = 0; j < 8; j++) e[j] = E[j];
    return fe_pow(x, e);
  }
}


// ---- binary extended GCD inversion on plain 256-bit words (HAC 14.61 for an odd modulus) ----------
// ~250 subtract steps + ~500 halvings of 8-word integers (~20 k instructions) instead of the ~380
// Montgomery multiplications (~125 k instructions) of Fermat: the one serial dependency chain per proof
// in the verifier's scalar assembly (y^-1 and u_j^-1, r1cs/verifier.rs:468, inner_product_proof.rs:283).
// Variable time (verification handles public data only).  a in [1, m); returns a^-1 mod m; 0 -> 0.
BP_HD bool w8_is_even(const uint32_t a[8]) { return (a[0] & 1u) == 0; }
BP_HD bool w8_is_one(const uint32_t a[8]) {
  uint32_t o = a[0] ^ 1u;
#pragma unroll
  for (int j = 1; j < 8; j++) o |= a[j];
  return o == 0;
}
BP_HD bool w8_is_zero(const uint32_t a[8]) {
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) o |= a[j];
  return o == 0;
}
BP_HD bool w8_geq(const uint32_t a[8], const uint32_t b[8]) {
  bool ge = true, decided = false;
#pragma unroll
  for (int j = 7; j >= 0; j--) {
    bool l = a[j] < b[j], g = a[j] > b[j];
    ge = decided ? ge : !l;
    decided = decided || l || g;
  }
  return ge;
}
BP_HD uint32_t w8_add(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { c += (uint64_t)a[j] + b[j]; r[j] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
BP_HD void w8_sub(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  int64_t br = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { int64_t d = (int64_t)a[j] - (int64_t)b[j] + br; r[j] = (uint32_t)d; br = d >> 32; }
}
BP_HD void w8_shr1(uint32_t a[8], uint32_t top) {
#pragma unroll
  for (int j = 0; j < 7; j++) a[j] = (a[j] >> 1) | (a[j + 1] << 31);
  a[7] = (a[7] >> 1) | (top << 31);
}
template <class F> BP_HD void w8_inv_mod(uint32_t out[8], const uint32_t a[8]) {
  uint32_t M[8];
  if constexpr (F::sparse) { constexpr uint32_t C[8] = FP_MOD_W; for (int j = 0; j < 8; j++) M[j] = C[j]; }
  else { constexpr uint32_t C[8] = FN_MOD_W; for (int j = 0; j < 8; j++) M[j] = C[j]; }
  uint32_t u[8], v[8], x1[8], x2[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { u[j] = a[j]; v[j] = M[j]; x1[j] = j == 0; x2[j] = 0; }
  if (w8_is_zero(u)) { for (int j = 0; j < 8; j++) out[j] = 0; return; }
  for (int guard = 0; guard < 1100 && !w8_is_one(u) && !w8_is_one(v); guard++) {
    if (w8_is_even(u)) {
      w8_shr1(u, 0);
      uint32_t top = 0;
      if (!w8_is_even(x1)) top = w8_add(x1, x1, M);
      w8_shr1(x1, top);
    } else if (w8_is_even(v)) {
      w8_shr1(v, 0);
      uint32_t top = 0;
      if (!w8_is_even(x2)) top = w8_add(x2, x2, M);
      w8_shr1(x2, top);
    } else if (w8_geq(u, v)) {
      w8_sub(u, u, v);
      if (w8_geq(x1, x2)) w8_sub(x1, x1, x2);
      else { uint32_t t[8]; w8_sub(t, x2, x1); w8_sub(x1, M, t); }
    } else {
      w8_sub(v, v, u);
      if (w8_geq(x2, x1)) w8_sub(x2, x2, x1);
      else { uint32_t t[8]; w8_sub(t, x1, x2); w8_sub(x2, M, t); }
    }
  }
  const bool pick_u = w8_is_one(u);
#pragma unroll
  for (int j = 0; j < 8; j++) out[j] = pick_u ? x1[j] : x2[j];
}
// Montgomery-form inverse through the word-level GCD: x = aR -> a = x/R -> a^-1 -> a^-1 R
template <class F> BP_HD Fe<F> inv_gcd(const Fe<F> &x) {
  uint32_t w[8], iw[8];
  pack(w, from_mont(x));
  w8_inv_mod<F>(iw, w);
  return to_mont(unpack<F>(iw));
}

// ---- inversion by divsteps (Bernstein-Yang "safegcd": the half-delta variant with a 2x2 transition matrix per batch of
// steps, as libsecp256k1's modinv32 organises it, restated for 9 limbs of 29 bits: a batch is 29 divsteps on the low limbs
// of f and g, then one pass over the limbs of (f, g) and of (d, e)).  21 batches = 609 divsteps >= the 590 that suffice for
// any odd modulus below 2^256.  No data-dependent branch anywhere (every lane of a wave inverts its own value in lock
// step), ~17 k instructions against ~34 k (F_p) / ~90 k (F_n) for Fermat's x^(m-2).  0 -> 0.
struct DsMat { int32_t u, v, q, r; };   // t (f, g) = 2^29 (f', g')
BP_HD int32_t ds_divsteps(int32_t zeta, uint32_t f0, uint32_t g0, DsMat &t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;   // matrix entries: signed values held mod 2^32 (left shifts stay defined)
#pragma unroll
  for (int i = 0; i < LB; i++) {
    uint32_t c1 = (uint32_t)(zeta >> 31);             // zeta < 0
    const uint32_t c2 = 0u - (g & 1u);                // g odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // +-(f, u, v)
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;        // -zeta - 2 or zeta - 1
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return zeta;
}
// x^-1 in the representation x is in (Montgomery form in, Montgomery form out: the run starts from e = R^2 instead of 1).
// Output: lazy signed limbs, |value| < 2m.
template <class F> BP_HD Fe<F> inv_ds(const Fe<F> &x) {
  int32_t MOD[NL];
  if constexpr (F::sparse) { constexpr int32_t C[NL] = FP_MOD; for (int j = 0; j < NL; j++) MOD[j] = C[j]; }
  else { constexpr int32_t C[NL] = FN_MOD; for (int j = 0; j < NL; j++) MOD[j] = C[j]; }
  Fe<F> f, g = canon(x), d = fe_zero<F>(), e = fe_r2<F>();
#pragma unroll
  for (int j = 0; j < NL; j++) f.v[j] = MOD[j];
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 21; it++) {
    DsMat t;
    zeta = ds_divsteps(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    {   // (d, e) <- t (d, e) / 2^29 mod m, both kept in (-2m, m)
      const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
      int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
      int64_t cd = (int64_t)t.u * d.v[0] + (int64_t)t.v * e.v[0], ce = (int64_t)t.q * d.v[0] + (int64_t)t.r * e.v[0];
      // multiples of m that clear the low 29 bits: m^-1 mod 2^29 is 1 for p, -N0 for n
      if constexpr (F::sparse) {
        md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & (uint32_t)LMASK);
        me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & (uint32_t)LMASK);
      } else {
        md -= (int32_t)(((uint32_t)md - FN_N0 * (uint32_t)cd) & (uint32_t)LMASK);
        me -= (int32_t)(((uint32_t)me - FN_N0 * (uint32_t)ce) & (uint32_t)LMASK);
      }
      cd += (int64_t)MOD[0] * md; ce += (int64_t)MOD[0] * me;
      cd >>= LB; ce >>= LB;
#pragma unroll
      for (int j = 1; j < NL; j++) {
        cd += (int64_t)t.u * d.v[j] + (int64_t)t.v * e.v[j];
        ce += (int64_t)t.q * d.v[j] + (int64_t)t.r * e.v[j];
        if (MOD[j] != 0) { cd += (int64_t)MOD[j] * md; ce += (int64_t)MOD[j] * me; }
        d.v[j - 1] = (int32_t)(cd & LMASK); cd >>= LB;
        e.v[j - 1] = (int32_t)(ce & LMASK); ce >>= LB;
      }
      d.v[NL - 1] = (int32_t)cd; e.v[NL - 1] = (int32_t)ce;
    }
    {   // (f, g) <- t (f, g) / 2^29, exact
      int64_t cf = (int64_t)t.u * f.v[0] + (int64_t)t.v * g.v[0], cg = (int64_t)t.q * f.v[0] + (int64_t)t.r * g.v[0];
      cf >>= LB; cg >>= LB;
#pragma unroll
      for (int j = 1; j < NL; j++) {
        cf += (int64_t)t.u * f.v[j] + (int64_t)t.v * g.v[j];
        cg += (int64_t)t.q * f.v[j] + (int64_t)t.r * g.v[j];
        f.v[j - 1] = (int32_t)(cf & LMASK); cf >>= LB;
        g.v[j - 1] = (int32_t)(cg & LMASK); cg >>= LB;
      }
      f.v[NL - 1] = (int32_t)cf; g.v[NL - 1] = (int32_t)cg;
    }
  }
  // g = 0, f = +-gcd = +-1 (or +-m for x = 0, where d = 0): the inverse is d * sign(f)
  const int32_t sf = f.v[NL - 1] >> 31;
#pragma unroll
  for (int j = 0; j < NL; j++) d.v[j] = (d.v[j] ^ sf) - sf;
  return d;
}
// THE field inversion of every kernel (0 -> 0)
template <class F> BP_HD Fe<F> inv(const Fe<F> &x) { return inv_ds(x); }

}  // namespace bp
