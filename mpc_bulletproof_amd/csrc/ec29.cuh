// ec29.cuh -- Stark-curve group law (y^2 = x^3 + x + b over F_p) on top of fe29.cuh.
//
// Jacobian (X:Y:Z), identity <=> all limbs of Z are literally zero (every producer below writes
// exact zeros for the identity).  Affine (x, y) in Montgomery form; identity is (0, 0), which is
// not on the curve (b != 0) and matches the all-zero boundary encoding (reference
// src/util.rs:274-289).  All formulas are COMPLETE: P+P, P+(-P), identity operands and the
// B == B_blinding duplicate of reference src/generators.rs:61-70 take the (rare, divergent) slow
// paths guarded by the one-limb filter fp_maybe_zero().
//
// Replaces mpc-stark's StarkPoint add / double / scalar-mul.
#pragma once
#include "fe29.cuh"

namespace bp {

// F_p multiply / square as used by the group law.  With BP_EC_OUTLINE (device builds) they are ONE
// out-of-line copy per code object: a multiplication is ~1.6 KB of code, a mixed addition inlines 11 of
// them and k_straus came to 96 KB -- beyond the instruction cache once several kernels share a CU.
#if defined(__HIP_DEVICE_COMPILE__) && defined(BP_EC_OUTLINE)
static __device__ __noinline__ Fp fpmul(const Fp &a, const Fp &b) { return mul(a, b); }
static __device__ __noinline__ Fp fpsqr(const Fp &a) { return sqr(a); }
#else
BP_HD Fp fpmul(const Fp &a, const Fp &b) { return mul(a, b); }
BP_HD Fp fpsqr(const Fp &a) { return sqr(a); }
#endif

struct Jac { Fp X, Y, Z; };
struct Aff { Fp x, y; };

BP_HD Jac jac_inf() {
  Jac r;
  r.X = fe_one<FP>(); r.Y = fe_one<FP>(); r.Z = fe_zero<FP>();
  return r;
}
BP_HD bool jac_is_inf(const Jac &p) { return is_zero_limbs(p.Z); }
BP_HD bool aff_is_inf(const Aff &p) { return is_zero_limbs(p.x) && is_zero_limbs(p.y); }
BP_HD Jac jac_from_aff(const Aff &a) {
  Jac r;
  if (aff_is_inf(a)) return jac_inf();
  r.X = a.x; r.Y = a.y; r.Z = fe_one<FP>();
  return r;
}
BP_HD Aff aff_neg(const Aff &a) { Aff r; r.x = a.x; r.y = neg(a.y); return r; }
BP_HD Jac jac_neg(const Jac &a) { Jac r = a; r.Y = neg(a.Y); return r; }

// The group law comes in two copies.  `*_full` is complete (P+P, P+(-P), order-2 input) and exact; on the device it
// is ONE out-of-line function per code object, reached only when the one-limb filter fp_maybe_zero() fires
// (probability 33 / 2^29 for an honest operand pair).  The inlined hot path below keeps just the filter: without the
// canonicalisation and the nested doubling of the rare branch it is a third smaller and needs far fewer registers.
#if defined(__HIP_DEVICE_COMPILE__)
#define BP_COLD static __device__ __noinline__
#elif defined(__HIPCC__)
#define BP_COLD static __host__ __device__ inline
#else
#define BP_COLD inline
#endif

// Operands of an out-of-line call travel through the stack.  Copied plainly, the optimizer merges the stack temporary
// with its source -- and the loop-carried accumulator of every caller then LIVES in scratch memory (a scratch load and
// a full s_waitcnt per use in the hot loop).  Copies made through hide() are opaque: only the rare branch touches memory.
BP_HD Fp fp_hide(const Fp &a) {
  Fp r = a;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int j = 0; j < NL; j++) asm volatile("" : "+v"(r.v[j]));
#endif
  return r;
}
BP_HD Jac jac_hide(const Jac &a) { Jac r; r.X = fp_hide(a.X); r.Y = fp_hide(a.Y); r.Z = fp_hide(a.Z); return r; }
BP_HD Aff aff_hide(const Aff &a) { Aff r; r.x = fp_hide(a.x); r.y = fp_hide(a.y); return r; }

// dbl-2007-bl with a = 1, Z3 = 2*Y*Z and S by a product: 6S + 3M
BP_HD Jac jac_dbl_fast(const Jac &p) {
  Fp XX = fpsqr(p.X), YY = fpsqr(p.Y), YYYY = fpsqr(YY), ZZ = fpsqr(p.Z);
  // S = 4 X Y^2 as ONE product: the (X + YY)^2 - XX - YYYY form of dbl-2007-bl trades it for a square, which here is
  // only 27 instructions cheaper than a product (107 / 134) and costs an addition, two subtractions and a carry pass
  Fp S = mul_small<4>(fpmul(p.X, YY));
  Fp M = norm(add_nr(add_nr(add_nr(XX, XX), XX), fpsqr(ZZ)));        // 3 XX + a ZZ^2, a = 1: four tight terms, limbs < 2^31
  Fp T = norm(sub_nr(sub_nr(fpsqr(M), S), S));
  Jac r;
  r.X = T;
  r.Y = sub(fpmul(M, sub_nr(S, T)), mul_small<8>(YYYY));   // S, T tight: the difference has limbs in (-2^29, 2^29), fine for ONE product
  r.Z = mul_small<2>(fpmul(p.Y, p.Z));
  return r;
}
BP_COLD void jac_dbl_full(Jac *out, const Jac *pp) {
  const Jac p = *pp;
  if (jac_is_inf(p) || is_zero_exact(p.Y)) { *out = jac_inf(); return; }   // order-2 points do not exist (odd order), kept for completeness
  *out = jac_dbl_fast(p);
}
BP_HD Jac jac_dbl(const Jac &p) {   // the identity needs no test: Z3 = 2 * Y * 0 has all limbs zero again
  if (fp_maybe_zero(p.Y)) { Jac pc = jac_hide(p), r; jac_dbl_full(&r, &pc); return jac_hide(r); }
  return jac_dbl_fast(p);
}

// mixed addition (Z2 = 1): 8M + 3S
struct MaddMid { Fp H, rr; };
BP_HD MaddMid jac_madd_mid(const Jac &p, const Aff &q) {
  MaddMid m;
  Fp Z1Z1 = fpsqr(p.Z);
  Fp U2 = fpmul(q.x, Z1Z1);
  Fp S2 = fpmul(q.y, fpmul(p.Z, Z1Z1));
  // differences of two tight values (coordinates are always stored normalised) are left un-normalised: limbs in
  // (-2^29 - 8, 2^29 + 8), which products and squares accept on both sides -- three carry passes less per addition
  m.H = sub_nr(U2, p.X);
  m.rr = sub_nr(S2, p.Y);
  return m;
}
BP_HD Jac jac_madd_tail(const Jac &p, const MaddMid &m) {
  Fp HH = fpsqr(m.H), HHH = fpmul(m.H, HH), V = fpmul(p.X, HH);
  Jac r;
  r.X = norm(sub_nr(sub_nr(sub_nr(fpsqr(m.rr), HHH), V), V));
  r.Y = sub(fpmul(m.rr, sub_nr(V, r.X)), fpmul(p.Y, HHH));
  r.Z = fpmul(p.Z, m.H);
  return r;
}
BP_COLD void jac_madd_full(Jac *out, const Jac *pp, const Aff *qq) {
  const Jac p = *pp;
  const Aff q = *qq;
  if (aff_is_inf(q)) { *out = p; return; }
  if (jac_is_inf(p)) { Jac r; r.X = q.x; r.Y = q.y; r.Z = fe_one<FP>(); *out = r; return; }
  MaddMid m = jac_madd_mid(p, q);
  if (is_zero_exact(m.H)) {
    if (is_zero_exact(m.rr)) { Jac t; t.X = q.x; t.Y = q.y; t.Z = fe_one<FP>(); *out = jac_dbl_fast(t); return; }   // y != 0: odd order
    *out = jac_inf();
    return;
  }
  *out = jac_madd_tail(p, m);
}
// Identity operands are resolved by SELECTS after the (then meaningless, but harmless) arithmetic: branch-free, and
// measured equal to an early return (the addition is bound by its 981 v_mad_i64_i32).
BP_HD Jac jac_select(bool c, const Jac &a, const Jac &b) {
  Jac r;
#pragma unroll
  for (int j = 0; j < NL; j++) { r.X.v[j] = c ? a.X.v[j] : b.X.v[j]; r.Y.v[j] = c ? a.Y.v[j] : b.Y.v[j]; r.Z.v[j] = c ? a.Z.v[j] : b.Z.v[j]; }
  return r;
}
BP_HD Jac jac_madd(const Jac &p, const Aff &q) {
  MaddMid m = jac_madd_mid(p, q);
  if (fp_maybe_zero(m.H)) { Jac pc = jac_hide(p), r; Aff qc = aff_hide(q); jac_madd_full(&r, &pc, &qc); return jac_hide(r); }
  Jac r = jac_madd_tail(p, m);
  Jac qj; qj.X = q.x; qj.Y = q.y; qj.Z = fe_one<FP>();
  r = jac_select(jac_is_inf(p), qj, r);
  return jac_select(aff_is_inf(q), p, r);
}

// the same when the caller has already excluded q = identity
BP_HD Jac jac_madd_nzq(const Jac &p, const Aff &q) {
  MaddMid m = jac_madd_mid(p, q);
  if (fp_maybe_zero(m.H)) { Jac pc = jac_hide(p), r; Aff qc = aff_hide(q); jac_madd_full(&r, &pc, &qc); return jac_hide(r); }
  Jac r = jac_madd_tail(p, m);
  Jac qj; qj.X = q.x; qj.Y = q.y; qj.Z = fe_one<FP>();
  return jac_select(jac_is_inf(p), qj, r);
}

// ---- extended Jacobian accumulators (X : Y : ZZ : ZZZ), x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2 ------------------------------
// A mixed addition into such an accumulator is 8M + 2S (madd-2008-s) against 8M + 3S for the Jacobian one: Z^2 and Z^3 are
// carried instead of recomputed.  Used where a lane only ADDS table entries to its accumulator (fixed-base walks, window sums,
// bucket accumulation); the lane converts to Jacobian once (3M + 1S) before its result meets doublings or reductions.
// Identity <=> all limbs of ZZ are literally zero (ZZZ then is, too).
struct Xyzz { Fp X, Y, ZZ, ZZZ; };
BP_HD Xyzz xyzz_inf() {
  Xyzz r;
  r.X = fe_one<FP>(); r.Y = fe_one<FP>(); r.ZZ = fe_zero<FP>(); r.ZZZ = fe_zero<FP>();
  return r;
}
BP_HD bool xyzz_is_inf(const Xyzz &p) { return is_zero_limbs(p.ZZ); }
BP_HD Xyzz xyzz_from_jac(const Jac &p) {
  Xyzz r;
  r.X = p.X; r.Y = p.Y; r.ZZ = fpsqr(p.Z); r.ZZZ = fpmul(r.ZZ, p.Z);
  return r;
}
// (X ZZ^2 : Y ZZ^3 : ZZZ) is the same point in Jacobian coordinates (Z = ZZZ: Z^2 = ZZ^3, Z^3 = ZZZ^3 = ZZ^3 ZZZ)
BP_HD Jac xyzz_to_jac(const Xyzz &p) {
  Jac r;
  Fp z2 = fpsqr(p.ZZ);
  r.X = fpmul(p.X, z2);
  r.Y = fpmul(p.Y, fpmul(z2, p.ZZ));
  r.Z = p.ZZZ;
  return r;
}
BP_HD Xyzz xyzz_hide(const Xyzz &a) { Xyzz r; r.X = fp_hide(a.X); r.Y = fp_hide(a.Y); r.ZZ = fp_hide(a.ZZ); r.ZZZ = fp_hide(a.ZZZ); return r; }
struct XMid { Fp P, R; };
BP_HD XMid xyzz_madd_mid(const Xyzz &p, const Aff &q) {
  XMid m;
  m.P = sub_nr(fpmul(q.x, p.ZZ), p.X);     // a product minus a tight coordinate, left un-normalised (jac_madd_mid)
  m.R = sub_nr(fpmul(q.y, p.ZZZ), p.Y);
  return m;
}
BP_HD Xyzz xyzz_madd_tail(const Xyzz &p, const XMid &m) {
  Fp PP = fpsqr(m.P), PPP = fpmul(m.P, PP), Q = fpmul(p.X, PP);
  Xyzz r;
  r.X = norm(sub_nr(sub_nr(sub_nr(fpsqr(m.R), PPP), Q), Q));
  r.Y = sub(fpmul(m.R, sub_nr(Q, r.X)), fpmul(p.Y, PPP));
  r.ZZ = fpmul(p.ZZ, PP);
  r.ZZZ = fpmul(p.ZZZ, PPP);
  return r;
}
BP_COLD void xyzz_madd_full(Xyzz *out, const Xyzz *pp, const Aff *qq) {
  const Xyzz p = *pp;
  const Aff q = *qq;
  if (aff_is_inf(q)) { *out = p; return; }
  Xyzz qe; qe.X = q.x; qe.Y = q.y; qe.ZZ = fe_one<FP>(); qe.ZZZ = fe_one<FP>();
  if (xyzz_is_inf(p)) { *out = qe; return; }
  XMid m = xyzz_madd_mid(p, q);
  if (is_zero_exact(m.P)) {
    if (is_zero_exact(m.R)) { Jac t; t.X = q.x; t.Y = q.y; t.Z = fe_one<FP>(); *out = xyzz_from_jac(jac_dbl_fast(t)); return; }   // y != 0: odd order
    *out = xyzz_inf();
    return;
  }
  *out = xyzz_madd_tail(p, m);
}
BP_HD Xyzz xyzz_select(bool c, const Xyzz &a, const Xyzz &b) {
  Xyzz r;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    r.X.v[j] = c ? a.X.v[j] : b.X.v[j]; r.Y.v[j] = c ? a.Y.v[j] : b.Y.v[j];
    r.ZZ.v[j] = c ? a.ZZ.v[j] : b.ZZ.v[j]; r.ZZZ.v[j] = c ? a.ZZZ.v[j] : b.ZZZ.v[j];
  }
  return r;
}
// q must not be the identity (callers skip identity table entries); an identity accumulator is resolved by a select
BP_HD Xyzz xyzz_madd_nzq(const Xyzz &p, const Aff &q) {
  XMid m = xyzz_madd_mid(p, q);
  if (fp_maybe_zero(m.P)) { Xyzz pc = xyzz_hide(p), r; Aff qc = aff_hide(q); xyzz_madd_full(&r, &pc, &qc); return xyzz_hide(r); }
  Xyzz r = xyzz_madd_tail(p, m);
  Xyzz qe; qe.X = q.x; qe.Y = q.y; qe.ZZ = fe_one<FP>(); qe.ZZZ = fe_one<FP>();
  return xyzz_select(xyzz_is_inf(p), qe, r);
}
BP_HD Xyzz xyzz_madd(const Xyzz &p, const Aff &q) {
  Xyzz r = xyzz_madd_nzq(p, q);
  return xyzz_select(aff_is_inf(q), p, r);
}

// general addition: 12M + 4S
struct AddMid { Fp U1, S1, H, rr; };
BP_HD AddMid jac_add_mid(const Jac &p, const Jac &q) {
  AddMid m;
  Fp Z1Z1 = fpsqr(p.Z), Z2Z2 = fpsqr(q.Z);
  m.U1 = fpmul(p.X, Z2Z2);
  Fp U2 = fpmul(q.X, Z1Z1);
  m.S1 = fpmul(p.Y, fpmul(q.Z, Z2Z2));
  Fp S2 = fpmul(q.Y, fpmul(p.Z, Z1Z1));
  m.H = sub_nr(U2, m.U1);      // products on both sides: see jac_madd_mid
  m.rr = sub_nr(S2, m.S1);
  return m;
}
BP_HD Jac jac_add_tail(const Jac &p, const Jac &q, const AddMid &m) {
  Fp HH = fpsqr(m.H), HHH = fpmul(m.H, HH), V = fpmul(m.U1, HH);
  Jac r;
  r.X = norm(sub_nr(sub_nr(sub_nr(fpsqr(m.rr), HHH), V), V));
  r.Y = sub(fpmul(m.rr, sub_nr(V, r.X)), fpmul(m.S1, HHH));
  r.Z = fpmul(fpmul(p.Z, q.Z), m.H);
  return r;
}
BP_COLD void jac_add_full(Jac *out, const Jac *pp, const Jac *qq) {
  const Jac p = *pp, q = *qq;
  if (jac_is_inf(p)) { *out = q; return; }
  if (jac_is_inf(q)) { *out = p; return; }
  AddMid m = jac_add_mid(p, q);
  if (is_zero_exact(m.H)) {
    if (is_zero_exact(m.rr)) { jac_dbl_full(out, pp); return; }
    *out = jac_inf();
    return;
  }
  *out = jac_add_tail(p, q, m);
}
BP_HD Jac jac_add(const Jac &p, const Jac &q) {
  AddMid m = jac_add_mid(p, q);
  if (fp_maybe_zero(m.H)) { Jac pc = jac_hide(p), qc = jac_hide(q), r; jac_add_full(&r, &pc, &qc); return jac_hide(r); }
  Jac r = jac_add_tail(p, q, m);
  r = jac_select(jac_is_inf(p), q, r);
  return jac_select(jac_is_inf(q), p, r);
}

// Jacobian -> affine with a known 1/Z
BP_HD Aff jac_to_aff_with_zinv(const Jac &p, const Fp &zinv) {
  Aff a;
  Fp zi2 = fpsqr(zinv);
  a.x = fpmul(p.X, zi2);
  a.y = fpmul(p.Y, fpmul(zi2, zinv));
  return a;
}
BP_HD Aff jac_to_aff(const Jac &p) {
  if (jac_is_inf(p)) { Aff a; a.x = fe_zero<FP>(); a.y = fe_zero<FP>(); return a; }
  return jac_to_aff_with_zinv(p, inv(p.Z));
}
// y^2 == x^3 + x + b ?
BP_HD bool aff_on_curve(const Aff &a) {
  Fp B;
  constexpr int32_t C[NL] = CURVE_B_MONT;
  for (int j = 0; j < NL; j++) B.v[j] = C[j];
  Fp lhs = fpsqr(a.y);
  Fp rhs = norm(add_nr(add_nr(fpmul(fpsqr(a.x), a.x), a.x), B));
  return is_zero_exact(sub(lhs, rhs));
}
BP_HD Aff aff_generator() {
  Aff g;
  constexpr int32_t X[NL] = CURVE_GX_MONT, Y[NL] = CURVE_GY_MONT;
  for (int j = 0; j < NL; j++) { g.x.v[j] = X[j]; g.y.v[j] = Y[j]; }
  return g;
}

// ---- packed HBM formats -------------------------------------------------------------------------
// packed field element = 8 x u32 (256-bit canonical integer).  "Boundary" = plain integer (the C-ABI
// byte encoding); "device" = Montgomery residue (x * 2^261 mod m), also canonical.
struct PackedAff { uint32_t x[8], y[8]; };   // 64 B
struct PackedJac { uint32_t X[8], Y[8], Z[8]; };  // 96 B

BP_HD Aff aff_load_dev(const PackedAff &s) { Aff a; a.x = unpack<FP>(s.x); a.y = unpack<FP>(s.y); return a; }
BP_HD void aff_store_dev(PackedAff &d, const Aff &a) { pack(d.x, canon(a.x)); pack(d.y, canon(a.y)); }
BP_HD Jac jac_load_dev(const PackedJac &s) { Jac a; a.X = unpack<FP>(s.X); a.Y = unpack<FP>(s.Y); a.Z = unpack<FP>(s.Z); return a; }
BP_HD void jac_store_dev(PackedJac &d, const Jac &a) {
  pack(d.X, canon(a.X)); pack(d.Y, canon(a.Y)); pack(d.Z, canon(a.Z));
}
// boundary (plain, x||y little-endian; zeros = identity) -> device Montgomery affine; false if invalid
BP_HD bool aff_from_boundary(Aff &out, const uint32_t xy[16]) {
  uint32_t o = 0;
  for (int j = 0; j < 16; j++) o |= xy[j];
  if (o == 0) { out.x = fe_zero<FP>(); out.y = fe_zero<FP>(); return true; }
  if (!words_lt_mod<FP>(xy) || !words_lt_mod<FP>(xy + 8)) return false;
  out.x = to_mont(unpack<FP>(xy));
  out.y = to_mont(unpack<FP>(xy + 8));
  return aff_on_curve(out);
}
BP_HD void aff_to_boundary(uint32_t xy[16], const Aff &a) {
  if (aff_is_inf(a)) { for (int j = 0; j < 16; j++) xy[j] = 0; return; }
  pack(xy, from_mont(a.x));
  pack(xy + 8, from_mont(a.y));
}

// 256-bit scalar integer helpers (canonical words): signed fixed-window digits via the "+K" trick:
// s' = s + sum_w 2^(c-1) 2^(cw); digit_w = window_w(s') - 2^(c-1) in [-2^(c-1), 2^(c-1)-1].
template <int C> BP_HD constexpr int num_windows() { return 252 / C + 1; }
// sp has 9 words (288 bits) so that the top window of any C <= 20 is representable (W * C <= 273)
template <int C> BP_HD void recode_add_k(uint32_t out[9], const uint32_t s[8]) {
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    uint32_t kw = 0;
#pragma unroll
    for (int w = 0; w < 252 / C + 1; w++) {
      const int bit = C * w + C - 1;
      if ((bit >> 5) == j) kw |= 1u << (bit & 31);
    }
    uint64_t t = (uint64_t)(j < 8 ? s[j] : 0u) + kw + carry;
    out[j] = (uint32_t)t;
    carry = t >> 32;
  }
}
template <int C> BP_HD int recode_digit(const uint32_t sp[9], int w) {
  const int bit = C * w, k = bit >> 5, sft = bit & 31;
  uint64_t two = (uint64_t)sp[k] | (k + 1 < 9 ? (uint64_t)sp[k + 1] << 32 : 0);
  return (int)((two >> sft) & ((1u << C) - 1)) - (1 << (C - 1));
}

}  // namespace bp
