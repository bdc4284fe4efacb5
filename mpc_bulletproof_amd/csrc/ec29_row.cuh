// ec29_row.cuh -- ROW-DISTRIBUTED F_p and point arithmetic: one field element per 16-lane DPP row, limb j in lane j,
// one point per WAVE (the four rows run the four independent multiplications of a level of the group law).
//
// Why: a lone wave issues one instruction per ~4.6 cycles whatever it computes, so the latency of the ~252 dependent
// doublings that end every variable-base MSM (Horner over the windows; inner_product_proof.rs:226-227, verifier.rs:516-547,
// r1cs_mpc/mpc_prover.rs:621-657 all wait for it) is the NUMBER OF INSTRUCTIONS on the chain.  The quad form (ec29_quad.cuh)
// runs whole multiplications side by side -- a doubling is three of them deep, 134 instructions each, 675 in all.  Here ONE
// multiplication is spread over the lanes of a row: lane k accumulates column k of the product (9 chained v_mad_i64_i32 instead
// of 81), the operands arrive by DPP (row_newbcast:i for a_i, row_shr / row_shl for b_{k-i}), and the Montgomery reduction
// is done WITHOUT a carry chain:
//   * p = 1 + 2^192 (17 + 2^59) = 1 (mod 2^174 = 2^(29*6)): the Montgomery digit block M1 of the low SIX columns is those
//     columns themselves (p^-1 = 1 mod 2^174), in whatever lazy form they are -- T - M1 p clears them exactly, column by column,
//     and adds -P6 m_k, -P8 m_k (P6 = 17 << 18, P8 = 1 << 19: limbs 6 and 8 of p) to columns k+6, k+8;
//   * a second block of THREE columns (6..8 of T) finishes R = 2^261 = 2^(29*9) the same way;
//   * between the blocks the 64-bit column sums are brought back to ~30-bit limbs by a parallel split: c = l + 2^29 h + 2^58 g,
//     d_k = l_k + h_{k-1} + g_{k-2} (two DPP shifts and an add3) -- no lane waits for another lane's carry.
// A multiplication is ~75 instructions deep instead of 134, a field addition ONE instead of nine, and a doubling ~270 instead of
// 675.  Per point it spends 16x the lanes of the quad form: this is the form for chains that have the chip to themselves (the
// Horner tails of the MSMs, a lone batch), not for the pipelined verification, which is bound by instruction issue.
//
// Layout: columns 0..5 ("LO") accumulate in lanes 0..5; columns 6..16 ("HI") in lanes 0..10 of a second accumulator, so that
// the reduction's shifts by 6 / 8 limbs and the final realignment stay inside the 16 lanes.  Element invariant: lanes 9..15 of
// the row hold 0.  Products return limbs in (-2^24, 2^29 + 2^24), values in (-2.2 p, 1.1 p); inputs: |limb| up to ~2^30.1 on one
// side against ~2^29.1 on the other (9 * 2^59.2 + the reduction terms < 2^63), |value| < 2^256.
// Model and bounds: tests/row_model.py (lane-level Python restatement, run by the CPU suite); device parity:
// tests/csrc/fe29_gpu_test.hip (k_row*) against the Python big-integer model.  Device only.
#pragma once
#include "ec29.cuh"

#if defined(__HIPCC__)
namespace bp {

typedef int32_t Rfe;   // this lane's limb of a row-distributed element

template <int CTRL> __device__ __forceinline__ int32_t rdpp(int32_t x) {   // DPP move, lanes shifted in from outside the row read 0
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
#else
  return x;   // (the host pass only parses device functions)
#endif
}
constexpr int RSHL = 0x100, RSHR = 0x110, RBC = 0x150;   // row_shl:n (lane l <- l + n), row_shr:n (l <- l - n), row_newbcast:n

__device__ __forceinline__ int32_t rk_hide(int32_t x) {   // keeps a per-lane constant in its VGPR (else: a compare + select per use)
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
  return x;
}
struct RowK {
  int lane, row;
  int32_t mk_lo;   // LMASK in lanes 0..5
  int32_t le5;     // ~0 in lanes 0..5
  int32_t mk_hp;   // LMASK, ~0 in lane 10 (column 16 = a_8 b_8 keeps its sign in the top limb)
  int32_t le2;     // ~0 in lanes 0..2
  int32_t mk_fl;   // LMASK in lanes 3..10, ~0 in lane 11
  int32_t mk_fc;   // ~0 in lanes 3..10
  int32_t mk_n;    // LMASK in lanes 0..7, ~0 in lane 8 (norm)
  int32_t lt8;     // ~0 in lanes 0..7
  int32_t plimb;   // limb `lane` of p
};
__device__ __forceinline__ RowK rowk_init() {
  RowK K;
  K.lane = threadIdx.x & 15;
  K.row = (threadIdx.x >> 4) & 3;
  const int l = K.lane;
  K.mk_lo = rk_hide(l <= 5 ? LMASK : 0);
  K.le5 = rk_hide(l <= 5 ? -1 : 0);
  K.mk_hp = rk_hide(l == 10 ? -1 : LMASK);
  K.le2 = rk_hide(l <= 2 ? -1 : 0);
  K.mk_fl = rk_hide(l >= 3 && l <= 10 ? LMASK : (l == 11 ? -1 : 0));
  K.mk_fc = rk_hide(l >= 3 && l <= 10 ? -1 : 0);
  K.mk_n = rk_hide(l < 8 ? LMASK : (l == 8 ? -1 : 0));
  K.lt8 = rk_hide(l < 8 ? -1 : 0);
  K.plimb = rk_hide(l == 0 ? 1 : (l == 6 ? (17 << 18) : (l == 8 ? (1 << 19) : 0)));
  return K;
}

__device__ __forceinline__ int32_t r_alignbit(uint32_t hi, uint32_t lo, int sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int32_t)__builtin_amdgcn_alignbit(hi, lo, sh);
#else
  return (int32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
}

// a * b / 2^261 (mod p, lazy)
__device__ __forceinline__ Rfe rmul(const RowK &K, Rfe a, Rfe b) {
  const int32_t s0 = rdpp<RBC + 0>(a), s1 = rdpp<RBC + 1>(a), s2 = rdpp<RBC + 2>(a), s3 = rdpp<RBC + 3>(a), s4 = rdpp<RBC + 4>(a),
                s5 = rdpp<RBC + 5>(a), s6 = rdpp<RBC + 6>(a), s7 = rdpp<RBC + 7>(a), s8 = rdpp<RBC + 8>(a);
  const int32_t br1 = rdpp<RSHR + 1>(b), br2 = rdpp<RSHR + 2>(b), br3 = rdpp<RSHR + 3>(b), br4 = rdpp<RSHR + 4>(b), br5 = rdpp<RSHR + 5>(b);
  const int32_t bl1 = rdpp<RSHL + 1>(b), bl2 = rdpp<RSHL + 2>(b), bl3 = rdpp<RSHL + 3>(b), bl4 = rdpp<RSHL + 4>(b), bl5 = rdpp<RSHL + 5>(b),
                bl6 = rdpp<RSHL + 6>(b);
  // LO: lane k <= 5 = column k;  HI: lane l <= 10 = column l + 6
  int64_t LO = (int64_t)s0 * b, HI = (int64_t)s0 * bl6;
  LO += (int64_t)s1 * br1; HI += (int64_t)s1 * bl5;
  LO += (int64_t)s2 * br2; HI += (int64_t)s2 * bl4;
  LO += (int64_t)s3 * br3; HI += (int64_t)s3 * bl3;
  LO += (int64_t)s4 * br4; HI += (int64_t)s4 * bl2;
  LO += (int64_t)s5 * br5; HI += (int64_t)s5 * bl1;
  HI += (int64_t)s6 * b;
  HI += (int64_t)s7 * br1;
  HI += (int64_t)s8 * br2;
  // split the low six columns: d_k = l_k + h_{k-1} + g_{k-2} = the first Montgomery block M1 (p^-1 = 1 mod 2^174)
  const uint32_t ll = (uint32_t)LO, lh = (uint32_t)((uint64_t)LO >> 32);
  const int32_t l = (int32_t)ll & K.mk_lo;
  const int32_t h = r_alignbit(lh, ll, LB) & K.mk_lo;
  const int32_t g = ((int32_t)lh >> (2 * LB - 32)) & K.le5;
  const int32_t m1 = (l + rdpp<RSHR + 1>(h) + rdpp<RSHR + 2>(g)) & K.le5;
  // what the split carries out of column 5 (into columns 6, 7), then T - M1 p on columns 6..13
  const int32_t cin = rdpp<RSHL + 5>(h) + rdpp<RSHL + 4>(g);
  HI += (int64_t)cin * (int64_t)opaque(1);
  HI += (int64_t)m1 * (int64_t)(-4456448);                          // -P6 m1_l        (column l + 6)
  HI += (int64_t)rdpp<RSHR + 2>(m1) * (int64_t)opaque(-524288);     // -P8 m1_{l-2}    (column l + 6 = (l - 2) + 8)
  // split again: ~30-bit limbs d'_l, l = 0..11 (columns 6..17)
  const uint32_t hl = (uint32_t)HI, hh = (uint32_t)((uint64_t)HI >> 32);
  const int32_t lp = (int32_t)hl & LMASK;
  const int32_t hp = r_alignbit(hh, hl, LB) & K.mk_hp;
  const int32_t gp = (int32_t)hh >> (2 * LB - 32);
  const int32_t dp = lp + rdpp<RSHR + 1>(hp) + rdpp<RSHR + 2>(gp);
  // second block: M2 = d'_0..2 (columns 6..8); they cancel, columns 12..14 take -P6 m2, columns 14..16 take -P8 m2
  const int32_t m2 = dp & K.le2;
  const int32_t dm = dp - m2;
  int64_t U = (int64_t)dm;
  U += (int64_t)rdpp<RSHR + 6>(m2) * (int64_t)(-4456448);
  U += (int64_t)rdpp<RSHR + 8>(m2) * (int64_t)opaque(-524288);
  // result limb j = column 9 + j = lane j + 3: low 29 bits + the carry of the lane below; lane 11 (limb 8) keeps everything
  const uint32_t ul = (uint32_t)U, uh = (uint32_t)((uint64_t)U >> 32);
  const int32_t lo = (int32_t)ul & K.mk_fl;
  const int32_t hc = r_alignbit(uh, ul, LB) & K.mk_fc;
  return rdpp<RSHL + 3>(lo) + rdpp<RSHL + 2>(hc);
}

__device__ __forceinline__ Rfe rnorm(const RowK &K, Rfe x) {   // parallel carry: |limb| < 2^31 -> limbs in [-4, 2^29 + 4)
  return (x & K.mk_n) + rdpp<RSHR + 1>((x >> LB) & K.lt8);
}
// x / 2 mod p (limbs |x_j| <= 2^30; output limbs < 2^29 + 2^28 + 2^22): as fp_half_nr of ec29_quad.cuh
__device__ __forceinline__ Rfe rhalf_nr(const RowK &K, Rfe x) {
  const int32_t odd = -(rdpp<RBC + 0>(x) & 1);
  const int32_t s = x + (odd & K.plimb);
  return (s >> 1) + ((rdpp<RSHL + 1>(s) & 1) << (LB - 1));
}
__device__ __forceinline__ Rfe rneg_half_nr(const RowK &K, Rfe w) { return rhalf_nr(K, -w); }

// ---- moving whole elements: rows of a wave, one-lane form <-> row form
struct R4 { Rfe r0, r1, r2, r3; };
// row r of x, in every row (v_permlane16_swap: odd rows of the first operand <-> even rows of the second; v_permlane32_swap:
// upper half of the first <-> lower half of the second -- gfx950)
__device__ __forceinline__ R4 rbc4(Rfe x) {
  R4 o;
#if defined(__HIP_DEVICE_COMPILE__)
  const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);   // [x0 x0 x2 x2], [x1 x1 x3 x3]
  const auto e = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false);                 // [x0 x0 x0 x0], [x2 x2 x2 x2]
  const auto d = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false);                 // [x1 ...], [x3 ...]
  o.r0 = (Rfe)e[0]; o.r2 = (Rfe)e[1]; o.r1 = (Rfe)d[0]; o.r3 = (Rfe)d[1];
#else
  o.r0 = o.r1 = o.r2 = o.r3 = x;
#endif
  return o;
}
// rows 0 and 1 only (lower half of the wave), in every row
__device__ __forceinline__ void rbc01(Rfe x, Rfe &x0, Rfe &x1) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto s = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
  const auto e = __builtin_amdgcn_permlane32_swap(s[0], s[0], false, false);
  const auto d = __builtin_amdgcn_permlane32_swap(s[1], s[1], false, false);
  x0 = (Rfe)e[0]; x1 = (Rfe)d[0];
#else
  x0 = x1 = x;
#endif
}
template <int J> __device__ __forceinline__ void rgather_limb(Fp &o, Rfe x) {
  o.v[J] = rdpp<RBC + J>(x);
  if constexpr (J + 1 < NL) rgather_limb<J + 1>(o, x);
}
__device__ __forceinline__ Fp rgather(Rfe x) { Fp o; rgather_limb<0>(o, x); return o; }   // every lane: the whole element
__device__ __forceinline__ Rfe rscatter(const RowK &K, const Fp &x) {                      // every lane holds x -> its limb
  Rfe r = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) r = K.lane == j ? x.v[j] : r;
  return r;
}
__device__ __forceinline__ Rfe rload(const RowK &K, const int32_t *limbs) { return K.lane < NL ? limbs[K.lane] : 0; }
__device__ __forceinline__ bool r_is_zero_limbs(Rfe z) { return __ballot(z != 0) == 0; }   // (replicated rows: wave-uniform)

// ---- point arithmetic: modified Jacobian (X : Y : Z : Th), Th = -Z^4 / 2, the halved doubling of ec29_quad.cuh.
// Every coordinate is REPLICATED in the four rows on entry and exit; row r computes the r-th product of a level.
struct JacR { Rfe X, Y, Z, Th; };   // X, Y: limbs in (-2^29 - 2^25, 2^29 + 2^25); Z, Th: products (near-tight)

__device__ __forceinline__ JacR jacr_inf(const RowK &K) {
  JacR r;
  constexpr int32_t ONE[NL] = FP_ONE;
  Rfe one = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) one = K.lane == j ? ONE[j] : one;
  r.X = one; r.Y = one; r.Z = 0; r.Th = 0;
  return r;
}
__device__ __forceinline__ bool jacr_is_inf(const JacR &p) { return r_is_zero_limbs(p.Z); }
__device__ __forceinline__ JacR jacr_from_limbs(const RowK &K, const int32_t *v27) {   // JacRaw: X | Y | Z, 9 limbs each
  JacR r;
  r.X = rload(K, v27); r.Y = rload(K, v27 + NL); r.Z = rload(K, v27 + 2 * NL);
  const Rfe zz = rmul(K, r.Z, r.Z);
  r.Th = rmul(K, zz, rneg_half_nr(K, zz));
  return r;
}
// an ADDEND of radd: its Th is never read (only doublings consume Th)
__device__ __forceinline__ JacR jacr_addend_from_limbs(const RowK &K, const int32_t *v27) {
  JacR r;
  r.X = rload(K, v27); r.Y = rload(K, v27 + NL); r.Z = rload(K, v27 + 2 * NL); r.Th = 0;
  return r;
}
// lanes 0..8 of row 0 write the point (X, Y normalised)
__device__ __forceinline__ void jacr_store(const RowK &K, int32_t *v27, const JacR &p) {
  const Rfe x = rnorm(K, p.X), y = rnorm(K, p.Y);
  if (K.row == 0 && K.lane < NL) { v27[K.lane] = x; v27[NL + K.lane] = y; v27[2 * NL + K.lane] = p.Z; }
}
__device__ __forceinline__ Jac jacr_gather(const RowK &K, const JacR &p) {   // one-lane Jacobian form in every lane
  Jac r;
  r.X = rgather(rnorm(K, p.X)); r.Y = rgather(rnorm(K, p.Y)); r.Z = rgather(rnorm(K, p.Z));
  return r;
}
__device__ __forceinline__ JacR jacr_scatter(const RowK &K, const Jac &p) {
  JacR r;
  r.X = rscatter(K, p.X); r.Y = rscatter(K, p.Y); r.Z = rscatter(K, p.Z);
  const Rfe zz = rmul(K, r.Z, r.Z);
  r.Th = rmul(K, zz, rneg_half_nr(K, zz));
  return r;
}

// 2 P:  E = X (3X/2) - Th,  X3 = E^2 - 2 X Y^2,  Y3 = E (X Y^2 - X3) - Y^4,  Z3 = Y Z,  Th3 = Y^4 Th
__device__ __forceinline__ JacR rdbl(const RowK &K, const JacR &p) {
  const bool r0 = K.row == 0, r1 = K.row == 1, r3 = K.row == 3;
  const Rfe X15 = p.X + rhalf_nr(K, p.X);
  // level 1 -- row 0: X (3X/2), rows 1, 2: Y Y, row 3: Y Z
  const Rfe P = rmul(K, r0 ? p.X : p.Y, r0 ? X15 : (r3 ? p.Z : p.Y));
  const R4 Pb = rbc4(P);
  const Rfe E = Pb.r0 - p.Th, YY = Pb.r1;
  // level 2 -- row 0: E E, row 1: X YY, row 2: YY YY, row 3: (2X) YY
  const Rfe Q = rmul(K, r0 ? E : (r1 ? p.X : (r3 ? p.X + p.X : YY)), r0 ? E : YY);
  const R4 Qb = rbc4(Q);
  const Rfe X3 = Qb.r0 - Qb.r3;
  // level 3 -- row 0: E (X YY - X3), the other rows: Y^4 Th
  const Rfe R = rmul(K, r0 ? E : Qb.r2, r0 ? Qb.r1 - X3 : p.Th);
  Rfe R0, R1;
  rbc01(R, R0, R1);
  JacR o;
  o.X = X3; o.Y = R0 - Qb.r2; o.Z = Pb.r3; o.Th = R1;
  return o;
}

// exact cases of the addition (P1 = +-P2), through the complete one-lane law: reached with probability ~2^-25 per addition
static __device__ __noinline__ void radd_exact(JacR *out, const JacR *a, const JacR *b) {
  const RowK K = rowk_init();
  const Jac s = jac_add(jacr_gather(K, *a), jacr_gather(K, *b));
  *out = jacr_scatter(K, (jac_is_inf(s) || is_zero_exact(s.Z)) ? jac_inf() : s);
}
// P1 + P2, complete
__device__ __forceinline__ JacR radd(const RowK &K, const JacR &p1, const JacR &p2) {
  const bool r0 = K.row == 0, r1 = K.row == 1, r2 = K.row == 2, lo = K.row < 2;
  // level 1 -- rows 0, 3: Z1 Z1, row 1: Z2 Z2, row 2: Z1 Z2
  const R4 P = rbc4(rmul(K, r1 ? p2.Z : p1.Z, (r1 || r2) ? p2.Z : p1.Z));
  const Rfe Z1Z1 = P.r0, Z2Z2 = P.r1;
  // level 2 -- row 0: U1 = X1 Z2Z2, row 1: U2 = X2 Z1Z1, row 2: Z2 Z2Z2, row 3: Z1 Z1Z1
  const R4 Q = rbc4(rmul(K, lo ? (r0 ? p1.X : p2.X) : (r2 ? p2.Z : p1.Z), (r0 || r2) ? Z2Z2 : Z1Z1));
  const Rfe U1 = Q.r0, H = Q.r1 - Q.r0;
  // level 3 -- row 0: S1 = Y1 Z2^3, row 1: S2 = Y2 Z1^3, row 2: Z3 = Z1Z2 H, row 3: H H
  const R4 R = rbc4(rmul(K, lo ? (r0 ? p1.Y : p2.Y) : (r2 ? P.r2 : H), lo ? (r0 ? Q.r2 : Q.r3) : H));
  const Rfe S1 = R.r0, Z3 = R.r2, HH = R.r3, rr = R.r1 - R.r0;
  const bool inf1 = jacr_is_inf(p1), inf2 = jacr_is_inf(p2);
  // H = 0 (mod p) needs value(H) = k p, hence limb 0 = k (mod 2^29) for a small k: the one-limb filter of ec29.cuh
  const int32_t h0 = (rdpp<RBC + 0>(H) + 16) & LMASK;
  if (__builtin_expect(h0 < 32 && !inf1 && !inf2, 0)) {    // (replicated values: wave-uniform)
    if (is_zero_exact(rgather(rnorm(K, H)))) {
      JacR o, a = p1, b = p2;
      radd_exact(&o, &a, &b);
      return o;
    }
  }
  // level 4 -- row 0: H HH, row 1: U1 HH, row 2: rr rr, row 3: Z3 Z3
  const Rfe A4 = lo ? (r0 ? H : U1) : (r2 ? rr : Z3);
  const R4 W = rbc4(rmul(K, A4, lo ? HH : A4));
  const Rfe HHH = W.r0, V = W.r1;
  const Rfe X3 = rnorm(K, W.r2 - HHH - V - V);
  // level 5 -- row 0: rr (V - X3), row 1: S1 HHH, rows 2, 3: Th3 = (Z3 Z3) (-(Z3 Z3) / 2)
  const R4 F = rbc4(rmul(K, r0 ? rr : (r1 ? S1 : W.r3), r0 ? V - X3 : (r1 ? HHH : rneg_half_nr(K, W.r3))));
  JacR o;
  o.X = X3; o.Y = F.r0 - F.r1; o.Z = Z3; o.Th = F.r2;
  if (inf1) {   // (wave-uniform)  p2 may be an addend that carries no Th (jacr_addend_from_limbs): the sum needs it
    o = p2;
    const Rfe zz = rmul(K, p2.Z, p2.Z);
    o.Th = rmul(K, zz, rneg_half_nr(K, zz));
  }
  if (inf2) o = p1;
  return o;
}

}  // namespace bp
#endif
