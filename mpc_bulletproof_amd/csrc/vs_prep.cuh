// vs_prep.cuh -- the inversion pass of the verifier's scalar assembly, shared by k_scalar.hip (stand-alone launch) and
// k_ec.hip (fused with the proof-point tables of a verification batch: neither depends on the other).
// One LANE per proof: y^-1 and u_j^-1 (r1cs/verifier.rs:468, inner_product_proof.rs:283) by Montgomery's trick around
// one Fermat inversion, then u_j^2, u_j^-2, prod u_j^-1.  (Run by one lane of a per-proof block this serial chain cost a
// full wave's issue slots per proof: a quarter of all instructions of a verification.)
// aux per proof (NL ints each): 0 y_inv, 1 allinv, 2.. u_sq[32], 34.. u_inv_sq[32] (+ partials, large path)
#pragma once
#include "fn_dev.cuh"

namespace bpk {

// Wave-sized proofs (padded n = 64, k = 6: the 64-bit range gadget of BASELINE configs[1]) -- `fast`: this lane also does every
// piece of the scalar assembly that is SERIAL per proof, so that the wave-per-proof kernel (k_verify_scalars) is left with the
// n-wide work only.  A wave runs a serial step on all 64 lanes: the ~60 one-off field products of a proof (power-table seeds,
// z^64, x^2..x^6, the T / B / B_blinding scalars, the Montgomery conversions of the challenges) cost the wave kernel 60 wave-wide
// products, this lane 60 lane products.  Extra aux slots (NL ints each; "M" = Montgomery form x R, "P" = plain x):
//   66..73 Y1[e] = M(y^-e), 74..81 Y2[j] = M(y^-8j)            -> y^-i  = Y1[i & 7] Y2[i >> 3]   (M)
//   82..89 Z1[e] = P(z^(e+1)), 90..97 Z2[j] = M(z^(8j)), 98 M(z^64)   -> z^(r+1) = Z1[r & 7] Z2[r >> 3] (P), rows >= 64 by z^64 steps
//   99..106 S1[e] = P(allinv prod u_sq[5-bit]^e_bit), 107..114 S2[j] = M(prod u_sq[2-bit]^j_bit)   -> s_i = S1[i & 7] S2[i >> 3]   (P)
//   115 M(x), 116 M(u), 117 M(a), 118 M(b), 119 P(w (t_x - a b) - r t_x), 120 M(r x^2)
// A product of an M and a P value is P: the kernel's outputs come out plain without a conversion multiplication each.
// The lane also writes the scalars that need nothing but challenges straight into the MSM scalar arrays: A_I1 A_O1 S1 A_I2 A_O2 S2,
// T_1 T_3 T_4 T_5 T_6, L_j, R_j, B_blinding (verifier.rs:508-532).
constexpr int VS_AUX = 128;
constexpr int VSF_Y1 = 66, VSF_Y2 = 74, VSF_Z1 = 82, VSF_Z2 = 90, VSF_Z64 = 98, VSF_S1 = 99, VSF_S2 = 107, VSF_X = 115, VSF_U = 116, VSF_A = 117,
              VSF_B = 118, VSF_C0 = 119, VSF_C1 = 120;
struct VsPrepArgs {
  VerifyDims d; const Words8 *challenges; int32_t *aux_all; size_t aux_stride;
  // fast path (all four set, padded n = 64, k = 6): see above
  const Words8 *proof_scalars = nullptr; Words8 *fixed_sc = nullptr, *var_sc = nullptr, *full_sc = nullptr;
};
__host__ __device__ inline bool vs_fast_shape(const VerifyDims &d) { return d.padded_n == 64 && d.k == 6; }
__device__ __forceinline__ void vs_store_p(Words8 *p, const Fn &x) {   // a PLAIN value -> canonical words (no conversion product)
  uint32_t w[8];
  pack(w, canon(x));
#pragma unroll
  for (int j = 0; j < 8; j++) p->w[j] = w[j];
}
__device__ __forceinline__ Fn vs_load_p(const Words8 *p) {             // canonical words -> PLAIN limbs (no conversion product)
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = p->w[j];
  return unpack<FN>(w);
}

// proof p (one lane)
__device__ __forceinline__ void vs_prep_lane(const VsPrepArgs &a, size_t p) {
  // a short, serial link of every batch's dependency chain that shares the chip with the bulk MSM waves of the other
  // in-flight batches: raise its waves' issue priority
  __builtin_amdgcn_s_setprio(3);
  const size_t k = a.d.k;
  if (p >= a.d.nb) return;
  const Words8 *ch = a.challenges + p * (6 + k);
  int32_t *aux = a.aux_all + p * a.aux_stride * NL;
  Fn acc = fe_one<FN>();
  // prefix products live in aux (slots 2.. as scratch) to keep the lane's register footprint small
  raw_put(aux + 2 * NL, acc);
  acc = load_plain(&ch[0]);
  for (size_t i = 0; i < k; i++) { raw_put(aux + (3 + i) * NL, acc); acc = mul(acc, load_plain(&ch[6 + i])); }
  // Fermat here: with a different proof in every lane the binary GCD's data-dependent branches diverge (measured
  // 315 k wave instructions, 1 ms of latency per launch) while the fixed exponent keeps the wave uniform (~125 k)
  Fn ai = inv(acc);
  Fn allinv = fe_one<FN>();
  for (int i = (int)k; i >= 1; i--) {
    Fn val = load_plain(&ch[6 + i - 1]);
    Fn vi = mul(ai, raw_get(aux + (2 + i) * NL));   // (prod_{t<i} val_t)^-1 ... * prefix = val_i^-1
    ai = mul(ai, val);
    allinv = mul(allinv, vi);
    raw_put(aux + (2 + i) * NL, sqr(val));          // slot 2 + i is free again: final home of u_sq[i-1] is 2 + (i-1)
    raw_put(aux + (34 + i - 1) * NL, sqr(vi));
  }
  // shift u_sq down by one slot (slot 2 + i -> 2 + i - 1) and store y_inv, allinv
  for (size_t i = 1; i <= k; i++) raw_put(aux + (2 + i - 1) * NL, raw_get(aux + (2 + i) * NL));
  raw_put(aux, ai);   // after the loop ai = val_0^-1 = y^-1
  raw_put(aux + NL, allinv);
  if (!a.proof_scalars || !a.fixed_sc || !a.var_sc || !vs_fast_shape(a.d) || a.aux_stride < (size_t)VS_AUX) return;
  // ---- fast path extras (see the top of this file)
  const size_t m = a.d.m, np = a.d.padded_n, nvar = 11 + m + 2 * k, nterms = 13 + m + 2 * np + 2 * k;
  const Words8 *ps = a.proof_scalars + p * 5;
  Words8 *vs = a.var_sc + p * nvar, *fx = a.fixed_sc + p * (2 + 2 * np), *full = a.full_sc ? a.full_sc + p * nterms : nullptr;
  const Fn one_m = fe_one<FN>();
  {   // y^-i tables
    Fn t = one_m;
    for (int e = 0; e < 8; e++) { raw_put(aux + (VSF_Y1 + e) * NL, t); t = mul(t, ai); }      // t ends as y^-8
    Fn y8 = t;
    t = one_m;
    for (int j = 0; j < 8; j++) { raw_put(aux + (VSF_Y2 + j) * NL, t); t = mul(t, y8); }
  }
  {   // z^(r+1) tables
    const Fn zp = vs_load_p(&ch[1]), zm = to_mont(zp);
    Fn t = zp;
    for (int e = 0; e < 8; e++) { raw_put(aux + (VSF_Z1 + e) * NL, t); t = mul(t, zm); }      // P(z^(e+1))
    const Fn z8 = sqr(sqr(sqr(zm)));
    t = one_m;
    for (int j = 0; j < 8; j++) { raw_put(aux + (VSF_Z2 + j) * NL, t); t = mul(t, z8); }      // M(z^(8j)); t ends as M(z^64)
    raw_put(aux + VSF_Z64 * NL, t);
  }
  {   // s_i tables (inner_product_proof.rs:298-307, closed form): entry e = entry (e with its lowest set bit cleared) x u_sq[..]
    Fn one_p = fe_zero<FN>();
    one_p.v[0] = 1;
    raw_put(aux + VSF_S1 * NL, mul(allinv, one_p));            // P(allinv)
    raw_put(aux + VSF_S2 * NL, one_m);
    for (int e = 1; e < 8; e++) {
      const int bit = __builtin_ctz(e), prev = e & (e - 1);
      raw_put(aux + (VSF_S1 + e) * NL, mul(raw_get(aux + (VSF_S1 + prev) * NL), raw_get(aux + (2 + (5 - bit)) * NL)));
      raw_put(aux + (VSF_S2 + e) * NL, mul(raw_get(aux + (VSF_S2 + prev) * NL), raw_get(aux + (2 + (2 - bit)) * NL)));
    }
  }
  // the scalars of verifier.rs:508-532 that need only challenges and proof scalars
  const Fn xp = vs_load_p(&ch[3]), xm = to_mont(xp), um = load_plain(&ch[2]), rm = load_plain(&ch[5]), wm = load_plain(&ch[4]);
  const Fn am = load_plain(&ps[3]), bm = load_plain(&ps[4]);
  raw_put(aux + VSF_X * NL, xm); raw_put(aux + VSF_U * NL, um); raw_put(aux + VSF_A * NL, am); raw_put(aux + VSF_B * NL, bm);
  const Fn xx = mul(xm, xp), xxx = mul(xm, xx), x4 = mul(xm, xxx), x5 = mul(xm, x4), x6 = mul(xm, x5);      // plain powers
  auto put = [&](size_t v, size_t fpos, const Fn &val) { vs_store_p(&vs[v], val); if (full) vs_store_p(&full[fpos], val); };
  put(0, 0, xp); put(1, 1, xx); put(2, 2, xxx);                                            // A_I1 A_O1 S1
  put(3, 3, mul(um, xp)); put(4, 4, mul(um, xx)); put(5, 5, mul(um, xxx));                 // A_I2 A_O2 S2
  put(6 + m, 6 + m, mul(rm, xp)); put(7 + m, 7 + m, mul(rm, xxx)); put(8 + m, 8 + m, mul(rm, x4));   // T_1 T_3 T_4
  put(9 + m, 9 + m, mul(rm, x5)); put(10 + m, 10 + m, mul(rm, x6));                        // T_5 T_6
  Fn one_p = fe_zero<FN>();
  one_p.v[0] = 1;
  for (size_t j = 0; j < k; j++) {                                                         // L_j: u_j^2, R_j: u_j^-2
    put(11 + m + j, 13 + m + 2 * np + j, mul(raw_get(aux + (2 + j) * NL), one_p));
    put(11 + m + k + j, 13 + m + 2 * np + k + j, mul(raw_get(aux + (34 + j) * NL), one_p));
  }
  const Fn txp = vs_load_p(&ps[0]), txbp = vs_load_p(&ps[1]), ebp = vs_load_p(&ps[2]);
  // B = w (t_x - a b) + r (x^2 (w_c + delta) - t_x) = c0 + (r x^2) (w_c + delta), c0 = w (t_x - a b) - r t_x; B_blinding = -e_blinding - r t_x_blinding
  const Fn abp = mul(am, mul(bm, one_p));
  raw_put(aux + VSF_C0 * NL, sub(mul(wm, sub(txp, abp)), mul(rm, txp)));
  raw_put(aux + VSF_C1 * NL, mul(rm, to_mont(xx)));                                        // M(r x^2)
  const Fn bb = neg(add(ebp, mul(rm, txbp)));
  vs_store_p(&fx[1], bb);
  if (full) vs_store_p(&full[12 + m], bb);
}
__device__ __forceinline__ void vs_prep_body(const VsPrepArgs &a, size_t blk) { vs_prep_lane(a, blk * 64 + threadIdx.x); }   // 64-thread blocks

}  // namespace bpk
