// vs_prep.cuh -- the inversion pass of the verifier's scalar assembly, shared by k_scalar.hip (stand-alone launch) and
// k_ec.hip (fused with the proof-point tables of a verification batch: neither depends on the other).
// One LANE per proof: y^-1 and u_j^-1 (r1cs/verifier.rs:468, inner_product_proof.rs:283) by Montgomery's trick around
// one Fermat inversion, then u_j^2, u_j^-2, prod u_j^-1.  (Run by one lane of a per-proof block this serial chain cost a
// full wave's issue slots per proof: a quarter of all instructions of a verification.)
// aux per proof (NL ints each): 0 y_inv, 1 allinv, 2.. u_sq[32], 34.. u_inv_sq[32] (+ partials, large path)
#pragma once
#include "fn_dev.cuh"

namespace bpk {

constexpr int VS_AUX = 66;
struct VsPrepArgs { VerifyDims d; const Words8 *challenges; int32_t *aux_all; size_t aux_stride; };

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
}
__device__ __forceinline__ void vs_prep_body(const VsPrepArgs &a, size_t blk) { vs_prep_lane(a, blk * 64 + threadIdx.x); }   // 64-thread blocks

}  // namespace bpk
