// fn_dev.cuh -- device helpers for F_n kernels shared by k_scalar.hip and the fused verification launches of k_ec.hip.
#pragma once
#include "fe29.cuh"
#include "kernels.h"

namespace bpk {
using namespace bp;

__device__ __forceinline__ Fn load_plain(const Words8 *p) {   // plain canonical words -> Montgomery
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = p->w[j];
  return to_mont(unpack<FN>(w));
}
__device__ __forceinline__ void store_plain(Words8 *p, const Fn &x) {   // Montgomery -> plain canonical words
  uint32_t w[8];
  pack(w, from_mont(x));
#pragma unroll
  for (int j = 0; j < 8; j++) p->w[j] = w[j];
}
__device__ __forceinline__ Fn fn_from_u32(uint32_t v) {
  Fn t = fe_zero<FN>();
  t.v[0] = (int32_t)(v & LMASK);
  t.v[1] = (int32_t)(v >> LB);
  return to_mont(t);
}
__device__ __forceinline__ Fn fn_pow_u32(Fn base, uint32_t e) {   // base^e, e >= 0
  Fn acc = fe_one<FN>();
  while (e) {
    if (e & 1) acc = mul(acc, base);
    e >>= 1;
    if (e) base = sqr(base);
  }
  return acc;
}
// Lazy sums keep limbs small but let the VALUE grow (top limb has ~11 spare bits over a 252-bit
// modulus): fold the value back into (-eps, (1+eps) n) with one Montgomery multiplication by R mod n.
// Rule used below: never add more than ~64 reduced values (x 64 lanes of a wave sum) without it.
__device__ __forceinline__ Fn fn_reduce(const Fn &x) { return mul(x, fe_one<FN>()); }
// raw limb I/O for device scratch (zpow tables, partial sums)
__device__ __forceinline__ void raw_put(int32_t *d, const Fn &x) {
#pragma unroll
  for (int j = 0; j < NL; j++) d[j] = x.v[j];
}
__device__ __forceinline__ Fn raw_get(const int32_t *s) {
  Fn x;
#pragma unroll
  for (int j = 0; j < NL; j++) x.v[j] = s[j];
  return x;
}
// wave-level sum of one Fn per lane (shuffle tree); result in every lane
__device__ __forceinline__ Fn wave_sum(Fn x) {
#pragma unroll 1
  for (int off = 32; off > 0; off >>= 1) {
    Fn o;
#pragma unroll
    for (int j = 0; j < NL; j++) o.v[j] = __shfl_xor(x.v[j], off, 64);
    x = add(x, o);
  }
  return x;
}

}  // namespace bpk
