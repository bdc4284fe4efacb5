// fixed_body.cuh -- the small fixed-base MSM body shared by k_fixed.hip (stand-alone launches) and k_ec.hip (fused with
// the proof-point kernels of a verification batch).
#pragma once
#include "ec_dev.cuh"

namespace bpk {
using namespace bp;

// Many small MSMs (the 130-generator part of a range-proof verification): LPM lanes per MSM, 64/LPM MSMs per
// wave, lanes combined with a wave-shuffle butterfly.  Against one 128-lane block per MSM this removes the LDS
// tree (7 levels of full-wave point additions for ~16 table additions per lane).
template <int C, int LPM>
__device__ __forceinline__ void fixed_small_body(const AffDev *table, size_t n, size_t cap, const uint32_t *scalars,
                                                 size_t sc_stride, JacRaw *out, size_t nb, size_t blk) {
  constexpr int W = num_windows<C>();
  constexpr int HALF = 1 << (C - 1);
  const int lane = threadIdx.x & (LPM - 1);
  size_t b = blk * (64 / LPM) + (threadIdx.x / LPM);
  const bool live = b < nb;
  if (!live) b = nb - 1;
  const uint32_t *sc = scalars + b * sc_stride;
  const size_t total = (2 + 2 * n) * W;
  const size_t hshift = cap - n;
  Jac acc = jac_inf();
  // staged software prefetch (k_fixed.hip): the scalar words are requested a pair ahead of the table row whose address they
  // give, so that neither the scalar load nor the row load that depends on it is waited for in front of an addition
  auto load_sc = [&](size_t ll, uint32_t *s) {
    if (ll < total) {
      const size_t g = ll / W;
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] = sc[g * 8 + t];
    }
  };
  auto fetch_row = [&](size_t ll, const uint32_t *s, uint32_t *dst, int &dg) {
    dg = 0;
    if (ll < total) {
      size_t g = ll / W;
      int w = (int)(ll - g * W);
      uint32_t r[9];
      recode_add_k<C>(r, s);
      dg = recode_digit<C>(r, w);
      if (dg != 0) {
        size_t row = (g < 2 + n ? g : g + hshift) * W + w;
        const AffDev *e = table + row * HALF + ((dg < 0 ? -dg : dg) - 1);
#pragma unroll
        for (int t = 0; t < 16; t++) dst[t] = e->w[t];
      }
    }
  };
  // rows TWO pairs ahead (a random 64-byte row of a multi-GB table is a TLB miss + an HBM access: one addition, ~3 us on a SIMD
  // with a single wave, does not always cover it), scalar words three
  uint32_t cur[16], n1[16], sA[8], sB[8];
  int dcur = 0, d1 = 0;
  size_t l = lane;
  load_sc(l, sA);
  load_sc(l + LPM, sB);
  fetch_row(l, sA, cur, dcur);
  load_sc(l + 2 * LPM, sA);
  fetch_row(l + LPM, sB, n1, d1);
#pragma unroll
  for (int t = 0; t < 8; t++) sB[t] = sA[t];
  while (l < total) {
    uint32_t n2[16];
    int d2;
    load_sc(l + 3 * LPM, sA);
    fetch_row(l + 2 * LPM, sB, n2, d2);
    if (dcur != 0) {
      uint32_t any = 0;
#pragma unroll
      for (int t = 0; t < 16; t++) any |= cur[t];
      if (any != 0) {   // rows of an identity generator (the ABI accepts one) are the identity: nothing to add
        Aff q;
        q.x = unpack<FP>(cur);
        q.y = unpack<FP>(cur + 8);
        if (dcur < 0) q.y = neg(q.y);
        acc = jac_madd_nzq(acc, q);
      }
    }
#pragma unroll
    for (int t = 0; t < 16; t++) { cur[t] = n1[t]; n1[t] = n2[t]; }
#pragma unroll
    for (int t = 0; t < 8; t++) sB[t] = sA[t];
    dcur = d1; d1 = d2;
    l += LPM;
  }
#pragma unroll 1
  for (int off = LPM / 2; off > 0; off >>= 1) {
    Jac q;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      q.X.v[t] = __shfl_xor(acc.X.v[t], off, 64);
      q.Y.v[t] = __shfl_xor(acc.Y.v[t], off, 64);
      q.Z.v[t] = __shfl_xor(acc.Z.v[t], off, 64);
    }
    acc = jac_add(acc, q);
  }
  if (lane == 0 && live) raw_store(&out[b], acc);
}
struct FixedSmallArgs { const AffDev *table; size_t n, cap; const uint32_t *scalars; size_t sc_stride; JacRaw *out; size_t nb; };

}  // namespace bpk
