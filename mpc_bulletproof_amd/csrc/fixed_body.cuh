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
  Xyzz acc = xyzz_inf();       // the lane only adds table entries: extended-Jacobian accumulator, 8M + 2S per addition (ec29.cuh)
  uint32_t cur[16];
  int dcur = 0;
  size_t l = lane;
  auto fetch = [&](size_t ll, uint32_t *dst, int &dg) {
    dg = 0;
    if (ll < total) {
      size_t g = ll / W;
      int w = (int)(ll - g * W);
      uint32_t s[8], r[9];
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] = sc[g * 8 + t];
      recode_add_k<C>(r, s);
      dg = recode_digit<C>(r, w);
      if (dg != 0) {
        size_t row = (g < 2 + n ? g : g + hshift) * W + w;
        const AffDev *e = table + row * HALF + ((dg < 0 ? -dg : dg) - 1);
        uint32_t any = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) { dst[t] = e->w[t]; any |= dst[t]; }
        if (any == 0) dg = 0;   // rows of an identity generator (the ABI accepts one) are the identity: nothing to add
      }
    }
  };
  // (Measured and NOT kept here: rows two pairs ahead + scalar words three, as k_fixed.hip's block kernels do.  A lone batch's
  // Horner launch went from 0.49 to 0.41 ms, but the kernel grew from 100 to 208 VGPRs -- two waves per SIMD instead of four --
  // and with twenty batches in flight, where other waves cover a lane's wait anyway, the step rate fell by 7 %.)
  fetch(l, cur, dcur);
  while (l < total) {
    uint32_t nxt[16];
    int dnxt;
    fetch(l + LPM, nxt, dnxt);
    if (dcur != 0) {
      Aff q;
      q.x = unpack<FP>(cur);
      q.y = unpack<FP>(cur + 8);
      if (dcur < 0) q.y = neg(q.y);
      acc = xyzz_madd_nzq(acc, q);   // identity rows were dropped in fetch()
    }
#pragma unroll
    for (int t = 0; t < 16; t++) cur[t] = nxt[t];
    dcur = dnxt;
    l += LPM;
  }
  Jac accj = xyzz_to_jac(acc);
#pragma unroll 1
  for (int off = LPM / 2; off > 0; off >>= 1) {
    Jac q;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      q.X.v[t] = __shfl_xor(accj.X.v[t], off, 64);
      q.Y.v[t] = __shfl_xor(accj.Y.v[t], off, 64);
      q.Z.v[t] = __shfl_xor(accj.Z.v[t], off, 64);
    }
    accj = jac_add(accj, q);
  }
  if (lane == 0 && live) raw_store(&out[b], accj);
}
struct FixedSmallArgs { const AffDev *table; size_t n, cap; const uint32_t *scalars; size_t sc_stride; JacRaw *out; size_t nb; };

}  // namespace bpk
