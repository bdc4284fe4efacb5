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
// The same MSMs with a PROOF per lane and a run of whole generators per wave (`gens_per_chunk` of the 2 + 2n): the 64 lanes of a wave
// walk the same (generator, window) pairs, so a wave-load gathers 64 rows of ONE 2^(C-1)-row window of the table instead of rows of
// sixteen, the scalar of a generator is loaded and recoded once for its W windows, and no lane adds up its neighbours: every lane
// writes its partial sum (part[proof * chunks + chunk]) and the verdict launch adds the `chunks` partials of a proof densely.  The
// chunk length sets the wave count (chunks x ceil(nb / 64)) and the dependency chain (gens_per_chunk x W additions) without a
// butterfly whose cost grows with the lanes per MSM.
template <int C, int AHEAD>
__device__ __forceinline__ void fixed_chunk_body(const AffDev *table, size_t n, size_t cap, const uint32_t *scalars, size_t sc_stride,
                                                 JacRaw *part, size_t nb, unsigned chunks, unsigned gens_per_chunk, size_t blk,
                                                 unsigned kinds = 1, unsigned kind = 0) {
  // (kinds > 1: the MSMs come in `kinds` interleaved classes -- MSM i is of class i % kinds -- and a wave holds MSMs of ONE class:
  // classes whose scalars are mostly zero, a prover's bit vectors, then skip the additions a dense class needs wave-uniformly)
  constexpr int W = num_windows<C>();
  constexpr int HALF = 1 << (C - 1);
  const size_t set = blk / chunks;
  const unsigned q = (unsigned)(blk - set * chunks);
  size_t p = (set * 64 + threadIdx.x) * kinds + kind;
  const bool live = p < nb;
  if (!live) p = nb - 1;
  const uint32_t *sc = scalars + p * sc_stride;
  const size_t ngens = 2 + 2 * n, hshift = cap - n;
  const size_t g0 = (size_t)q * gens_per_chunk;
  const size_t g1 = g0 + gens_per_chunk < ngens ? g0 + gens_per_chunk : ngens;
  Xyzz acc = xyzz_inf();
  uint32_t r[9];
  // rows AHEAD pairs ahead of the addition: position `gc, wc` is the next pair to fetch (wave-uniform; only the digit differs between
  // lanes), r the recoded scalar of its generator
  uint32_t row[AHEAD + 1][16];
  int dg[AHEAD + 1];
  size_t gc = g0;
  int wc = 0;
  auto step = [&](uint32_t *dst, int &d) {
    d = 0;
    if (gc < g1) {
      if (wc == 0) {
        uint32_t s[8];
#pragma unroll
        for (int t = 0; t < 8; t++) s[t] = sc[gc * 8 + t];
        recode_add_k<C>(r, s);
      }
      d = recode_digit<C>(r, wc);
      if (d != 0) {
        const size_t rw = (gc < 2 + n ? gc : gc + hshift) * W + wc;
        const AffDev *e = table + rw * HALF + ((d < 0 ? -d : d) - 1);
        uint32_t any = 0;
#pragma unroll
        for (int t = 0; t < 16; t++) { dst[t] = e->w[t]; any |= dst[t]; }
        if (any == 0) d = 0;   // rows of an identity generator are the identity
      }
      if (++wc == W) { wc = 0; gc++; }
    }
  };
#pragma unroll
  for (int a = 0; a < AHEAD; a++) step(row[a], dg[a]);
  const size_t npairs = (g1 > g0 ? g1 - g0 : 0) * W;
#pragma unroll 1
  for (size_t it = 0; it < npairs; it++) {
    step(row[AHEAD], dg[AHEAD]);
    if (dg[0] != 0) {
      Aff a;
      a.x = unpack<FP>(row[0]);
      a.y = unpack<FP>(row[0] + 8);
      if (dg[0] < 0) a.y = neg(a.y);
      acc = xyzz_madd_nzq(acc, a);
    }
#pragma unroll
    for (int a = 0; a < AHEAD; a++) {
#pragma unroll
      for (int t = 0; t < 16; t++) row[a][t] = row[a + 1][t];
      dg[a] = dg[a + 1];
    }
  }
  if (live) raw_store(&part[p * chunks + q], xyzz_to_jac(acc));
}
struct FixedSmallArgs { const AffDev *table; size_t n, cap; const uint32_t *scalars; size_t sc_stride; JacRaw *out; size_t nb;
                        unsigned chunks = 0, gens_per_chunk = 0; /* chunks != 0: fixed_chunk_body, out = nb x chunks partial sums */ };

}  // namespace bpk
