// k_fixed.hip -- fixed-base (table-lookup) multi-scalar multiplication over the resident generators: no doublings,
// ceil(253 / c) mixed additions per term.  SURVEY.md 8a rows a1 (generator part), a3, a10; split from k_ec.hip so that
// the two translation units compile in parallel.
#include <cstdlib>
#include "fixed_body.cuh"

using namespace bp;

namespace bpk {

// Block (chunk, msm): lane l walks the (generator, window) pairs l, l+TPB, ... of its chunk.  The MSM uses the
// generators [B, Bb, G_0..G_{n-1}, H_0..H_{n-1}] of a table built for capacity cap >= n: used generator g lives
// in table row block g (g < 2 + n) or g + (cap - n) (the H section); scalars are compact (2 + 2n per MSM).
template <int C, int TPB>
__global__ void __launch_bounds__(TPB) k_fixed_msm(const AffDev *table, size_t n, size_t cap, const uint32_t *scalars,
                                                   size_t sc_stride, JacRaw *out, size_t pairs_per_chunk) {
  __shared__ int32_t red[27 * (TPB / 2)];
  constexpr int W = num_windows<C>();
  constexpr int HALF = 1 << (C - 1);
  const int tid = threadIdx.x;
  const uint32_t *sc = scalars + (size_t)blockIdx.y * sc_stride;
  const size_t total = (2 + 2 * n) * W;
  const size_t lo = (size_t)blockIdx.x * pairs_per_chunk;
  const size_t hi = lo + pairs_per_chunk < total ? lo + pairs_per_chunk : total;
  const size_t hshift = cap - n;
  Xyzz acc = xyzz_inf();      // additions only: extended-Jacobian accumulator (8M + 2S per table addition, ec29.cuh)
  // Staged software prefetch: the table row of pair l + 2 TPB is requested before the addition of pair l starts (a random 64-byte
  // row of a multi-GB table is a TLB miss + an HBM access: one addition, ~3 us, does not always cover it), and the scalar words
  // that row's ADDRESS is computed from one iteration earlier still -- the dependent chain scalar load -> recoding -> row load
  // would otherwise sit exposed in front of every addition (a third of the kernel's wave-cycles were spent waiting with two
  // waves per SIMD; the verification's Horner launch went from 0.49 to 0.41 ms alone with the same change in fixed_body.cuh).  The scalar words come from L1/L2 (the W lanes of one generator
  // read the same 32 bytes) and are recoded on the fly: staging all recoded scalars in LDS needs 36 B per generator -- 81 KB at
  // capacity 1024, past the 64 KB dynamic limit.
  auto load_sc = [&](size_t ll, uint32_t *s) {
    if (ll < hi) {
      const size_t g = ll / W;
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] = sc[g * 8 + t];
    }
  };
  auto fetch_row = [&](size_t ll, const uint32_t *s, uint32_t *dst, int &dg) {
    dg = 0;
    if (ll < hi) {
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
  uint32_t cur[16], n1[16], sA[8], sB[8];
  int dcur = 0, d1 = 0;
  size_t l = lo + tid;
  load_sc(l, sA);
  load_sc(l + TPB, sB);
  fetch_row(l, sA, cur, dcur);
  load_sc(l + 2 * TPB, sA);
  fetch_row(l + TPB, sB, n1, d1);
#pragma unroll
  for (int t = 0; t < 8; t++) sB[t] = sA[t];
  while (l < hi) {
    uint32_t n2[16];
    int d2;
    load_sc(l + 3 * TPB, sA);               // words for the pair three steps ahead
    fetch_row(l + 2 * TPB, sB, n2, d2);     // row for the pair two steps ahead, from the words requested an iteration ago
    if (dcur != 0) {
      Aff q;
      q.x = unpack<FP>(cur);
      q.y = unpack<FP>(cur + 8);
      if (dcur < 0) q.y = neg(q.y);
      acc = xyzz_madd(acc, q);
    }
#pragma unroll
    for (int t = 0; t < 16; t++) { cur[t] = n1[t]; n1[t] = n2[t]; }
#pragma unroll
    for (int t = 0; t < 8; t++) sB[t] = sA[t];
    dcur = d1; d1 = d2;
    l += TPB;
  }
  Jac accj = block_sum<TPB>(xyzz_to_jac(acc), red);
  if (tid == 0) raw_store(&out[(size_t)blockIdx.y * gridDim.x + blockIdx.x], accj);
}
template <int C, int LPM>
__global__ void __launch_bounds__(64) k_fixed_msm_small(const AffDev *table, size_t n, size_t cap, const uint32_t *scalars,
                                                        size_t sc_stride, JacRaw *out, size_t nb) {
  fixed_small_body<C, LPM>(table, n, cap, scalars, sc_stride, out, nb, blockIdx.x);
}
// out[i] = scalars[i] * P_0 by table lookups, one lane per scalar (GeneratorsChain::next, generators.rs:112-124:
// every Bulletproofs generator is a hashed scalar times the curve generator -- SURVEY 8f N2)
template <int C>
__global__ void __launch_bounds__(64) k_fixed_single(const AffDev *table, const uint32_t *scalars, JacRaw *out, size_t n) {
  constexpr int W = num_windows<C>();
  constexpr int HALF = 1 << (C - 1);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8], r[9];
#pragma unroll
  for (int t = 0; t < 8; t++) s[t] = scalars[i * 8 + t];
  recode_add_k<C>(r, s);
  Jac acc = jac_inf();
#pragma unroll 1
  for (int w = 0; w < W; w++) {
    int dg = recode_digit<C>(r, w);
    if (dg != 0) {
      Aff q = aff_load(table + (size_t)w * HALF + ((dg < 0 ? -dg : dg) - 1));
      if (dg < 0) q.y = neg(q.y);
      acc = jac_madd(acc, q);
    }
  }
  raw_store(&out[i], acc);
}
void fixed_single16(hipStream_t st, const AffDev *table, const uint32_t *scalars, JacRaw *out, size_t n) {
  if (!n) return;
  hipLaunchKernelGGL((k_fixed_single<16>), dim3((n + 63) / 64), dim3(64), 0, st, table, scalars, out, n);
}

// The L / R MSMs of an IPP round over resident generators (k_ipp_gens_scalars' compact layout: B, n0/2 G terms, n0/2 H
// terms per MSM; MSM 2p = L_p, 2p + 1 = R_p).  Block (chunk, msm) as k_fixed_msm; only non-zero terms are walked.
template <int C, int TPB>
__global__ void __launch_bounds__(TPB) k_fixed_msm_ipp(const AffDev *table, size_t n0, size_t cap, size_t cur,
                                                       const uint32_t *scalars, JacRaw *out, size_t pairs_per_chunk) {
  __shared__ int32_t red[27 * (TPB / 2)];
  constexpr int W = num_windows<C>();
  constexpr int HALF = 1 << (C - 1);
  const int tid = threadIdx.x;
  const size_t per = 1 + n0, half = n0 / 2, h = cur / 2;
  const uint32_t *sc = scalars + (size_t)blockIdx.y * per * 8;
  const bool is_R = (blockIdx.y & 1) != 0;
  const size_t total = per * W;
  const size_t lo = (size_t)blockIdx.x * pairs_per_chunk;
  const size_t hi = lo + pairs_per_chunk < total ? lo + pairs_per_chunk : total;
  Xyzz acc = xyzz_inf();
  // staged prefetch as in k_fixed_msm
  auto load_sc = [&](size_t ll, uint32_t *s) {
    if (ll < hi) {
      const size_t t = ll / W;
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = sc[t * 8 + k];
    }
  };
  auto fetch_row = [&](size_t ll, const uint32_t *s, uint32_t *dst, int &dg) {
    dg = 0;
    if (ll < hi) {
      size_t t = ll / W;
      int w = (int)(ll - t * W);
      uint32_t r[9];
      recode_add_k<C>(r, s);
      dg = recode_digit<C>(r, w);
      if (dg != 0) {
        size_t gen = 0;                                    // B
        if (t > 0) {
          const bool isH = t - 1 >= half;
          const size_t j = isH ? t - 1 - half : t - 1;
          const bool use_hi = is_R ? isH : !isH;           // L: G_hi, H_lo;  R: G_lo, H_hi
          const size_t i = (j / h) * cur + (use_hi ? h : 0) + j % h;
          gen = (isH ? 2 + cap : 2) + i;
        }
        const AffDev *e = table + (gen * W + w) * HALF + ((dg < 0 ? -dg : dg) - 1);
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = e->w[k];
      }
    }
  };
  // rows two pairs ahead, scalar words three (fixed_body.cuh)
  uint32_t curw[16], n1[16], sA[8], sB[8];
  int dcur = 0, d1 = 0;
  size_t l = lo + tid;
  load_sc(l, sA);
  load_sc(l + TPB, sB);
  fetch_row(l, sA, curw, dcur);
  load_sc(l + 2 * TPB, sA);
  fetch_row(l + TPB, sB, n1, d1);
#pragma unroll
  for (int k = 0; k < 8; k++) sB[k] = sA[k];
  while (l < hi) {
    uint32_t n2[16];
    int d2;
    load_sc(l + 3 * TPB, sA);
    fetch_row(l + 2 * TPB, sB, n2, d2);
    if (dcur != 0) {
      Aff q;
      q.x = unpack<FP>(curw);
      q.y = unpack<FP>(curw + 8);
      if (dcur < 0) q.y = neg(q.y);
      acc = xyzz_madd(acc, q);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) { curw[k] = n1[k]; n1[k] = n2[k]; }
#pragma unroll
    for (int k = 0; k < 8; k++) sB[k] = sA[k];
    dcur = d1; d1 = d2;
    l += TPB;
  }
  Jac accj = block_sum<TPB>(xyzz_to_jac(acc), red);
  if (tid == 0) raw_store(&out[(size_t)blockIdx.y * gridDim.x + blockIdx.x], accj);
}

// The same MSMs with the lanes of a wave spread over EIGHT MSMs x eight pair-lanes (a unit = 8 provers' L, or 8 provers' R, MSMs:
// they walk the same generators).  Why: a (generator, window) sub-table of the 16-bit-window table is exactly 2 MB -- one page
// fragment -- and the table is 69 GB (32 800 of them): with 64 lanes on 64 different pairs every wave-load touches 64 pages, far
// beyond any TLB's reach, and the kernel spends a third of its wave-cycles waiting although rows are requested two pairs ahead.
// Here the eight MSM lanes of a pair read eight random rows of ONE page: 8 pages per wave-load, and the units of a launch sweep
// the table in step.  The sum over the pair-lanes is a 3-level shuffle butterfly inside the wave: no LDS, no barrier.
template <int C>
__global__ void __launch_bounds__(64) k_fixed_msm_ipp_g(const AffDev *table, size_t n0, size_t cap, size_t cur, const uint32_t *scalars,
                                                        JacRaw *out, size_t nmsm, size_t pairs_per_chunk) {
  constexpr int W = num_windows<C>();
  constexpr int HALF = 1 << (C - 1);
  constexpr int PL = 8;                                   // pair-lanes per MSM
  const int pl = threadIdx.x >> 3, mj = threadIdx.x & 7;
  const bool is_R = (blockIdx.y & 1) != 0;
  size_t msm = 2 * ((size_t)(blockIdx.y >> 1) * 8 + mj) + (is_R ? 1 : 0);
  const bool live = msm < nmsm;
  if (!live) msm = is_R ? 1 : 0;                          // (whole waves stay active; a clamped lane's work is discarded)
  const size_t per = 1 + n0, half = n0 / 2, h = cur / 2;
  const uint32_t *sc = scalars + msm * per * 8;
  const size_t total = per * W;
  const size_t lo = (size_t)blockIdx.x * pairs_per_chunk;
  const size_t hi = lo + pairs_per_chunk < total ? lo + pairs_per_chunk : total;
  Xyzz acc = xyzz_inf();
  auto load_sc = [&](size_t ll, uint32_t *s) {
    if (ll < hi) {
      const size_t t = ll / W;
#pragma unroll
      for (int k = 0; k < 8; k++) s[k] = sc[t * 8 + k];
    }
  };
  auto fetch_row = [&](size_t ll, const uint32_t *s, uint32_t *dst, int &dg) {
    dg = 0;
    if (ll < hi) {
      size_t t = ll / W;
      int w = (int)(ll - t * W);
      uint32_t r[9];
      recode_add_k<C>(r, s);
      dg = recode_digit<C>(r, w);
      if (dg != 0) {
        size_t gen = 0;                                    // B
        if (t > 0) {
          const bool isH = t - 1 >= half;
          const size_t j = isH ? t - 1 - half : t - 1;
          const bool use_hi = is_R ? isH : !isH;           // L: G_hi, H_lo;  R: G_lo, H_hi
          const size_t i = (j / h) * cur + (use_hi ? h : 0) + j % h;
          gen = (isH ? 2 + cap : 2) + i;
        }
        const AffDev *e = table + (gen * W + w) * HALF + ((dg < 0 ? -dg : dg) - 1);
#pragma unroll
        for (int k = 0; k < 16; k++) dst[k] = e->w[k];
      }
    }
  };
  // rows two pairs ahead, scalar words three (as k_fixed_msm_ipp)
  uint32_t curw[16], n1[16], sA[8], sB[8];
  int dcur = 0, d1 = 0;
  size_t l = lo + pl;
  load_sc(l, sA);
  load_sc(l + PL, sB);
  fetch_row(l, sA, curw, dcur);
  load_sc(l + 2 * PL, sA);
  fetch_row(l + PL, sB, n1, d1);
#pragma unroll
  for (int k = 0; k < 8; k++) sB[k] = sA[k];
  while (l < hi) {
    uint32_t n2[16];
    int d2;
    load_sc(l + 3 * PL, sA);
    fetch_row(l + 2 * PL, sB, n2, d2);
    if (dcur != 0) {
      Aff q;
      q.x = unpack<FP>(curw);
      q.y = unpack<FP>(curw + 8);
      if (dcur < 0) q.y = neg(q.y);
      acc = xyzz_madd(acc, q);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) { curw[k] = n1[k]; n1[k] = n2[k]; }
#pragma unroll
    for (int k = 0; k < 8; k++) sB[k] = sA[k];
    dcur = d1; d1 = d2;
    l += PL;
  }
  Jac accj = xyzz_to_jac(acc);
#pragma unroll 1
  for (int off = 8; off < 64; off <<= 1) {               // over the pair-lanes: lanes mj, mj + 8, ... of the wave
    Jac q;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      q.X.v[t] = __shfl_xor(accj.X.v[t], off, 64);
      q.Y.v[t] = __shfl_xor(accj.Y.v[t], off, 64);
      q.Z.v[t] = __shfl_xor(accj.Z.v[t], off, 64);
    }
    accj = jac_add(accj, q);
  }
  if (pl == 0 && live) raw_store(&out[msm * gridDim.x + blockIdx.x], accj);
}
static bool ipp_grouped(size_t nmsm) { return nmsm >= 16; }   // (fewer MSMs than lanes of a unit: the block-per-chunk kernel)
size_t fixed_msm_ipp_chunks(int c, size_t n0, size_t nmsm) {
  size_t total = (1 + n0) * (252 / c + 1);
  if (ipp_grouped(nmsm)) {       // a wave = 8 MSMs x 8 pair-lanes: ~2 048 waves on the chip, at least 32 pairs per lane
    const size_t units = 2 * ((nmsm / 2 + 7) / 8);
    size_t by_fill = (2048 + units - 1) / units, by_work = (total + 255) / 256;
    size_t chg = by_work < by_fill ? by_work : by_fill;
    return chg ? chg : 1;
  }
  const size_t fill = 1024;   // blocks wanted on the chip (2 048: 7 % slower rounds for 256 provers -- the 7-level block sum weighs more on shorter lanes)
  size_t by_work = (total + 511) / 512, by_fill = (fill + nmsm - 1) / (nmsm ? nmsm : 1);
  size_t ch = by_work < by_fill ? by_work : by_fill;
  return ch ? ch : 1;
}
template <int C>
static void launch_fixed_ipp(hipStream_t st, const AffDev *table, size_t n0, size_t cap, size_t cur, const uint32_t *scalars,
                             JacRaw *dst, size_t nmsm, size_t chunks) {
  size_t total = (1 + n0) * num_windows<C>();
  size_t per = (total + chunks - 1) / chunks;
  if (ipp_grouped(nmsm)) {
    per = (per + 7) & ~(size_t)7;                          // (a multiple of the pair-lanes; the last chunk may be short or empty)
    hipLaunchKernelGGL((k_fixed_msm_ipp_g<C>), dim3(chunks, 2 * ((nmsm / 2 + 7) / 8)), dim3(64), 0, st, table, n0, cap, cur, scalars, dst, nmsm, per);
    return;
  }
  hipLaunchKernelGGL((k_fixed_msm_ipp<C, 128>), dim3(chunks, nmsm), dim3(128), 0, st, table, n0, cap, cur, scalars, dst, per);
}
// partials: nmsm * fixed_msm_ipp_chunks(c, n0, nmsm) points (unused when that is 1)
void fixed_msm_ipp(hipStream_t st, int c, const AffDev *table, size_t n0, size_t cap, size_t cur, const uint32_t *scalars,
                   JacRaw *out, size_t nmsm, JacRaw *partials, bool sum_partials) {
  if (!nmsm) return;
  size_t chunks = partials ? fixed_msm_ipp_chunks(c, n0, nmsm) : 1;
  JacRaw *dst = chunks > 1 ? partials : out;
  switch (c) {
    case 4: launch_fixed_ipp<4>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    case 8: launch_fixed_ipp<8>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    case 10: launch_fixed_ipp<10>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    case 12: launch_fixed_ipp<12>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    case 14: launch_fixed_ipp<14>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    case 16: launch_fixed_ipp<16>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    case 20: launch_fixed_ipp<20>(st, table, n0, cap, cur, scalars, dst, nmsm, chunks); break;
    default: return;
  }
  if (chunks > 1 && sum_partials) segmented_sum(st, partials, out, nmsm, chunks);
}

// Many large MSMs over the same generators (a batch's A_I, A_O, S commitments: 3 x 256 MSMs of 2 050 terms) with an MSM per LANE and a run
// of generators per wave (fixed_chunk_body): one table page per wave-load, the scalar recoded once per generator, no block sum -- and,
// with the MSMs of a wave all of one class (`kinds`), the windows of a bit vector that are zero in every lane cost nothing.  In the
// block-per-MSM kernel above a lane is a (generator, window) pair: the one window of a bit vector that is not zero keeps 4 lanes of
// every wave adding, so A_I and A_O cost as much as the dense S.
template <int C>
__global__ void __launch_bounds__(64) k_fixed_msm_m(const AffDev *table, size_t n, size_t cap, const uint32_t *scalars, size_t sc_stride,
                                                    JacRaw *part, size_t nb, unsigned chunks, unsigned gens_per_chunk, unsigned kinds) {
  fixed_chunk_body<C, 1>(table, n, cap, scalars, sc_stride, part, nb, chunks, gens_per_chunk, (size_t)blockIdx.y / kinds * chunks + blockIdx.x,
                         kinds, blockIdx.y % kinds);
}
static bool fixed_per_lane(size_t n, size_t nb, int kinds) { return kinds > 0 && nb >= 64 * (size_t)kinds && n >= 64; }
static unsigned fixed_per_lane_gens(size_t n, size_t nb, int kinds) {   // ~2 048 waves of ONE class on the chip, at least 2 generators per wave
  const size_t sets = (nb / kinds + 63) / 64;
  size_t target = 2048 / sets;
  if (target < 1) target = 1;
  if (target > 1024) target = 1024;
  const size_t g = (2 + 2 * n + target - 1) / target;
  return (unsigned)(g < 2 ? 2 : g);
}
// chunks per MSM: enough blocks to fill the chip when there are few MSMs, at least 4 pairs per lane
size_t fixed_msm_chunks(int c, size_t n, size_t nb, int kinds) {
  if (fixed_per_lane(n, nb, kinds)) { const size_t g = fixed_per_lane_gens(n, nb, kinds); return (2 + 2 * n + g - 1) / g; }
  size_t total = (2 + 2 * n) * (252 / c + 1);
  size_t by_work = (total + 511) / 512, by_fill = (1024 + nb - 1) / (nb ? nb : 1);
  size_t ch = by_work < by_fill ? by_work : by_fill;
  return ch ? ch : 1;
}
template <int C>
static void launch_fixed(hipStream_t st, const AffDev *table, size_t n, size_t cap, const uint32_t *scalars, size_t stride,
                         JacRaw *out, size_t nb, size_t chunks, int lpm, int kinds) {
  constexpr int TPB = 128;
  size_t total = (2 + 2 * n) * num_windows<C>();
  if (fixed_per_lane(n, nb, kinds) && chunks > 1) {
    const unsigned sets = (unsigned)((nb / kinds + 63) / 64);
    hipLaunchKernelGGL((k_fixed_msm_m<C>), dim3((unsigned)chunks, sets * kinds), dim3(64), 0, st, table, n, cap, scalars, stride, out, nb,
                       (unsigned)chunks, fixed_per_lane_gens(n, nb, kinds), (unsigned)kinds);
    return;
  }
  if (chunks == 1 && nb >= 64 && total <= 16384) {
    if (lpm == 64) hipLaunchKernelGGL((k_fixed_msm_small<C, 64>), dim3(nb), dim3(64), 0, st, table, n, cap, scalars, stride, out, nb);
    else if (nb >= 1024 && lpm != 32) hipLaunchKernelGGL((k_fixed_msm_small<C, 16>), dim3((nb + 3) / 4), dim3(64), 0, st, table, n, cap, scalars, stride, out, nb);
    else hipLaunchKernelGGL((k_fixed_msm_small<C, 32>), dim3((nb + 1) / 2), dim3(64), 0, st, table, n, cap, scalars, stride, out, nb);
    return;
  }
  size_t per = (total + chunks - 1) / chunks;
  hipLaunchKernelGGL((k_fixed_msm<C, TPB>), dim3(chunks, nb), dim3(TPB), 0, st, table, n, cap, scalars, stride, out, per);
}
void fixed_msm(hipStream_t st, int c, const AffDev *table, size_t n, size_t cap, const uint32_t *scalars,
               size_t stride, JacRaw *out, size_t nb, JacRaw *partials, int lpm, int kinds) {
  if (!nb) return;
  size_t chunks = partials ? fixed_msm_chunks(c, n, nb, kinds) : 1;
  JacRaw *dst = chunks > 1 ? partials : out;
  switch (c) {
    case 4: launch_fixed<4>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    case 8: launch_fixed<8>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    case 10: launch_fixed<10>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    case 12: launch_fixed<12>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    case 14: launch_fixed<14>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    case 16: launch_fixed<16>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    case 20: launch_fixed<20>(st, table, n, cap, scalars, stride, dst, nb, chunks, lpm, kinds); break;
    default: return;   // rejected by the C-ABI before reaching here
  }
  if (chunks > 1) segmented_sum(st, partials, out, nb, chunks);
}

}  // namespace bpk
