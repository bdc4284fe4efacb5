// k_ec_misc.hip -- the colder elliptic-curve kernels (import/export, normalisation, table build,
// verification tail), compiled with the field multiply inlined.  Split from k_ec.hip because the
// ROCm 7.2 backend crashes (Machine Copy Propagation) on k_verify_finalize with out-of-line calls.
// k_ec.hip -- elliptic-curve kernels for gfx950 (wave64): point import/export, per-lane Straus
// scalar multiplication with LDS-resident tables, segmented point sums, signed fixed-window
// fixed-base tables + lookup MSM, verification tail.
//
// Hot-path rows: a1 StarkPoint::msm_iter / msm, a2 fold_witness (point half),
// a3 first-round generator scaling, a9 mega_check.  All integer work (F_p, 9 x 29-bit limbs); the
// kernels are VALU-integer bound (v_mad_u64_u32), not HBM bound -- DESIGN.md has the numbers.
#include "ec_dev.cuh"

using namespace bp;

namespace bpk {

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_points_from_boundary(const Words8 *xy, AffDev *out, size_t n, int *bad, int32_t *bad_unit,
                                                              size_t per_unit) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 8; j++) { w[j] = xy[2 * i].w[j]; w[8 + j] = xy[2 * i + 1].w[j]; }
  Aff a;
  bool ok = aff_from_boundary(a, w);
  if (!ok) {
    atomicOr(bad, 1);
    if (bad_unit) bad_unit[i / per_unit] = 1;
    a.x = fe_zero<FP>();
    a.y = fe_zero<FP>();
  }
  aff_store(&out[i], a);
}
void points_from_boundary(hipStream_t st, const Words8 *xy, AffDev *out, size_t n, int *bad, int32_t *bad_unit, size_t per_unit) {
  if (!n) return;
  hipLaunchKernelGGL(k_points_from_boundary, dim3((n + 255) / 256), dim3(256), 0, st, xy, out, n, bad, bad_unit, per_unit ? per_unit : 1);
}

__global__ void __launch_bounds__(64) k_jac_to_boundary(const JacRaw *in, Words8 *xy, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac p = raw_load(&in[i]);
  if (!jac_is_inf(p) && is_zero_exact(p.Z)) p = jac_inf();
  uint32_t w[16];
  aff_to_boundary(w, jac_to_aff(p));
#pragma unroll
  for (int j = 0; j < 8; j++) { xy[2 * i].w[j] = w[j]; xy[2 * i + 1].w[j] = w[8 + j]; }
}
void jac_to_boundary(hipStream_t st, const JacRaw *in, Words8 *xy, size_t n) {
  if (!n) return;
  hipLaunchKernelGGL(k_jac_to_boundary, dim3((n + 63) / 64), dim3(64), 0, st, in, xy, n);
}

// Montgomery's trick over RUN consecutive points per lane
template <int RUN>
__global__ void __launch_bounds__(128) k_batch_normalize(const JacRaw *in, AffDev *out, size_t n) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t base = r * RUN;
  if (base >= n) return;
  Fp pref[RUN];
  Fp acc = fe_one<FP>();
#pragma unroll
  for (int i = 0; i < RUN; i++) {
    pref[i] = acc;
    if (base + i < n) {
      Fp z;
#pragma unroll
      for (int j = 0; j < NL; j++) z.v[j] = in[base + i].v[2 * NL + j];
      if (!is_zero_limbs(z)) acc = mul(acc, z);
    }
  }
  Fp ai = inv(acc);
#pragma unroll
  for (int i = RUN - 1; i >= 0; i--) {
    if (base + i < n) {
      Jac p = raw_load(&in[base + i]);
      Aff a;
      if (jac_is_inf(p)) {
        a.x = fe_zero<FP>();
        a.y = fe_zero<FP>();
      } else {
        Fp zi = mul(ai, pref[i]);
        ai = mul(ai, p.Z);
        a = jac_to_aff_with_zinv(p, zi);
      }
      aff_store(&out[base + i], a);
    }
  }
}
void batch_normalize(hipStream_t st, const JacRaw *in, AffDev *out, size_t n, int run) {
  if (!n) return;
  (void)run;
  size_t lanes = (n + 7) / 8;
  hipLaunchKernelGGL(k_batch_normalize<8>, dim3((lanes + 127) / 128), dim3(128), 0, st, in, out, n);
}

// ------------------------------------------------------------------------------------------------
// fixed-base tables
size_t fixed_table_entries(int c, size_t ngens) { return ngens * (size_t)(252 / c + 1) << (c - 1); }

// base[g*W + w] = 2^(c w) P_g
__global__ void __launch_bounds__(64) k_tab_bases(int c, int W, const AffDev *gens, size_t ngens, JacRaw *bases) {
  size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngens) return;
  Jac b = jac_from_aff(aff_load(&gens[g]));
  for (int w = 0; w < W; w++) {
    raw_store(&bases[g * W + w], b);
    for (int d = 0; d < c; d++) b = jac_dbl(b);
  }
}
// fill[(l << (c-1)) + r*8 + i] = (r*8 + i + 1) * base[l]
__global__ void __launch_bounds__(128) k_tab_fill(int c, size_t nbases, const JacRaw *bases, JacRaw *fill) {
  const int runs = 1 << (c - 1 - 3);
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbases * runs) return;
  size_t l = t / runs;
  int r = (int)(t % runs);
  Jac base = raw_load(&bases[l]);
  unsigned d0 = (unsigned)r * 8 + 1;
  Jac acc = jac_inf();
  for (int bit = c - 1; bit >= 0; bit--) {   // d0 < 2^(c-1)
    acc = jac_dbl(acc);
    if ((d0 >> bit) & 1) acc = jac_add(acc, base);
  }
  JacRaw *dst = fill + (l << (c - 1)) + (size_t)r * 8;
  for (int i = 0; i < 8; i++) {
    raw_store(&dst[i], acc);
    if (i < 7) acc = jac_add(acc, base);
  }
}
void fixed_table_build(hipStream_t st, int c, const AffDev *gens, size_t ngens, AffDev *table, JacRaw *scratch) {
  const int W = 252 / c + 1;
  size_t nbases = ngens * W, entries = nbases << (c - 1);
  JacRaw *bases = scratch, *fill = scratch + nbases;
  hipLaunchKernelGGL(k_tab_bases, dim3((ngens + 63) / 64), dim3(64), 0, st, c, W, gens, ngens, bases);
  size_t threads = nbases << (c - 1 - 3);
  hipLaunchKernelGGL(k_tab_fill, dim3((threads + 127) / 128), dim3(128), 0, st, c, nbases, bases, fill);
  batch_normalize(st, fill, table, entries, 8);
}

// ------------------------------------------------------------------------------------------------
// 32 lanes per proof (two proofs per wave): gather, butterfly-reduce with wave shuffles, test identity.
__global__ void __launch_bounds__(64) k_verify_finalize(const JacRaw *var, size_t nvar, const JacRaw *fixed,
                                                        size_t nb, int32_t *ok, Words8 *mega, const int32_t *bad_sc,
                                                        const int32_t *bad_pt) {
  __builtin_amdgcn_s_setprio(2);   // last link of the per-batch chain: finish ahead of other batches' bulk MSM waves
  const int lane = threadIdx.x & 31;
  size_t p = (size_t)blockIdx.x * 2 + (threadIdx.x >> 5);
  const bool live = p < nb;
  if (!live) p = nb - 1;
  Jac acc = jac_inf();
  for (size_t v = lane; v < nvar + 1; v += 32) {
    Jac q = raw_load(v < nvar ? &var[p * nvar + v] : &fixed[p]);
    acc = jac_add(acc, q);
  }
#pragma unroll 1
  for (int off = 16; off > 0; off >>= 1) {
    Jac q;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      q.X.v[t] = __shfl_xor(acc.X.v[t], off, 64);
      q.Y.v[t] = __shfl_xor(acc.Y.v[t], off, 64);
      q.Z.v[t] = __shfl_xor(acc.Z.v[t], off, 64);
    }
    acc = jac_add(acc, q);
  }
  if (lane == 0 && live) {
    bool inf = jac_is_inf(acc) || is_zero_exact(acc.Z);
    const bool malformed = (bad_sc && bad_sc[p]) || (bad_pt && bad_pt[p]);
    ok[p] = (inf && !malformed) ? 1 : 0;
    if (mega) {
      uint32_t w[16];
      if (inf) {
#pragma unroll
        for (int j = 0; j < 16; j++) w[j] = 0;
      } else {
        aff_to_boundary(w, jac_to_aff(acc));
      }
#pragma unroll
      for (int j = 0; j < 8; j++) { mega[2 * p].w[j] = w[j]; mega[2 * p + 1].w[j] = w[8 + j]; }
    }
  }
}
void verify_finalize(hipStream_t st, const JacRaw *var, size_t nvar, const JacRaw *fixed, size_t nb,
                     int32_t *ok, Words8 *mega, const int32_t *bad_sc, const int32_t *bad_pt) {
  if (!nb) return;
  hipLaunchKernelGGL(k_verify_finalize, dim3((nb + 1) / 2), dim3(64), 0, st, var, nvar, fixed, nb, ok, mega, bad_sc, bad_pt);
}

// sum of n boundary points (validated here), one block: lane-strided mixed additions, LDS tree, boundary bytes out
__global__ void __launch_bounds__(128) k_points_sum(const Words8 *xy, size_t n, Words8 *out_xy, int *bad) {
  __shared__ int32_t smem[27 * 64];
  Jac acc = jac_inf();
  for (size_t i = threadIdx.x; i < n; i += 128) {
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 8; j++) { w[j] = xy[2 * i].w[j]; w[8 + j] = xy[2 * i + 1].w[j]; }
    Aff a;
    if (!aff_from_boundary(a, w)) { atomicOr(bad, 1); continue; }
    acc = jac_madd(acc, a);
  }
  acc = block_sum<128>(acc, smem);
  if (threadIdx.x == 0) {
    if (!jac_is_inf(acc) && is_zero_exact(acc.Z)) acc = jac_inf();
    uint32_t w[16];
    aff_to_boundary(w, jac_to_aff(acc));
#pragma unroll
    for (int j = 0; j < 8; j++) { out_xy[0].w[j] = w[j]; out_xy[1].w[j] = w[8 + j]; }
  }
}
void points_sum(hipStream_t st, const Words8 *xy, size_t n, Words8 *out_xy, int *bad) {
  hipLaunchKernelGGL(k_points_sum, dim3(1), dim3(128), 0, st, xy, n, out_xy, bad);
}

// dst[p * dst_outer + i] = src[p * src_outer + i], i < cnt (16-byte vector copies)
__global__ void __launch_bounds__(256) k_gather16(const uint4 *src, size_t src_outer, size_t cnt, uint4 *dst, size_t dst_outer, int vec) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (t >= cnt * vec) return;
  dst[p * dst_outer * vec + t] = src[p * src_outer * vec + t];
}
void gather_points(hipStream_t st, const AffDev *src, size_t src_outer, size_t cnt, size_t nb, AffDev *dst, size_t dst_outer) {
  if (!nb || !cnt) return;
  hipLaunchKernelGGL(k_gather16, dim3((cnt * 4 + 255) / 256, nb), dim3(256), 0, st, (const uint4 *)src, src_outer, cnt, (uint4 *)dst, dst_outer, 4);
}
void gather_scalars(hipStream_t st, const Words8 *src, size_t src_outer, size_t cnt, size_t nb, Words8 *dst, size_t dst_outer) {
  if (!nb || !cnt) return;
  hipLaunchKernelGGL(k_gather16, dim3((cnt * 2 + 255) / 256, nb), dim3(256), 0, st, (const uint4 *)src, src_outer, cnt, (uint4 *)dst, dst_outer, 2);
}

}  // namespace bpk
