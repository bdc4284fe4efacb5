// k_scalar.hip -- scalar-field (F_n) kernels: canonical checks, batched inversion, inner products,
// IPP scalar fold, verification scalars, constraint flattening, verifier scalar assembly.
//
// Hot-path rows: a2 (scalar half of fold_witness), a4 inner_product, a5
// verification_scalars, a6 batch_inverse, a7 flattened_constraints, a9 verifier scalar assembly.
#include <cstdlib>
#include "fn_dev.cuh"
#include "vs_prep.cuh"

using namespace bp;

namespace bpk {

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scalars_check(const Words8 *in, size_t n, int *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = in[i].w[j];
  if (!words_lt_mod<FN>(w)) atomicOr(bad, 1);
}
// the same with the verdict attributed to a proof: element i belongs to proof i / per; writes 1 only (zero the array first)
__global__ void __launch_bounds__(256) k_scalars_check_proof(const Words8 *in, size_t n, size_t per, int *bad, int32_t *bad_proof) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = in[i].w[j];
  if (!words_lt_mod<FN>(w)) { if (bad) atomicOr(bad, 1); bad_proof[i / per] = 1; }
}
void scalars_check(hipStream_t st, const Words8 *in, size_t n, int *bad) {
  if (!n) return;
  hipLaunchKernelGGL(k_scalars_check, dim3((n + 255) / 256), dim3(256), 0, st, in, n, bad);
}
// ---- constraint rows (CSR, as the reference holds its Vec<LinearCombination>) -> column-major by output variable, on the device.
// Outputs are ordered wL[0..n) wR[0..n) wO[0..n) wV[0..m) wc; within a column the order of the terms is whatever the atomics
// give: the flattened weights are exact sums in F_n, so they do not depend on it.
__device__ __forceinline__ bool csr_out_of(uint32_t kd, uint32_t ix, size_t n_mul, size_t m, size_t &o) {
  if (kd <= 2) { if (ix >= n_mul) return false; o = (size_t)kd * n_mul + ix; return true; }
  if (kd == 3) { if (ix >= m) return false; o = 3 * n_mul + ix; return true; }
  if (kd == 4) { o = 3 * n_mul + m; return true; }
  return false;
}
// Terms of the CONSTANT column are counted / placed per wave, not per lane: a gadget like the shuffle puts a constant into every
// second row, and 32 766 atomics on one counter took 0.37 ms in either kernel.
__global__ void __launch_bounds__(256) k_csr_count(size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx, size_t n_mul,
                                                   size_t m, uint32_t *cnt, int *bad) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= q) return;
  const size_t oc = 3 * n_mul + m;
  for (uint32_t t = row_ptr[r]; t < row_ptr[r + 1]; t++) {
    size_t o;
    const bool ok = csr_out_of(kind[t], idx[t], n_mul, m, o);
    const bool is_c = ok && o == oc;
    const unsigned long long mc = __ballot(is_c);
    if (!ok) { atomicOr(bad, 1); continue; }
    if (is_c) {
      if ((int)__lane_id() == __ffsll((long long)mc) - 1) atomicAdd(&cnt[oc + 1], (uint32_t)__popcll(mc));
    } else atomicAdd(&cnt[o + 1], 1u);
  }
}
// exclusive scan in place over cnt[0 .. n] (cnt[0] = 0 on entry): ONE block of 1024 lanes, each owning a contiguous run of a
// multiple of four entries (16-byte loads, several in flight; cnt is 256-byte aligned)
__global__ void __launch_bounds__(1024) k_csr_scan(uint32_t *cnt, size_t n1) {
  __shared__ uint32_t part[1024];
  const size_t per = ((n1 + 1023) / 1024 + 3) / 4 * 4, lo = threadIdx.x * per, hi = lo + per < n1 ? lo + per : n1;
  uint32_t s = 0;
  if (lo < hi) {
    const size_t full = lo + (hi - lo) / 4 * 4;
#pragma unroll 4
    for (size_t i = lo; i < full; i += 4) { const uint4 v = *(const uint4 *)(cnt + i); s += v.x + v.y + v.z + v.w; }
    for (size_t i = full; i < hi; i++) s += cnt[i];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  if (lo < hi) {                                                          // inclusive over cnt[1..]: cnt[o + 1] = end of column o
    const size_t full = lo + (hi - lo) / 4 * 4;
#pragma unroll 4
    for (size_t i = lo; i < full; i += 4) {
      uint4 v = *(const uint4 *)(cnt + i);
      v.x += run; v.y += v.x; v.z += v.y; v.w += v.z; run = v.w;
      *(uint4 *)(cnt + i) = v;
    }
    for (size_t i = full; i < hi; i++) { run += cnt[i]; cnt[i] = run; }
  }
}
template <bool ARK>
__global__ void __launch_bounds__(256) k_csr_scatter(size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx,
                                                     const Words8 *coeff_in, size_t n_mul, size_t m, const uint32_t *col_ptr, uint32_t *fill,
                                                     uint32_t *rows, Words8 *coeff_out, int *bad) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= q) return;
  for (uint32_t t = row_ptr[r]; t < row_ptr[r + 1]; t++) {
    size_t o;
    const bool ok = csr_out_of(kind[t], idx[t], n_mul, m, o);
    const bool is_c = ok && o == 3 * n_mul + m;
    const unsigned long long mc = __ballot(is_c);
    if (!ok) continue;
    uint32_t pos;
    if (is_c) {                           // one atomic per wave for the constant column; a lane's slot = its rank among the wave's
      const int leader = __ffsll((long long)mc) - 1;
      uint32_t base = 0;
      if ((int)__lane_id() == leader) base = atomicAdd(&fill[o], (uint32_t)__popcll(mc));
      base = __shfl(base, leader, 64);
      pos = col_ptr[o] + base + (uint32_t)__popcll(mc & ((1ull << __lane_id()) - 1));
    } else pos = col_ptr[o] + atomicAdd(&fill[o], 1u);
    rows[pos] = (uint32_t)r;
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = coeff_in[t].w[j];
    if (!words_lt_mod<FN>(w)) { atomicOr(bad, 1); continue; }
    Fn x;
    if (ARK) {       // ark-ff Montgomery limbs (x 2^256 mod n): one multiplication by 2^266 instead of the host's de-Montgomery
      constexpr int32_t C[NL] = FN_ARK_MONT;
      Fn k;
#pragma unroll
      for (int j = 0; j < NL; j++) k.v[j] = C[j];
      x = canon(mul(unpack<FN>(w), k));
    } else x = canon(to_mont(unpack<FN>(w)));
    pack(w, x);
#pragma unroll
    for (int j = 0; j < 8; j++) coeff_out[pos].w[j] = w[j];
  }
}
void circuit_transpose(hipStream_t st, size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx, const Words8 *coeff_in,
                       size_t n_mul, size_t m, bool ark, uint32_t *col_ptr, uint32_t *fill, uint32_t *rows, Words8 *coeff_out, int *bad) {
  const size_t nout = 3 * n_mul + m + 1;
  (void)hipMemsetAsync(col_ptr, 0, (nout + 1) * 4, st);
  (void)hipMemsetAsync(fill, 0, nout * 4, st);
  if (!q) return;
  hipLaunchKernelGGL(k_csr_count, dim3((q + 255) / 256), dim3(256), 0, st, q, row_ptr, kind, idx, n_mul, m, col_ptr, bad);
  hipLaunchKernelGGL(k_csr_scan, dim3(1), dim3(1024), 0, st, col_ptr, nout + 1);
  if (ark) hipLaunchKernelGGL(k_csr_scatter<true>, dim3((q + 255) / 256), dim3(256), 0, st, q, row_ptr, kind, idx, coeff_in, n_mul, m, col_ptr, fill, rows, coeff_out, bad);
  else hipLaunchKernelGGL(k_csr_scatter<false>, dim3((q + 255) / 256), dim3(256), 0, st, q, row_ptr, kind, idx, coeff_in, n_mul, m, col_ptr, fill, rows, coeff_out, bad);
}
// rows of the commitment MSMs from the witness planes (see kernels.h): one lane per row element
__global__ void __launch_bounds__(256) k_commit_rows(size_t nb, size_t n, size_t lo, size_t stride, const Words8 *aL, const Words8 *aR,
                                                     const Words8 *aO, const Words8 *sL, const Words8 *sR, const Words8 *blinds, Words8 *rows,
                                                     size_t slo, size_t shi, int with_blind) {
  const size_t per = 2 + 2 * n;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * 3 * per) return;
  const size_t e = t % per, w = (t / per) % 3, p = t / (3 * per);
  const Words8 *src = nullptr;
  if (e == 1) { if (with_blind) src = blinds + p * 3 + w; }
  else if (e >= 2) {
    const size_t i = (e - 2) % n;
    const bool h = e - 2 >= n;
    if (i >= lo && i >= slo && i < shi) {
      const Words8 *pl = w == 0 ? (h ? aR : aL) : (w == 1 ? (h ? nullptr : aO) : (h ? sR : sL));
      if (pl) src = pl + p * stride + i;
    }
  }
  Words8 v{};
  if (src) v = *src;
  rows[t] = v;
}
void commit_rows(hipStream_t st, size_t nb, size_t n, size_t lo, size_t stride, const Words8 *aL, const Words8 *aR, const Words8 *aO,
                 const Words8 *sL, const Words8 *sR, const Words8 *blinds, Words8 *rows, size_t slo, size_t shi, bool with_blind) {
  const size_t tot = nb * 3 * (2 + 2 * n);
  if (tot) hipLaunchKernelGGL(k_commit_rows, dim3((tot + 255) / 256), dim3(256), 0, st, nb, n, lo, stride, aL, aR, aO, sL, sR, blinds, rows,
                              slo, shi, with_blind ? 1 : 0);
}
// one proof's MSM scalars restricted to a rank's share: fixed = [B, B_blinding, G_0..G_{np-1}, H_0..H_{np-1}] keeps the
// generators [slo, shi) (B, B_blinding on the rank with keep_pedersen), var = nvar proof-point scalars keeps [vlo, vhi)
__global__ void __launch_bounds__(256) k_shard_mask(Words8 *fixed, size_t np, size_t slo, size_t shi, int keep_pedersen, Words8 *var, size_t nvar,
                                                    size_t vlo, size_t vhi) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nfix = 2 + 2 * np;
  Words8 z{};
  if (t < nfix) {
    bool keep = t < 2 ? keep_pedersen != 0 : ((t - 2) % np >= slo && (t - 2) % np < shi);
    if (!keep) fixed[t] = z;
  } else if (t < nfix + nvar) {
    const size_t v = t - nfix;
    if (v < vlo || v >= vhi) var[v] = z;
  }
}
void shard_mask(hipStream_t st, Words8 *fixed, size_t np, size_t slo, size_t shi, bool keep_pedersen, Words8 *var, size_t nvar, size_t vlo,
                size_t vhi) {
  const size_t tot = 2 + 2 * np + nvar;
  hipLaunchKernelGGL(k_shard_mask, dim3((tot + 255) / 256), dim3(256), 0, st, fixed, np, slo, shi, keep_pedersen ? 1 : 0, var, nvar, vlo, vhi);
}
void scalars_check_proof(hipStream_t st, const Words8 *in, size_t n, size_t per_unit, int *bad, int32_t *bad_unit) {
  if (!n) return;
  hipLaunchKernelGGL(k_scalars_check_proof, dim3((n + 255) / 256), dim3(256), 0, st, in, n, per_unit, bad, bad_unit);
}

// Scalar::batch_inverse: Montgomery's trick, RUN elements per lane, one Fermat inversion per lane
template <int RUN>
__global__ void __launch_bounds__(64) k_batch_inverse(Words8 *io, size_t n, int *bad_zero) {
  size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * RUN;
  if (base >= n) return;
  Fn pref[RUN], val[RUN];
  Fn acc = fe_one<FN>();
#pragma unroll
  for (int i = 0; i < RUN; i++) {
    pref[i] = acc;
    if (base + i < n) {
      val[i] = load_plain(&io[base + i]);
      if (is_zero_exact(val[i])) { atomicOr(bad_zero, 1); val[i] = fe_one<FN>(); }
      acc = mul(acc, val[i]);
    }
  }
  Fn ai = inv(acc);
#pragma unroll
  for (int i = RUN - 1; i >= 0; i--) {
    if (base + i < n) {
      store_plain(&io[base + i], mul(ai, pref[i]));
      ai = mul(ai, val[i]);
    }
  }
}
void batch_inverse(hipStream_t st, Words8 *io, size_t n, int *bad_zero) {
  if (!n) return;
  size_t lanes = (n + 3) / 4;
  hipLaunchKernelGGL(k_batch_inverse<4>, dim3((lanes + 63) / 64), dim3(64), 0, st, io, n, bad_zero);
}

// inner_product: grid-stride partial sums, one raw partial per block, then a single-block finish
constexpr int IP_TPB = 256;
__global__ void __launch_bounds__(IP_TPB) k_inner_product_partial(const Words8 *a, const Words8 *b, size_t n, int32_t *partials) {
  __shared__ int32_t sm[NL * (IP_TPB / 64)];
  Fn acc = fe_zero<FN>();
  int cnt = 0;
  for (size_t i = (size_t)blockIdx.x * IP_TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * IP_TPB) {
    acc = add(acc, mul(load_plain(&a[i]), load_plain(&b[i])));
    if ((++cnt & 15) == 0) acc = fn_reduce(acc);
  }
  acc = wave_sum(fn_reduce(acc));
  int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) raw_put(sm + wv * NL, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    Fn t = raw_get(sm);
    for (int w = 1; w < IP_TPB / 64; w++) t = add(t, raw_get(sm + w * NL));
    raw_put(partials + (size_t)blockIdx.x * NL, fn_reduce(t));
  }
}
__global__ void __launch_bounds__(64) k_inner_product_finish(const int32_t *partials, int nparts, Words8 *out) {
  Fn acc = fe_zero<FN>();
  for (int i = threadIdx.x; i < nparts; i += 64) acc = add(acc, raw_get(partials + (size_t)i * NL));   // <= 16 reduced terms
  acc = wave_sum(fn_reduce(acc));
  if (threadIdx.x == 0) store_plain(out, acc);
}
static int ip_blocks(size_t n) {
  size_t b = (n + IP_TPB - 1) / IP_TPB;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
size_t inner_product_scratch_bytes(size_t n) { return (size_t)ip_blocks(n) * NL * 4; }
void inner_product(hipStream_t st, const Words8 *a, const Words8 *b, size_t n, Words8 *out, void *scratch) {
  int nb = ip_blocks(n);
  hipLaunchKernelGGL(k_inner_product_partial, dim3(nb), dim3(IP_TPB), 0, st, a, b, n, (int32_t *)scratch);
  hipLaunchKernelGGL(k_inner_product_finish, dim3(1), dim3(64), 0, st, (const int32_t *)scratch, nb, out);
}

// inner_product_proof.rs:224-225,239-240: a' = a_L u + u^-1 a_R ; b' = b_L u^-1 + u b_R
__global__ void __launch_bounds__(256) k_fold_scalars(size_t n, const Words8 *u, const Words8 *u_inv, const Words8 *a,
                                                      const Words8 *b, Words8 *a_out, Words8 *b_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fn uu = load_plain(u), ui = load_plain(u_inv);
  Fn aL = load_plain(&a[i]), aR = load_plain(&a[n + i]), bL = load_plain(&b[i]), bR = load_plain(&b[n + i]);
  store_plain(&a_out[i], add(mul(aL, uu), mul(ui, aR)));
  store_plain(&b_out[i], add(mul(bL, ui), mul(uu, bR)));
}
void fold_scalars(hipStream_t st, size_t n, const Words8 *u, const Words8 *u_inv, const Words8 *a,
                  const Words8 *b, Words8 *a_out, Words8 *b_out) {
  if (!n) return;
  hipLaunchKernelGGL(k_fold_scalars, dim3((n + 255) / 256), dim3(256), 0, st, n, u, u_inv, a, b, a_out, b_out);
}

// ---- batched (proof-major) IPP scalar kernels ---------------------------------------------------
__global__ void __launch_bounds__(256) k_sc_mul_strided(size_t cnt, const Words8 *x, size_t x_outer, size_t x_stride,
                                                        const Words8 *y, size_t y_outer, size_t y_stride, Words8 *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= cnt) return;
  store_plain(&out[p * cnt + i], mul(load_plain(&x[p * x_outer + i * x_stride]), load_plain(&y[p * y_outer + i * y_stride])));
}
void sc_mul_strided(hipStream_t st, size_t nb, size_t cnt, const Words8 *x, size_t x_outer, size_t x_stride,
                    const Words8 *y, size_t y_outer, size_t y_stride, Words8 *out) {
  if (!nb || !cnt) return;
  hipLaunchKernelGGL(k_sc_mul_strided, dim3((cnt + 255) / 256, nb), dim3(256), 0, st, cnt, x, x_outer, x_stride, y,
                     y_outer, y_stride, out);
}
__global__ void __launch_bounds__(256) k_sc_dot_batched(size_t cnt, const Words8 *x, size_t x_outer, const Words8 *y,
                                                        size_t y_outer, Words8 *out, size_t out_stride) {
  __shared__ int32_t sm[NL * 4];
  size_t p = blockIdx.x;
  Fn acc = fe_zero<FN>();
  int c = 0;
  for (size_t i = threadIdx.x; i < cnt; i += 256) {
    acc = add(acc, mul(load_plain(&x[p * x_outer + i]), load_plain(&y[p * y_outer + i])));
    if ((++c & 15) == 0) acc = fn_reduce(acc);
  }
  acc = wave_sum(fn_reduce(acc));
  if ((threadIdx.x & 63) == 0) raw_put(sm + (threadIdx.x >> 6) * NL, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    Fn t = raw_get(sm);
    for (int w = 1; w < 4; w++) t = add(t, raw_get(sm + w * NL));
    store_plain(&out[p * out_stride], t);
  }
}
void sc_dot_batched(hipStream_t st, size_t nb, size_t cnt, const Words8 *x, size_t x_outer, const Words8 *y,
                    size_t y_outer, Words8 *out, size_t out_stride) {
  if (!nb) return;
  hipLaunchKernelGGL(k_sc_dot_batched, dim3(nb), dim3(256), 0, st, cnt, x, x_outer, y, y_outer, out, out_stride);
}
__global__ void __launch_bounds__(256) k_fold_scalars_batched(size_t h, const Words8 *u, const Words8 *u_inv,
                                                              const Words8 *a, const Words8 *b, Words8 *a_out, Words8 *b_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= h) return;
  Fn uu = load_plain(&u[p]), ui = load_plain(&u_inv[p]);
  const Words8 *ap = a + p * 2 * h, *bp = b + p * 2 * h;
  store_plain(&a_out[p * h + i], add(mul(load_plain(&ap[i]), uu), mul(ui, load_plain(&ap[h + i]))));
  store_plain(&b_out[p * h + i], add(mul(load_plain(&bp[i]), ui), mul(uu, load_plain(&bp[h + i]))));
}
void fold_scalars_batched(hipStream_t st, size_t nb, size_t h, const Words8 *u, const Words8 *u_inv, const Words8 *a,
                          const Words8 *b, Words8 *a_out, Words8 *b_out) {
  if (!nb || !h) return;
  hipLaunchKernelGGL(k_fold_scalars_batched, dim3((h + 255) / 256, nb), dim3(256), 0, st, h, u, u_inv, a, b, a_out, b_out);
}

// combined batch check helpers: per-proof random weights rho_p
__global__ void __launch_bounds__(256) k_sc_scale_rows(size_t cnt, Words8 *x, const Words8 *w) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= cnt) return;
  store_plain(&x[p * cnt + i], mul(load_plain(&x[p * cnt + i]), load_plain(&w[p])));
}
void sc_scale_rows(hipStream_t st, size_t nb, size_t cnt, Words8 *x, const Words8 *w) {
  if (!nb || !cnt) return;
  hipLaunchKernelGGL(k_sc_scale_rows, dim3((cnt + 255) / 256, nb), dim3(256), 0, st, cnt, x, w);
}
__global__ void __launch_bounds__(256) k_sc_weighted_colsum(size_t nb, size_t cnt, const Words8 *x, const Words8 *w, Words8 *out) {
  __shared__ int32_t sm[NL * 4];
  size_t i = blockIdx.x;
  Fn acc = fe_zero<FN>();
  int c = 0;
  for (size_t p = threadIdx.x; p < nb; p += 256) {
    acc = add(acc, mul(load_plain(&x[p * cnt + i]), load_plain(&w[p])));
    if ((++c & 15) == 0) acc = fn_reduce(acc);
  }
  acc = wave_sum(fn_reduce(acc));
  if ((threadIdx.x & 63) == 0) raw_put(sm + (threadIdx.x >> 6) * NL, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    Fn t = raw_get(sm);
    for (int wv = 1; wv < 4; wv++) t = add(t, raw_get(sm + wv * NL));
    store_plain(&out[i], t);
  }
}
void sc_weighted_colsum(hipStream_t st, size_t nb, size_t cnt, const Words8 *x, const Words8 *w, Words8 *out) {
  if (!cnt) return;
  hipLaunchKernelGGL(k_sc_weighted_colsum, dim3(cnt), dim3(256), 0, st, nb, cnt, x, w, out);
}

// inner_product_proof.rs:280-309.  One block; lane 0 inverts (Montgomery trick over the k challenges);
// s_i = allinv * prod_{b : bit b of i} u_sq[(k-1)-b]  (closed form of the reference's induction :298-307)
__global__ void __launch_bounds__(256) k_verification_scalars(const Words8 *ch, int k, size_t n, Words8 *u_sq,
                                                             Words8 *u_inv_sq, Words8 *s) {
  __shared__ int32_t sm[(32 + 1) * NL];   // u_sq[k] (k < 32), allinv
  if (threadIdx.x == 0) {
    Fn pref[32], val[32];
    Fn acc = fe_one<FN>();
    for (int i = 0; i < k; i++) { pref[i] = acc; val[i] = load_plain(&ch[i]); acc = mul(acc, val[i]); }
    Fn ai = inv(acc);
    Fn allinv = ai;   // 1 / (u_1 ... u_k)
    for (int i = k - 1; i >= 0; i--) {
      Fn ui = mul(ai, pref[i]);
      ai = mul(ai, val[i]);
      Fn us = sqr(val[i]);
      store_plain(&u_sq[i], us);
      store_plain(&u_inv_sq[i], sqr(ui));
      raw_put(sm + i * NL, us);
    }
    raw_put(sm + 32 * NL, allinv);
  }
  __syncthreads();
  Fn allinv = raw_get(sm + 32 * NL);
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    Fn v = allinv;
    for (int b = 0; b < k; b++)
      if ((i >> b) & 1) v = mul(v, raw_get(sm + (k - 1 - b) * NL));
    store_plain(&s[i], v);
  }
}
void verification_scalars(hipStream_t st, const Words8 *challenges, size_t k, size_t n, Words8 *u_sq,
                          Words8 *u_inv_sq, Words8 *s) {
  hipLaunchKernelGGL(k_verification_scalars, dim3(1), dim3(256), 0, st, challenges, (int)k, n, u_sq, u_inv_sq, s);
}

// ------------------------------------------------------------------------------------------------
// ---- IPP over RESIDENT generators (no generator folding): after j rounds the folded generator G^(j)_t is
// sum over the original i == t (mod cur) of cG_i * G_i, so the round's L and R (inner_product_proof.rs:90-114,
// 159-172) are MSMs over the ORIGINAL generators with scalars a_.. * cG_i / b_.. * cH_i -- table lookups
// without a single doubling -- and the fold of G, H (:125-146, 202-248) becomes a scalar update of cG, cH.
// msc layout, compact (only the non-zero terms): [proof][L | R][1 + n0] = B scalar (c * w, Q = w * B), n0/2 G terms,
// n0/2 H terms.  With h = cur / 2 the j-th "hi" generator index is (j / h) cur + h + j % h, the j-th "lo" one
// (j / h) cur + j % h;  L uses G_hi and H_lo, R uses G_lo and H_hi (k_fixed_msm_ipp applies the same enumeration).
__global__ void __launch_bounds__(256) k_ipp_gens_scalars(size_t n0, size_t cur, const Words8 *a, const Words8 *b,
                                                          const Words8 *cG, const Words8 *cH, const Words8 *cLR,
                                                          const Words8 *w, Words8 *msc, size_t slo, size_t shi, int with_q) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= n0) return;
  const bool mine = i >= slo && i < shi;       // (a rank of a sharded proof keeps the generators [slo, shi): the others' terms are zero)
  const size_t h = cur / 2, t = i & (cur - 1), tl = t & (h - 1), per = 1 + n0, half = n0 / 2;
  const bool hi = t >= h;
  const size_t j = (i / cur) * h + tl;            // rank of i among the hi (or lo) indices
  const Words8 *ap = a + p * cur, *bp = b + p * cur;
  Words8 *L = msc + (p * 2) * per, *R = L + per;
  Fn cg = load_plain(&cG[p * n0 + i]), ch = load_plain(&cH[p * n0 + i]);
  if (!mine) cg = ch = fe_zero<FN>();
  if (hi) {
    store_plain(&L[1 + j], mul(load_plain(&ap[tl]), cg));            // <a_L, G_R>
    store_plain(&R[1 + half + j], mul(load_plain(&bp[tl]), ch));     // <b_L, H_R>
  } else {
    store_plain(&L[1 + half + j], mul(load_plain(&bp[h + tl]), ch)); // <b_R, H_L>
    store_plain(&R[1 + j], mul(load_plain(&ap[h + tl]), cg));        // <a_R, G_L>
  }
  if (i == 0) {
    Fn ww = with_q ? load_plain(&w[p]) : fe_zero<FN>();
    store_plain(&L[0], mul(load_plain(&cLR[2 * p]), ww));       // c_L * Q
    store_plain(&R[0], mul(load_plain(&cLR[2 * p + 1]), ww));   // c_R * Q
  }
}
void ipp_gens_scalars(hipStream_t st, size_t nb, size_t n0, size_t cur, const Words8 *a, const Words8 *b,
                      const Words8 *cG, const Words8 *cH, const Words8 *cLR, const Words8 *w, Words8 *msc, size_t slo, size_t shi, bool with_q) {
  if (!nb || !n0) return;
  hipLaunchKernelGGL(k_ipp_gens_scalars, dim3((n0 + 255) / 256, nb), dim3(256), 0, st, n0, cur, a, b, cG, cH, cLR, w, msc, slo, shi, with_q ? 1 : 0);
}
// G_factors / H_factors of the R1CS proof's inner-product argument (prover.rs:689-697): G_i factor 1 for the
// phase-1 multipliers and u for the rest, H_i factor y^-i times that
__global__ void __launch_bounds__(256) k_ipp_r1cs_factors(size_t np, size_t n1, const Words8 *u, const Words8 *y_inv,
                                                          Words8 *cG, Words8 *cH) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= np) return;
  Fn gf = i < n1 ? fe_one<FN>() : load_plain(&u[p]);
  store_plain(&cG[p * np + i], gf);
  store_plain(&cH[p * np + i], mul(fn_pow_u32(load_plain(&y_inv[p]), (uint32_t)i), gf));
}
void ipp_r1cs_factors(hipStream_t st, size_t nb, size_t np, size_t n1, const Words8 *u, const Words8 *y_inv, Words8 *cG,
                      Words8 *cH) {
  if (!nb || !np) return;
  hipLaunchKernelGGL(k_ipp_r1cs_factors, dim3((np + 255) / 256, nb), dim3(256), 0, st, np, n1, u, y_inv, cG, cH);
}
// G' = u^-1 G_L + u G_R ; H' = u H_L + u^-1 H_R as coefficient updates
__global__ void __launch_bounds__(256) k_ipp_gens_fold(size_t n0, size_t cur, const Words8 *u, const Words8 *u_inv,
                                                       Words8 *cG, Words8 *cH) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= n0) return;
  const bool hi = (i & (cur - 1)) >= cur / 2;
  Fn uu = load_plain(&u[p]), ui = load_plain(&u_inv[p]);
  store_plain(&cG[p * n0 + i], mul(load_plain(&cG[p * n0 + i]), hi ? uu : ui));
  store_plain(&cH[p * n0 + i], mul(load_plain(&cH[p * n0 + i]), hi ? ui : uu));
}
void ipp_gens_fold(hipStream_t st, size_t nb, size_t n0, size_t cur, const Words8 *u, const Words8 *u_inv, Words8 *cG,
                   Words8 *cH) {
  if (!nb || !n0) return;
  hipLaunchKernelGGL(k_ipp_gens_fold, dim3((n0 + 255) / 256, nb), dim3(256), 0, st, n0, cur, u, u_inv, cG, cH);
}

// flattened_constraints: zpow[b][r] = z_b^(r+1); output o = sum over its column of coeff * zpow[row]
// (w_V and w_c carry the reference's minus sign: prover.rs:367-369, verifier.rs:349-354)
__global__ void __launch_bounds__(256) k_zpow(const Words8 *z, size_t z_stride, size_t q, int32_t *zpow, size_t nchi, const Words8 *chi) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t b = blockIdx.y;
  if (r >= q) return;
  Fn zz = load_plain((const Words8 *)((const uint32_t *)z + b * z_stride));
  const Fn zr = fn_pow_u32(zz, (uint32_t)r + 1);
  const size_t qz = (1 + nchi) * q;
  raw_put(zpow + (b * qz + r) * NL, zr);
  for (size_t j = 0; j < nchi; j++) raw_put(zpow + (b * qz + (j + 1) * q + r) * NL, mul(zr, load_plain(&chi[b * nchi + j])));
}
__device__ __forceinline__ Fn flatten_column(const CircuitDev &c, size_t o, const int32_t *zp) {
  Fn acc = fe_zero<FN>();
  uint32_t cnt = 0;
  for (uint32_t t = c.col_ptr[o]; t < c.col_ptr[o + 1]; t++) {
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = c.coeff[t].w[j];
    acc = add(acc, mul(unpack<FN>(w), raw_get(zp + (size_t)c.row[t] * NL)));
    if ((++cnt & 15) == 0) acc = fn_reduce(acc);
  }
  if (o >= 3 * c.n) acc = neg(acc);
  return acc;
}
__global__ void __launch_bounds__(256) k_flatten(CircuitDev c, const int32_t *zpow, Words8 *wL, Words8 *wR, Words8 *wO,
                                                 Words8 *wV, Words8 *wc) {
  size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t b = blockIdx.y;
  size_t nout = 3 * c.n + c.m + 1;
  if (o >= nout) return;
  Fn v = flatten_column(c, o, zpow + b * c.qz * NL);
  if (o < c.n) store_plain(&wL[b * c.n + o], v);
  else if (o < 2 * c.n) store_plain(&wR[b * c.n + o - c.n], v);
  else if (o < 3 * c.n) store_plain(&wO[b * c.n + o - 2 * c.n], v);
  else if (o < 3 * c.n + c.m) store_plain(&wV[b * c.m + o - 3 * c.n], v);
  else if (wc) store_plain(&wc[b], v);
}
void flatten(hipStream_t st, const CircuitDev &c, size_t nb, const Words8 *z, size_t z_stride_words,
             Words8 *wL, Words8 *wR, Words8 *wO, Words8 *wV, Words8 *wc, int32_t *zpow, const Words8 *chi) {
  if (!nb) return;
  if (c.q) hipLaunchKernelGGL(k_zpow, dim3((c.q + 255) / 256, nb), dim3(256), 0, st, z, z_stride_words, c.q, zpow, c.nchi, chi);
  size_t nout = 3 * c.n + c.m + 1;
  hipLaunchKernelGGL(k_flatten, dim3((nout + 255) / 256, nb), dim3(256), 0, st, c, zpow, wL, wR, wO, wV, wc);
}

void zpow_table(hipStream_t st, size_t nb, size_t q, const Words8 *z, size_t z_stride_words, int32_t *zpow, size_t nchi, const Words8 *chi) {
  if (!nb || !q) return;
  hipLaunchKernelGGL(k_zpow, dim3((q + 255) / 256, nb), dim3(256), 0, st, z, z_stride_words, q, zpow, nchi, chi);
}

// ------------------------------------------------------------------------------------------------
// R1CS prover: l(x), r(x) coefficient vectors (prover.rs:596-617)
//   l1 = a_L + y^-i wR ; l2 = a_O ; l3 = s_L ; r0 = wO - y^i ; r1 = y^i a_R + wL ; r3 = y^i s_R
__global__ void __launch_bounds__(128) k_prover_polys(CircuitDev c, size_t nb, const Words8 *y, const Words8 *y_inv,
                                                      const Words8 *a_L, const Words8 *a_R, const Words8 *a_O,
                                                      const Words8 *s_L, const Words8 *s_R, const int32_t *zpow_all,
                                                      int32_t *polys, Words8 *wV_out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  const size_t n = c.n;
  const int32_t *zp = zpow_all + p * c.qz * NL;
  if (i < c.m && wV_out) store_plain(&wV_out[p * c.m + i], flatten_column(c, 3 * n + i, zp));
  if (i >= n) return;
  Fn yi = fn_pow_u32(load_plain(&y[p]), (uint32_t)i), yni = fn_pow_u32(load_plain(&y_inv[p]), (uint32_t)i);
  Fn wL = flatten_column(c, i, zp), wR = flatten_column(c, n + i, zp), wO = flatten_column(c, 2 * n + i, zp);
  size_t e = p * n + i, plane = nb * n * NL;
  int32_t *dst = polys + e * NL;
  raw_put(dst + 0 * plane, add(load_plain(&a_L[e]), mul(yni, wR)));
  raw_put(dst + 1 * plane, load_plain(&a_O[e]));
  raw_put(dst + 2 * plane, load_plain(&s_L[e]));
  raw_put(dst + 3 * plane, sub(wO, yi));
  raw_put(dst + 4 * plane, add(mul(yi, load_plain(&a_R[e])), wL));
  raw_put(dst + 5 * plane, mul(yi, load_plain(&s_R[e])));
}
void prover_polys(hipStream_t st, const CircuitDev &c, size_t nb, const Words8 *y, const Words8 *y_inv,
                  const Words8 *a_L, const Words8 *a_R, const Words8 *a_O, const Words8 *s_L, const Words8 *s_R,
                  const int32_t *zpow, int32_t *polys, Words8 *wV_out) {
  size_t span = c.n > c.m ? c.n : c.m;
  if (!nb || !span) return;
  hipLaunchKernelGGL(k_prover_polys, dim3((span + 127) / 128, nb), dim3(128), 0, st, c, nb, y, y_inv, a_L, a_R, a_O,
                     s_L, s_R, zpow, polys, wV_out);
}
// one block per (proof, coefficient): t1 = <l1,r0>; t2 = <l1,r1> + <l2,r0>; t3 = <l2,r1> + <l3,r0>;
// t4 = <l1,r3> + <l3,r1>; t5 = <l2,r3>; t6 = <l3,r3>     (planes: 0 l1, 1 l2, 2 l3, 3 r0, 4 r1, 5 r3)
__global__ void __launch_bounds__(256) k_prover_tcoeffs(size_t nb, size_t n, const int32_t *polys, Words8 *t_out) {
  __shared__ int32_t sm[NL * 4];
  const size_t p = blockIdx.x, plane = nb * n * NL;
  const int which = blockIdx.y;
  const int A1[6] = {0, 0, 1, 0, 1, 2}, B1[6] = {3, 4, 4, 5, 5, 5}, A2[6] = {-1, 1, 2, 2, -1, -1}, B2[6] = {-1, 3, 3, 4, -1, -1};
  const int a1 = A1[which], b1 = B1[which], a2 = A2[which], b2 = B2[which];
  Fn acc = fe_zero<FN>();
  int c = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    const int32_t *e = polys + (p * n + i) * NL;
    acc = add(acc, mul(raw_get(e + a1 * plane), raw_get(e + b1 * plane)));
    if (a2 >= 0) acc = add(acc, mul(raw_get(e + a2 * plane), raw_get(e + b2 * plane)));
    if ((++c & 7) == 0) acc = fn_reduce(acc);
  }
  acc = wave_sum(fn_reduce(acc));
  if ((threadIdx.x & 63) == 0) raw_put(sm + (threadIdx.x >> 6) * NL, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    Fn t = raw_get(sm);
    for (int w = 1; w < 4; w++) t = add(t, raw_get(sm + w * NL));
    store_plain(&t_out[p * 6 + which], t);
  }
}
void prover_tcoeffs(hipStream_t st, size_t nb, size_t n, const int32_t *polys, Words8 *t_out) {
  if (!nb) return;
  hipLaunchKernelGGL(k_prover_tcoeffs, dim3(nb, 6), dim3(256), 0, st, nb, n, polys, t_out);
}
// util.rs:172-181 VecPoly3::eval (l0 = 0, r2 = 0) and the padding of prover.rs:661-672
__global__ void __launch_bounds__(128) k_prover_eval(size_t nb, size_t n, size_t np, const Words8 *x, const Words8 *y,
                                                     const int32_t *polys, Words8 *l_vec, Words8 *r_vec) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y;
  if (i >= np) return;
  if (i < n) {
    Fn xx = load_plain(&x[p]);
    const size_t plane = nb * n * NL;
    const int32_t *e = polys + (p * n + i) * NL;
    Fn l = mul(xx, add(raw_get(e), mul(xx, add(raw_get(e + plane), mul(xx, raw_get(e + 2 * plane))))));
    Fn r = add(raw_get(e + 3 * plane), mul(xx, add(raw_get(e + 4 * plane), mul(xx, mul(xx, raw_get(e + 5 * plane))))));
    store_plain(&l_vec[p * np + i], l);
    store_plain(&r_vec[p * np + i], r);
  } else {
    store_plain(&l_vec[p * np + i], fe_zero<FN>());
    store_plain(&r_vec[p * np + i], neg(fn_pow_u32(load_plain(&y[p]), (uint32_t)i)));
  }
}
void prover_eval(hipStream_t st, size_t nb, size_t n, size_t padded_n, const Words8 *x, const Words8 *y,
                 const int32_t *polys, Words8 *l_vec, Words8 *r_vec) {
  if (!nb || !padded_n) return;
  hipLaunchKernelGGL(k_prover_eval, dim3((padded_n + 127) / 128, nb), dim3(128), 0, st, nb, n, padded_n, x, y, polys,
                     l_vec, r_vec);
}

// ------------------------------------------------------------------------------------------------
// Verifier scalar assembly (r1cs/verifier.rs:457-532) in two kernels.
// k_vs_prep, one LANE per proof: y^-1 and u_j^-1 (verifier.rs:468, inner_product_proof.rs:283) by Montgomery's
// trick around one binary-GCD inversion, then u_j^2, u_j^-2, prod u_j^-1.  (Run by one lane of a per-proof block
// this serial chain cost a full wave's issue slots per proof: a quarter of all instructions of a verification.)
// aux per proof (NL ints each): 0 y_inv, 1 allinv, 2.. u_sq[32], 34.. u_inv_sq[32] (+ partials, large path)
__global__ void __launch_bounds__(64) k_vs_prep(VsPrepArgs a) { vs_prep_body(a, blockIdx.x); }
// x^lane for the 64 lanes of a wave through two 8-entry tables in LDS: lanes 0..7 build x^0..x^7, lanes 8..15
// (x^8)^0..(x^8)^7 -- three conditional products on per-lane bases --, then every lane multiplies one entry of each.
// 4 products + 5 squares per wave instead of the 7 + 6 of per-lane square-and-multiply.  tab: 16 * NL ints.
__device__ __forceinline__ Fn wave64_powers(const Fn &x, int tid, int32_t *tab) {
  Fn x8 = sqr(sqr(sqr(x)));
  Fn b = tid < 8 ? x : x8, acc = fe_one<FN>();
  const int e = tid & 7;
#pragma unroll
  for (int bit = 0; bit < 3; bit++) {
    Fn t = mul(acc, b);
    if ((e >> bit) & 1) acc = t;
    if (bit < 2) b = sqr(b);
  }
  if (tid < 16) raw_put(tab + tid * NL, acc);
  __syncthreads();
  Fn r = mul(raw_get(tab + (tid & 7) * NL), raw_get(tab + (8 + (tid >> 3)) * NL));
  __syncthreads();   // tab may be reused
  return r;
}
// k_verify_scalars, one wave per proof: z powers, g_i / h_i, delta, w_c and the remaining scalars.
constexpr int VS_TPB = 64;
__global__ void __launch_bounds__(VS_TPB) k_verify_scalars(CircuitDev c, VerifyDims d, const Words8 *challenges,
                                                           const Words8 *proof_scalars, Words8 *fixed_sc,
                                                           Words8 *var_sc, Words8 *full_sc, int32_t *zpow_all,
                                                           const int32_t *aux_all, int *bad, int32_t *bad_proof) {
  __shared__ int32_t sm[(VS_AUX + 2) * NL];
  __shared__ int32_t tab[16 * NL], stab[64 * NL];   // two-level power tables / the s_i of the wave (np == 64 path)
  __builtin_amdgcn_s_setprio(2);   // latency-critical link of the per-batch chain (see k_vs_prep)
  // sm slots (NL ints each): 0 y_inv, 1 allinv, 2.. u_sq[k], 34.. u_inv_sq[k], 66 delta, 67 wc
  int32_t *s_usq = sm + 2 * NL, *s_uinvsq = sm + 34 * NL, *s_part = sm + VS_AUX * NL;
  const size_t p = blockIdx.x;
  const int tid = threadIdx.x;
  const size_t k = d.k, n = d.n, np = d.padded_n, m = d.m, n1 = d.n1;
  const Words8 *ch = challenges + p * (6 + k);
  const Words8 *ps = proof_scalars + p * 5;
  int32_t *zpow = zpow_all + p * c.qz * NL;
  Fn y = load_plain(&ch[0]), z = load_plain(&ch[1]), u = load_plain(&ch[2]), x = load_plain(&ch[3]);
  (void)y;
  for (int t = tid; t < VS_AUX * NL; t += VS_TPB) sm[t] = aux_all[p * VS_AUX * NL + t];
  if (bad || bad_proof) {   // canonical-encoding check of this proof's 6 + k challenges and 5 scalars (k_scalars_check)
    bool mine = false;
    for (size_t t = tid; t < 11 + k; t += VS_TPB) {
      const Words8 *src = t < 6 + k ? &ch[t] : &ps[t - 6 - k];
      uint32_t w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = src->w[j];
      if (!words_lt_mod<FN>(w)) mine = true;
    }
    const bool any = __any(mine);          // VS_TPB == 64: the block is one wave
    if (tid == 0) {
      if (any && bad) atomicOr(bad, 1);
      if (bad_proof) bad_proof[p] = any ? 1 : 0;   // the only writer of this proof's entry: no reset needed
    }
  }
  {   // z^(r+1) table (verifier.rs:336,358): lane r starts at z^(r+1) and steps by z^64
    Fn cur = mul(z, wave64_powers(z, tid, tab)), z64 = z;        // z^(tid + 1)
    for (int t = 0; t < 6; t++) z64 = sqr(z64);
    for (size_t r = tid; r < c.q; r += VS_TPB) {
      raw_put(zpow + r * NL, cur);
      for (size_t j = 0; j < c.nchi; j++) raw_put(zpow + ((j + 1) * c.q + r) * NL, mul(cur, load_plain(&d.chi[p * c.nchi + j])));
      cur = mul(cur, z64);
    }
  }
  __syncthreads();
  Fn y_inv = raw_get(sm + 0 * NL), allinv = raw_get(sm + 1 * NL);
  Fn a = load_plain(&ps[3]), b = load_plain(&ps[4]);
  const size_t nvar = 11 + m + 2 * k, nterms = 13 + m + 2 * np + 2 * k;
  Words8 *fx = fixed_sc + p * (2 + 2 * np);
  Words8 *vs = var_sc + p * nvar;
  Words8 *full = full_sc ? full_sc + p * nterms : nullptr;
  const size_t off_g = 13 + m, off_h = 13 + m + np;   // positions in verifier.rs:517-532 order

  Fn dpart = fe_zero<FN>();
  int dcnt = 0;
  // one wave = one pass when the padded size is the wave size (the 64-bit range gadget): y^-i and s_i come from
  // two-level tables shared through LDS, s_{63-i} is read back from the table of the s_i (16 of the ~100 wave-level
  // products of this kernel less)
  const bool wave_sized = np == (size_t)VS_TPB && k == 6;
  Fn yi_w = fe_zero<FN>(), si_w = fe_zero<FN>(), sr_w = fe_zero<FN>();
  if (wave_sized) {
    yi_w = wave64_powers(y_inv, tid, tab);
    // lanes 0..7: allinv * prod_{bit < 3} u_sq[5 - bit]^e_bit ; lanes 8..15: prod_{bit < 3} u_sq[2 - bit]^e_bit
    Fn acc = tid < 8 ? allinv : fe_one<FN>();
    const int e = tid & 7, hi = tid < 8 ? 0 : 3;
#pragma unroll
    for (int bit = 0; bit < 3; bit++) {
      Fn t = mul(acc, raw_get(s_usq + (5 - (hi + bit)) * NL));
      if ((e >> bit) & 1) acc = t;
    }
    if (tid < 16) raw_put(tab + tid * NL, acc);
    __syncthreads();
    si_w = mul(raw_get(tab + (tid & 7) * NL), raw_get(tab + (8 + (tid >> 3)) * NL));
    raw_put(stab + tid * NL, si_w);
    __syncthreads();
    sr_w = raw_get(stab + (VS_TPB - 1 - tid) * NL);
  }
  for (size_t i = tid; i < np; i += VS_TPB) {
    if ((++dcnt & 15) == 0) dpart = fn_reduce(dpart);
    Fn yi, si, sr;
    if (wave_sized) { yi = yi_w; si = si_w; sr = sr_w; }
    else {
      yi = fn_pow_u32(y_inv, (uint32_t)i);                   // y^-i (verifier.rs:469-471)
      // s_i and s_{np-1-i} (inner_product_proof.rs:298-307, closed form)
      si = allinv; sr = allinv;
      size_t ir = np - 1 - i;
      for (size_t bb = 0; bb < k; bb++) {
        Fn us = raw_get(s_usq + (k - 1 - bb) * NL);
        if ((i >> bb) & 1) si = mul(si, us);
        if ((ir >> bb) & 1) sr = mul(sr, us);
      }
    }
    Fn wLi = fe_zero<FN>(), wRi = fe_zero<FN>(), wOi = fe_zero<FN>();
    if (i < n) {
      wLi = flatten_column(c, i, zpow);
      wRi = flatten_column(c, n + i, zpow);
      wOi = flatten_column(c, 2 * n + i, zpow);
    }
    Fn yneg_wR = mul(wRi, yi);                                // verifier.rs:472-477
    dpart = add(dpart, mul(yneg_wR, wLi));                    // delta, verifier.rs:479
    Fn g = sub(mul(x, yneg_wR), mul(a, si));                  // verifier.rs:487-491
    Fn h = sub(mul(yi, sub(add(mul(x, wLi), wOi), mul(b, sr))), fe_one<FN>());   // verifier.rs:493-501
    if (i >= n1) { g = mul(u, g); h = mul(u, h); }
    store_plain(&fx[2 + i], g);
    store_plain(&fx[2 + np + i], h);
    if (full) { store_plain(&full[off_g + i], g); store_plain(&full[off_h + i], h); }
  }
  // w_c: the `One` column, its terms strided over the block (verifier.rs:352-354)
  Fn wcp = fe_zero<FN>();
  {
    const size_t o = 3 * n + m;
    int cnt = 0;
    for (uint32_t t = c.col_ptr[o] + tid; t < c.col_ptr[o + 1]; t += VS_TPB) {
      uint32_t w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = c.coeff[t].w[j];
      wcp = add(wcp, mul(unpack<FN>(w), raw_get(zpow + (size_t)c.row[t] * NL)));
      if ((++cnt & 15) == 0) wcp = fn_reduce(wcp);
    }
  }
  dpart = wave_sum(fn_reduce(dpart));
  wcp = wave_sum(fn_reduce(wcp));
  if (tid == 0) {
    raw_put(s_part, fn_reduce(dpart));
    raw_put(s_part + NL, fn_reduce(wcp));
  }
  __syncthreads();
  // one lane per remaining output scalar (verifier.rs:508-532)
  if ((size_t)tid < nvar + 2 || m > (size_t)VS_TPB) {
    Fn delta = raw_get(s_part), wc = raw_get(s_part + NL);
    wc = neg(wc);
    Fn r = load_plain(&ch[5]);
    Fn xx = sqr(x), rxx = mul(r, xx), xxx = mul(x, xx);
    for (size_t v = tid; v < nvar + 2; v += VS_TPB) {
      Fn val;
      size_t fpos = v;
      if (v < 3) val = v == 0 ? x : (v == 1 ? xx : xxx);                                  // A_I1 A_O1 S1
      else if (v < 6) val = mul(u, v == 3 ? x : (v == 4 ? xx : xxx));                    // A_I2 A_O2 S2
      else if (v < 6 + m) val = mul(flatten_column(c, 3 * n + (v - 6), zpow), rxx);      // V_j: wV_j r x^2
      else if (v < 11 + m) {                                                              // T_1 T_3 T_4 T_5 T_6
        size_t ti = v - 6 - m;
        val = ti == 0 ? mul(r, x) : ti == 1 ? mul(rxx, x) : ti == 2 ? mul(rxx, xx) : ti == 3 ? mul(rxx, xxx) : mul(mul(rxx, xx), xx);
      } else if (v < 11 + m + k) { val = raw_get(s_usq + (v - 11 - m) * NL); fpos = 13 + m + 2 * np + (v - 11 - m); }           // L_j: u_j^2
      else if (v < nvar) { val = raw_get(s_uinvsq + (v - 11 - m - k) * NL); fpos = 13 + m + 2 * np + k + (v - 11 - m - k); }     // R_j: u_j^-2
      else if (v == nvar) {   // B: w (t_x - a b) + r (xx (wc + delta) - t_x)
        Fn w_ch = load_plain(&ch[4]), t_x = load_plain(&ps[0]);
        val = add(mul(w_ch, sub(t_x, mul(a, b))), mul(r, sub(mul(xx, add(wc, delta)), t_x)));
        fpos = 11 + m;
      } else {                // B_blinding: -e_blinding - r t_x_blinding
        val = neg(add(load_plain(&ps[2]), mul(r, load_plain(&ps[1]))));
        fpos = 12 + m;
      }
      if (v < nvar) store_plain(&vs[v], val);
      else store_plain(&fx[v - nvar], val);
      if (full) store_plain(&full[fpos], val);
    }
  }
}
// The same for WAVE-SIZED proofs (padded n = 64, k = 6) after the lane-per-proof pass has done everything serial (vs_prep.cuh,
// fast path): this kernel is left with ~25 wave-wide products per proof instead of ~85 -- one per table look-up (y^-i, s_i, z^(r+1)),
// the flattening, g_i / h_i, delta and w_c, V_j and B.  Values are kept in the domain (plain / Montgomery) that makes every output a
// plain value without a conversion product (see vs_prep.cuh).
__global__ void __launch_bounds__(VS_TPB) k_verify_scalars_fast(CircuitDev c, VerifyDims d, const Words8 *challenges,
                                                                const Words8 *proof_scalars, Words8 *fixed_sc, Words8 *var_sc,
                                                                Words8 *full_sc, int32_t *zpow_all, const int32_t *aux_all, int *bad,
                                                                int32_t *bad_proof) {
  __shared__ int32_t sm[VS_AUX * NL];
  __shared__ int32_t stab[64 * NL], s_part[2 * NL];
  __builtin_amdgcn_s_setprio(2);
  const size_t p = blockIdx.x;
  const int tid = threadIdx.x;
  const size_t k = 6, n = d.n, np = 64, m = d.m, n1 = d.n1;
  const Words8 *ch = challenges + p * (6 + k);
  const Words8 *ps = proof_scalars + p * 5;
  int32_t *zpow = zpow_all + p * c.qz * NL;
  for (int t = tid; t < VS_AUX * NL; t += VS_TPB) sm[t] = aux_all[p * VS_AUX * NL + t];
  if (bad || bad_proof) {   // canonical-encoding check of this proof's 6 + k challenges and 5 scalars
    bool mine = false;
    for (size_t t = tid; t < 11 + k; t += VS_TPB) {
      const Words8 *src = t < 6 + k ? &ch[t] : &ps[t - 6 - k];
      uint32_t w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = src->w[j];
      if (!words_lt_mod<FN>(w)) mine = true;
    }
    const bool any = __any(mine);
    if (tid == 0) {
      if (any && bad) atomicOr(bad, 1);
      if (bad_proof) bad_proof[p] = any ? 1 : 0;
    }
  }
  __syncthreads();
  {   // z^(r+1), plain (verifier.rs:336,358): one table product, then steps of z^64
    Fn cur = mul(raw_get(sm + (VSF_Z1 + (tid & 7)) * NL), raw_get(sm + (VSF_Z2 + (tid >> 3)) * NL));
    const Fn z64 = raw_get(sm + VSF_Z64 * NL);
    for (size_t r = tid; r < c.q; r += VS_TPB) {
      raw_put(zpow + r * NL, cur);
      if (r + VS_TPB < c.q) cur = mul(cur, z64);
    }
  }
  const Fn yi = mul(raw_get(sm + (VSF_Y1 + (tid & 7)) * NL), raw_get(sm + (VSF_Y2 + (tid >> 3)) * NL));     // M(y^-i)
  const Fn si = mul(raw_get(sm + (VSF_S1 + (tid & 7)) * NL), raw_get(sm + (VSF_S2 + (tid >> 3)) * NL));     // P(s_i)
  raw_put(stab + tid * NL, si);
  __syncthreads();           // the z-power table (global) and the s table (LDS) are complete
  const Fn sr = raw_get(stab + (VS_TPB - 1 - tid) * NL);
  const Fn xm = raw_get(sm + VSF_X * NL), um = raw_get(sm + VSF_U * NL), am = raw_get(sm + VSF_A * NL), bm = raw_get(sm + VSF_B * NL);
  const size_t nvar = 11 + m + 2 * k, nterms = 13 + m + 2 * np + 2 * k;
  Words8 *fx = fixed_sc + p * (2 + 2 * np);
  Words8 *vs = var_sc + p * nvar;
  Words8 *full = full_sc ? full_sc + p * nterms : nullptr;
  const size_t off_g = 13 + m, off_h = 13 + m + np, i = tid;
  Fn one_p = fe_zero<FN>();
  one_p.v[0] = 1;
  Fn wLi = fe_zero<FN>(), wRi = fe_zero<FN>(), wOi = fe_zero<FN>();
  if (i < n) {                                           // plain: Montgomery coefficients x plain z powers
    wLi = flatten_column(c, i, zpow);
    wRi = flatten_column(c, n + i, zpow);
    wOi = flatten_column(c, 2 * n + i, zpow);
  }
  const Fn yneg_wR = mul(wRi, yi);                                                       // P, verifier.rs:472-477
  Fn dpart = mul(yneg_wR, wLi);                                                          // delta_i / R (two plain factors)
  Fn g = sub(mul(xm, yneg_wR), mul(am, si));                                             // P, verifier.rs:487-491
  Fn h = sub(mul(yi, sub(add(mul(xm, wLi), wOi), mul(bm, sr))), one_p);                  // P, verifier.rs:493-501
  if (i >= n1) { g = mul(um, g); h = mul(um, h); }
  vs_store_p(&fx[2 + i], g);
  vs_store_p(&fx[2 + np + i], h);
  if (full) { vs_store_p(&full[off_g + i], g); vs_store_p(&full[off_h + i], h); }
  Fn wcp = fe_zero<FN>();                                                                // w_c (plain): the `One` column, verifier.rs:352-354
  {
    const size_t o = 3 * n + m;
    int cnt = 0;
    for (uint32_t t = c.col_ptr[o] + tid; t < c.col_ptr[o + 1]; t += VS_TPB) {
      uint32_t w[8];
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = c.coeff[t].w[j];
      wcp = add(wcp, mul(unpack<FN>(w), raw_get(zpow + (size_t)c.row[t] * NL)));
      if ((++cnt & 15) == 0) wcp = fn_reduce(wcp);
    }
  }
  dpart = wave_sum(fn_reduce(dpart));
  wcp = wave_sum(fn_reduce(wcp));
  if (tid == 0) { raw_put(s_part, fn_reduce(dpart)); raw_put(s_part + NL, fn_reduce(wcp)); }
  __syncthreads();
  // V_j (one lane each) and B (verifier.rs:508-532); everything else was written by the lane-per-proof pass
  const Fn c1m = raw_get(sm + VSF_C1 * NL);
  for (size_t v = tid; v < m + 1; v += VS_TPB) {
    Fn val;
    if (v < m) val = mul(c1m, flatten_column(c, 3 * n + v, zpow));                       // wV_j r x^2
    else {
      const Fn delta = mul(raw_get(s_part), fe_r2<FN>());                                // (delta / R) R^2 / R = delta, plain
      const Fn wc = neg(raw_get(s_part + NL));
      val = add(raw_get(sm + VSF_C0 * NL), mul(c1m, add(wc, delta)));
    }
    if (v < m) { vs_store_p(&vs[6 + v], val); if (full) vs_store_p(&full[6 + v], val); }
    else { vs_store_p(&fx[0], val); if (full) vs_store_p(&full[11 + m], val); }
  }
}
// ---- the same assembly for LARGE proofs, split over the grid (one proof of the 2^14-shuffle has padded_n = 2^15,
// q = 65 533, m = 32 768: a single block per proof would serialise ~10^5 field multiplications per lane).
// aux per proof (NL ints each): 0 y_inv, 1 allinv, 2.. u_sq[32], 34.. u_inv_sq[32], 66.. delta partials[VSL_PARTS],
// then w_c partials[VSL_PARTS]
constexpr int VSL_PARTS = 256, VSL_AUX = VS_AUX + 2 * VSL_PARTS, VSL_TPB = 128;
// g_i, h_i (verifier.rs:469-501) + per-block delta partial; blockIdx.y = proof; grid-stride over i
__global__ void __launch_bounds__(VSL_TPB) k_vsl_gh(CircuitDev c, VerifyDims d, const Words8 *challenges,
                                                    const Words8 *proof_scalars, Words8 *fixed_sc, Words8 *full_sc,
                                                    const int32_t *zpow_all, int32_t *aux_all) {
  __shared__ int32_t s_part[NL * (VSL_TPB / 64)];
  const size_t p = blockIdx.y;
  const int tid = threadIdx.x;
  const size_t k = d.k, n = d.n, np = d.padded_n, m = d.m, n1 = d.n1;
  const Words8 *ch = challenges + p * (6 + k);
  const Words8 *ps = proof_scalars + p * 5;
  const int32_t *zpow = zpow_all + p * c.qz * NL;
  int32_t *aux = aux_all + p * VSL_AUX * NL;
  const int32_t *s_usq = aux + 2 * NL;
  Fn u = load_plain(&ch[2]), x = load_plain(&ch[3]);
  Fn y_inv = raw_get(aux), allinv = raw_get(aux + NL);
  Fn a = load_plain(&ps[3]), b = load_plain(&ps[4]);
  const size_t nterms = 13 + m + 2 * np + 2 * k;
  Words8 *fx = fixed_sc + p * (2 + 2 * np);
  Words8 *full = full_sc ? full_sc + p * nterms : nullptr;
  const size_t off_g = 13 + m, off_h = 13 + m + np;
  Fn dpart = fe_zero<FN>();
  int dcnt = 0;
  for (size_t i = (size_t)blockIdx.x * VSL_TPB + tid; i < np; i += (size_t)gridDim.x * VSL_TPB) {
    if ((++dcnt & 15) == 0) dpart = fn_reduce(dpart);
    Fn yi = fn_pow_u32(y_inv, (uint32_t)i);
    Fn si = allinv, sr = allinv;
    size_t ir = np - 1 - i;
    for (size_t bb = 0; bb < k; bb++) {
      Fn us = raw_get(s_usq + (k - 1 - bb) * NL);
      if ((i >> bb) & 1) si = mul(si, us);
      if ((ir >> bb) & 1) sr = mul(sr, us);
    }
    Fn wLi = fe_zero<FN>(), wRi = fe_zero<FN>(), wOi = fe_zero<FN>();
    if (i < n) {
      wLi = flatten_column(c, i, zpow);
      wRi = flatten_column(c, n + i, zpow);
      wOi = flatten_column(c, 2 * n + i, zpow);
    }
    Fn yneg_wR = mul(wRi, yi);
    dpart = add(dpart, mul(yneg_wR, wLi));
    Fn g = sub(mul(x, yneg_wR), mul(a, si));
    Fn h = sub(mul(yi, sub(add(mul(x, wLi), wOi), mul(b, sr))), fe_one<FN>());
    if (i >= n1) { g = mul(u, g); h = mul(u, h); }
    store_plain(&fx[2 + i], g);
    store_plain(&fx[2 + np + i], h);
    if (full) { store_plain(&full[off_g + i], g); store_plain(&full[off_h + i], h); }
  }
  dpart = wave_sum(fn_reduce(dpart));
  if ((tid & 63) == 0) raw_put(s_part + (tid >> 6) * NL, fn_reduce(dpart));
  __syncthreads();
  if (tid == 0) {
    Fn t = raw_get(s_part);
    for (int w = 1; w < VSL_TPB / 64; w++) t = add(t, raw_get(s_part + w * NL));
    raw_put(aux + (66 + blockIdx.x) * NL, t);
  }
}
// partial sums of the `One` column (w_c before the sign, verifier.rs:352-354)
__global__ void __launch_bounds__(VSL_TPB) k_vsl_wc(CircuitDev c, VerifyDims d, const int32_t *zpow_all, int32_t *aux_all) {
  __shared__ int32_t s_part[NL * (VSL_TPB / 64)];
  const size_t p = blockIdx.y;
  const int tid = threadIdx.x;
  const int32_t *zpow = zpow_all + p * c.qz * NL;
  int32_t *aux = aux_all + p * VSL_AUX * NL;
  const size_t o = 3 * d.n + d.m;
  Fn wcp = fe_zero<FN>();
  int cnt = 0;
  for (size_t t = (size_t)c.col_ptr[o] + (size_t)blockIdx.x * VSL_TPB + tid; t < c.col_ptr[o + 1]; t += (size_t)gridDim.x * VSL_TPB) {
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = c.coeff[t].w[j];
    wcp = add(wcp, mul(unpack<FN>(w), raw_get(zpow + (size_t)c.row[t] * NL)));
    if ((++cnt & 15) == 0) wcp = fn_reduce(wcp);
  }
  wcp = wave_sum(fn_reduce(wcp));
  if ((tid & 63) == 0) raw_put(s_part + (tid >> 6) * NL, fn_reduce(wcp));
  __syncthreads();
  if (tid == 0) {
    Fn t = raw_get(s_part);
    for (int w = 1; w < VSL_TPB / 64; w++) t = add(t, raw_get(s_part + w * NL));
    raw_put(aux + (66 + VSL_PARTS + blockIdx.x) * NL, t);
  }
}
// the remaining 11 + m + 2k + 2 scalars (verifier.rs:508-532), one lane each
__global__ void __launch_bounds__(VSL_TPB) k_vsl_tail(CircuitDev c, VerifyDims d, const Words8 *challenges,
                                                      const Words8 *proof_scalars, Words8 *fixed_sc, Words8 *var_sc,
                                                      Words8 *full_sc, const int32_t *zpow_all, const int32_t *aux_all,
                                                      int gh_parts, int wc_parts) {
  const size_t p = blockIdx.y;
  const size_t k = d.k, n = d.n, np = d.padded_n, m = d.m;
  const size_t nvar = 11 + m + 2 * k, nterms = 13 + m + 2 * np + 2 * k;
  const size_t v = (size_t)blockIdx.x * VSL_TPB + threadIdx.x;
  if (v >= nvar + 2) return;
  const Words8 *ch = challenges + p * (6 + k);
  const Words8 *ps = proof_scalars + p * 5;
  const int32_t *zpow = zpow_all + p * c.qz * NL;
  const int32_t *aux = aux_all + p * VSL_AUX * NL;
  Words8 *fx = fixed_sc + p * (2 + 2 * np);
  Words8 *vs = var_sc + p * nvar;
  Words8 *full = full_sc ? full_sc + p * nterms : nullptr;
  Fn u = load_plain(&ch[2]), x = load_plain(&ch[3]), r = load_plain(&ch[5]);
  Fn xx = sqr(x), rxx = mul(r, xx), xxx = mul(x, xx);
  Fn val;
  size_t fpos = v;
  if (v < 3) val = v == 0 ? x : (v == 1 ? xx : xxx);
  else if (v < 6) val = mul(u, v == 3 ? x : (v == 4 ? xx : xxx));
  else if (v < 6 + m) val = mul(flatten_column(c, 3 * n + (v - 6), zpow), rxx);
  else if (v < 11 + m) {
    size_t ti = v - 6 - m;
    val = ti == 0 ? mul(r, x) : ti == 1 ? mul(rxx, x) : ti == 2 ? mul(rxx, xx) : ti == 3 ? mul(rxx, xxx) : mul(mul(rxx, xx), xx);
  } else if (v < 11 + m + k) { val = raw_get(aux + (2 + v - 11 - m) * NL); fpos = 13 + m + 2 * np + (v - 11 - m); }
  else if (v < nvar) { val = raw_get(aux + (34 + v - 11 - m - k) * NL); fpos = 13 + m + 2 * np + k + (v - 11 - m - k); }
  else if (v == nvar) {
    Fn delta = fe_zero<FN>(), wc = fe_zero<FN>();
    for (int i = 0; i < gh_parts; i++) { delta = add(delta, raw_get(aux + (66 + i) * NL)); if ((i & 7) == 7) delta = fn_reduce(delta); }
    for (int i = 0; i < wc_parts; i++) { wc = add(wc, raw_get(aux + (66 + VSL_PARTS + i) * NL)); if ((i & 7) == 7) wc = fn_reduce(wc); }
    delta = fn_reduce(delta);
    wc = neg(fn_reduce(wc));
    Fn w_ch = load_plain(&ch[4]), t_x = load_plain(&ps[0]);
    Fn a = load_plain(&ps[3]), b = load_plain(&ps[4]);
    val = add(mul(w_ch, sub(t_x, mul(a, b))), mul(r, sub(mul(xx, add(wc, delta)), t_x)));
    fpos = 11 + m;
  } else {
    val = neg(add(load_plain(&ps[2]), mul(r, load_plain(&ps[1]))));
    fpos = 12 + m;
  }
  if (v < nvar) store_plain(&vs[v], val);
  else store_plain(&fx[v - nvar], val);
  if (full) store_plain(&full[fpos], val);
}

bool verify_scalars_fast_shape(const CircuitDev &c, const VerifyDims &d) { return vs_fast_shape(d) && c.nchi == 0 && d.m <= 64; }
static bool vs_large(const CircuitDev &c, const VerifyDims &d) {
  const size_t thr = d.vs_large_min ? d.vs_large_min : 4096;
  return d.padded_n >= thr || d.m >= thr || c.q >= 4 * thr;
}
size_t verify_scalars_scratch_ints(const CircuitDev &c, const VerifyDims &d) {
  return d.nb * ((c.qz ? c.qz : 1) + (vs_large(c, d) ? VSL_AUX : VS_AUX)) * NL;
}
bool verify_scalars_aux(const CircuitDev &c, const VerifyDims &d, int32_t *zpow_scratch, int32_t **aux, size_t *aux_stride) {
  if (vs_large(c, d)) return false;
  *aux = zpow_scratch + d.nb * (c.qz ? c.qz : 1) * NL;
  *aux_stride = VS_AUX;
  return true;
}
void verify_scalars(hipStream_t st, const CircuitDev &c, const VerifyDims &d, const Words8 *challenges,
                    const Words8 *proof_scalars, Words8 *fixed_sc, Words8 *var_sc, Words8 *full_sc,
                    int32_t *zpow_scratch, int *bad, int32_t *bad_proof, bool prep_done, bool prep_fast) {
  if (!d.nb) return;
  int32_t *aux = zpow_scratch + d.nb * (c.qz ? c.qz : 1) * NL;
  if (!vs_large(c, d)) {
    // wave-sized proofs: the lane-per-proof pass also does the serial part of the assembly (vs_prep.cuh) -- when it runs here, or
    // when the caller's fused front launch has run it with the output arrays (prep_fast)
    const bool fast = vs_fast_shape(d) && c.nchi == 0 && d.m <= 64 && (!prep_done || prep_fast);
    if (!prep_done) {
      VsPrepArgs pa{d, challenges, aux, (size_t)VS_AUX};
      if (fast) { pa.proof_scalars = proof_scalars; pa.fixed_sc = fixed_sc; pa.var_sc = var_sc; pa.full_sc = full_sc; }
      hipLaunchKernelGGL(k_vs_prep, dim3((d.nb + 63) / 64), dim3(64), 0, st, pa);
    }
    if (fast) hipLaunchKernelGGL(k_verify_scalars_fast, dim3(d.nb), dim3(VS_TPB), 0, st, c, d, challenges, proof_scalars,
                                 fixed_sc, var_sc, full_sc, zpow_scratch, aux, bad, bad_proof);
    else hipLaunchKernelGGL(k_verify_scalars, dim3(d.nb), dim3(VS_TPB), 0, st, c, d, challenges, proof_scalars,
                            fixed_sc, var_sc, full_sc, zpow_scratch, aux, bad, bad_proof);
    return;
  }
  if (bad_proof) {
    (void)hipMemsetAsync(bad_proof, 0, d.nb * sizeof(int32_t), st);
    hipLaunchKernelGGL(k_scalars_check_proof, dim3((d.nb * (6 + d.k) + 255) / 256), dim3(256), 0, st, challenges, d.nb * (6 + d.k), 6 + d.k, bad, bad_proof);
    hipLaunchKernelGGL(k_scalars_check_proof, dim3((d.nb * 5 + 255) / 256), dim3(256), 0, st, proof_scalars, d.nb * 5, (size_t)5, bad, bad_proof);
  } else if (bad) { scalars_check(st, challenges, d.nb * (6 + d.k), bad); scalars_check(st, proof_scalars, d.nb * 5, bad); }
  auto parts = [](size_t work) { size_t b = (work + VSL_TPB - 1) / VSL_TPB; return (int)(b < 1 ? 1 : (b > VSL_PARTS ? VSL_PARTS : b)); };
  const size_t o = 3 * d.n + d.m;
  (void)o;
  // number of `One` terms is only known on the device (col_ptr); size its grid from the row count
  const int gh_parts = parts(d.padded_n), wc_parts = parts(c.q);
  hipLaunchKernelGGL(k_vs_prep, dim3((d.nb + 63) / 64), dim3(64), 0, st, VsPrepArgs{d, challenges, aux, (size_t)VSL_AUX});
  if (c.q) hipLaunchKernelGGL(k_zpow, dim3((c.q + 255) / 256, d.nb), dim3(256), 0, st, challenges + 1, (6 + d.k) * 8, c.q, zpow_scratch, c.nchi, d.chi);
  hipLaunchKernelGGL(k_vsl_gh, dim3(gh_parts, d.nb), dim3(VSL_TPB), 0, st, c, d, challenges, proof_scalars, fixed_sc,
                     full_sc, zpow_scratch, aux);
  hipLaunchKernelGGL(k_vsl_wc, dim3(wc_parts, d.nb), dim3(VSL_TPB), 0, st, c, d, zpow_scratch, aux);
  const size_t nvar = 11 + d.m + 2 * d.k;
  hipLaunchKernelGGL(k_vsl_tail, dim3((nvar + 2 + VSL_TPB - 1) / VSL_TPB, d.nb), dim3(VSL_TPB), 0, st, c, d, challenges,
                     proof_scalars, fixed_sc, var_sc, full_sc, zpow_scratch, aux, gh_parts, wc_parts);
}

}  // namespace bpk
