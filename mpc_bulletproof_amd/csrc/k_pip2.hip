// k_pip2.hip -- bucket-method MSM of ONE instance in six launches, and the combined batch check built on it.
//
// k_pip.hip's pipeline (batched instances, two-level LDS sort for 2^20 terms) costs ~17 launches per MSM; a
// verification service that pushes thousands of 2^14..2^15-term MSMs per second (the combined batch check of
// BASELINE.json configs[1]: sum_p rho_p * mega_check_p, 24 576 proof points per 1024-proof batch) is then bound by the
// command processor and by serial tails, not by arithmetic.  Here:
//   K1 digits    lane per term: optional multiplication by a per-group weight (rho_p), signed c-bit digits (+K recoding),
//                histogram with integer atomics; identity points are dropped
//   K2 scan      ONE block: bucket offsets, task offsets (a bucket list is cut into tasks of <= 16 entries) and task table
//   K3 scatter   counting-sort scatter (order inside a bucket is irrelevant: the sum commutes)
//   K4 accum     lane per task: gathers its points (64-byte rows, next one prefetched) and adds them up
//   K5 reduce    block per window: bucket totals, S_w = sum_d d * B_d by per-lane running sums, offset multiplication
//                and an LDS tree
//   K6 final     ONE quad (ec29_quad.cuh): Horner over the windows, 252 cooperative doublings; optional extra addend
//                and boundary output
// K1, K4 and K6 take riders for what else a caller has ready at that point (fused launches: see verify_combined2).
// Replaces StarkPoint::msm_iter for 2^10 <= n <= 2^18 single instances (call sites: r1cs/verifier.rs:516,
// r1cs/prover.rs:465-564, inner_product_proof.rs:90-172); parity: tests/test_gpu_parity.py::test_msm_pippenger_*.
#include <cstdlib>
#include "fixed_body.cuh"
#include "vs_prep.cuh"
#include "ec29_quad.cuh"
#include "ec29_row.cuh"

using namespace bp;

namespace bpk {

constexpr uint32_t P2_TASK = 16;
constexpr uint32_t P2_NONE = 0xFFFFFFFFu;
constexpr uint32_t P2_HEAVY = 24;     // a bucket with more task partials than this is summed by a block of its own (K4b)

struct Pip2 {
  int c, W, half, bits;       // window bits, windows, buckets per window = 2^bits
  uint32_t K[9];              // sum_w 2^(c-1) 2^(c w)
  size_t n, nbk, max_tasks;
  const AffDev *pts;          // device Montgomery affine; zeros = identity
  const uint32_t *scalars;    // n x 8 plain canonical words
  const uint32_t *rho;        // optional: term i is multiplied by rho[i / rho_div] (plain canonical words)
  size_t rho_div;
  int *bad;
  uint32_t *keys, *sorted, *counts, *offsets, *cursor, *toffsets, *task_bucket;
  uint32_t *heavy;            // [0] = number of heavy buckets, [1..] their ids (written by K2)
  size_t max_heavy;
  JacRaw *partial, *buckets, *win;
};

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
int pippenger2_window(size_t n) {      // minimise n W (bucket additions) + W 2^(c-1) * 6 (reduce), c in [7, 12]
  int best = 7;
  double bc = 1e300;
  for (int c = 7; c <= 12; c++) {
    double W = 252 / c + 1, cost = (double)n * W + W * (double)(1u << (c - 1)) * 6.0;
    if (cost < bc) { bc = cost; best = c; }
  }
  return best;
}
bool pippenger2_supported(size_t n) {
  return n >= 256 && n <= ((size_t)1 << 16);
}
static void p2_dims(size_t n, int c, size_t *W, size_t *nbk, size_t *mt) {
  *W = 252 / c + 1;
  *nbk = *W * ((size_t)1 << (c - 1));
  *mt = n * *W / P2_TASK + *nbk + 1;
}
size_t pippenger2_scratch_bytes(size_t n, int c) {
  size_t W, nbk, mt;
  p2_dims(n, c, &W, &nbk, &mt);
  return al(n * W * 4) * 2 + al((nbk + 1) * 4) * 4 + al(mt * 4) + al((mt / P2_HEAVY + 2) * 4) + al(mt * sizeof(JacRaw)) +
         al(nbk * sizeof(JacRaw)) + al(W * sizeof(JacRaw));
}
static Pip2 p2_plan(const AffDev *pts, const uint32_t *scalars, size_t n, int c, void *scratch, int *bad) {
  Pip2 p{};
  size_t W, nbk, mt;
  p2_dims(n, c, &W, &nbk, &mt);
  p.c = c; p.W = (int)W; p.half = 1 << (c - 1); p.bits = c - 1;
  for (int j = 0; j < 9; j++) p.K[j] = 0;
  for (int w = 0; w < p.W; w++) { int bit = c * w + c - 1; p.K[bit >> 5] |= 1u << (bit & 31); }
  p.n = n; p.nbk = nbk; p.max_tasks = mt; p.pts = pts; p.scalars = scalars; p.bad = bad;
  uint8_t *q = (uint8_t *)scratch;
  p.keys = (uint32_t *)q; q += al(n * W * 4);
  p.sorted = (uint32_t *)q; q += al(n * W * 4);
  p.counts = (uint32_t *)q; q += al((nbk + 1) * 4);
  p.offsets = (uint32_t *)q; q += al((nbk + 1) * 4);
  p.cursor = (uint32_t *)q; q += al((nbk + 1) * 4);
  p.toffsets = (uint32_t *)q; q += al((nbk + 1) * 4);
  p.task_bucket = (uint32_t *)q; q += al(mt * 4);
  p.heavy = (uint32_t *)q; q += al((mt / P2_HEAVY + 2) * 4);
  p.max_heavy = mt / P2_HEAVY + 1;
  p.partial = (JacRaw *)q; q += al(mt * sizeof(JacRaw));
  p.buckets = (JacRaw *)q; q += al(nbk * sizeof(JacRaw));
  p.win = (JacRaw *)q;
  return p;
}

__device__ __forceinline__ int p2_digit(const uint32_t sp[9], int c, int w) {
  const int bit = c * w, k = bit >> 5, sft = bit & 31;
  uint64_t two = (uint64_t)sp[k] | (k + 1 < 9 ? (uint64_t)sp[k + 1] << 32 : 0);
  return (int)((two >> sft) & ((1u << c) - 1)) - (1 << (c - 1));
}

// atomicAdd(&base[b], 1) for the lanes with `valid`, returning each lane's old value -- with the lanes of a wave that hit the
// SAME counter combined into one atomic (up to 4 distinct counters per wave, the rest individually).  Low-entropy digits
// (the top window holds 252 mod c bits; equal scalars) otherwise serialise thousands of atomics on one address:
// 0.11 ms each in K1 and K3 of a 2^14-term MSM.
__device__ __forceinline__ uint32_t p2_agg_inc(uint32_t *base, uint32_t b, bool valid) {
  const int lane = (int)(threadIdx.x & 63);
  uint32_t res = 0;
  uint64_t todo = __ballot(valid);
#pragma unroll 1
  for (int it = 0; it < 4 && todo; it++) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const uint32_t v = (uint32_t)__shfl((int)b, leader, 64);
    const uint64_t same = __ballot(valid && b == v) & todo;
    uint32_t old = 0;
    if (lane == leader) old = atomicAdd(&base[v], (uint32_t)__popcll(same));
    old = (uint32_t)__shfl((int)old, leader, 64);
    if ((same >> lane) & 1) res = old + (uint32_t)__popcll(same & ((1ull << lane) - 1));
    todo &= ~same;
  }
  if ((todo >> lane) & 1) res = atomicAdd(&base[b], 1u);
  return res;
}
// ---- K1: 256-thread blocks, lane per term
__device__ __forceinline__ void p2_digits_body(const Pip2 &p, size_t blk) {
  size_t i = blk * 256 + threadIdx.x;
  const bool live = i < p.n;            // whole waves stay in the loop below (wave-aggregated atomics)
  if (!live) i = p.n - 1;
  uint32_t any = 0;
#pragma unroll
  for (int t = 0; t < 16; t++) any |= p.pts[i].w[t];
  uint32_t s[8];
#pragma unroll
  for (int t = 0; t < 8; t++) s[t] = p.scalars[i * 8 + t];
  if (p.rho) {
    uint32_t r[8];
    const uint32_t *rp = p.rho + (i / p.rho_div) * 8;
#pragma unroll
    for (int t = 0; t < 8; t++) r[t] = rp[t];
    if (live && (!words_lt_mod<FN>(r) || !words_lt_mod<FN>(s))) atomicOr(p.bad, 1);
    Fn x = mul(to_mont(unpack<FN>(s)), to_mont(unpack<FN>(r)));
    pack(s, from_mont(x));
  }
  uint32_t sp[9];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    uint64_t t = (uint64_t)(j < 8 ? s[j] : 0u) + p.K[j] + carry;
    sp[j] = (uint32_t)t;
    carry = t >> 32;
  }
  for (int w = 0; w < p.W; w++) {
    const int d = p2_digit(sp, p.c, w);
    uint32_t key = P2_NONE;
    const bool valid = live && d != 0 && any != 0;
    const uint32_t b = (uint32_t)w * (uint32_t)p.half + (uint32_t)((d < 0 ? -d : d) - 1);
    if (valid) key = b | (d < 0 ? 0x80000000u : 0u);
    (void)p2_agg_inc(p.counts, valid ? b : 0u, valid);
    if (live) p.keys[(size_t)w * p.n + i] = key;
  }
}
// ---- K2: ONE block of 1024 threads
__global__ void __launch_bounds__(1024) k_p2_scan(Pip2 p) {
  __shared__ uint32_t sa[1024], sb[1024];
  const int tid = threadIdx.x;
  const size_t per = (p.nbk + 1023) / 1024, lo = (size_t)tid * per, hi = lo + per < p.nbk ? lo + per : p.nbk;
  uint32_t s = 0, ts = 0;
  for (size_t j = lo; j < hi; j++) { uint32_t c = p.counts[j]; s += c; ts += c ? (c + P2_TASK - 1) / P2_TASK : 1u; }
  sa[tid] = s; sb[tid] = ts;
  if (tid == 0) p.heavy[0] = 0;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t va = tid >= off ? sa[tid - off] : 0, vb = tid >= off ? sb[tid - off] : 0;
    __syncthreads();
    sa[tid] += va; sb[tid] += vb;
    __syncthreads();
  }
  uint32_t run = tid ? sa[tid - 1] : 0, trun = tid ? sb[tid - 1] : 0;
  for (size_t j = lo; j < hi; j++) {
    const uint32_t c = p.counts[j], nt = c ? (c + P2_TASK - 1) / P2_TASK : 1u;
    p.offsets[j] = run; p.cursor[j] = run; p.toffsets[j] = trun;
    for (uint32_t t = 0; t < nt; t++) p.task_bucket[trun + t] = (uint32_t)j;
    if (nt > P2_HEAVY) { const uint32_t slot = atomicAdd(&p.heavy[0], 1u); if (slot < p.max_heavy) p.heavy[1 + slot] = (uint32_t)j; }
    run += c; trun += nt;
  }
  if (tid == 1023) { p.offsets[p.nbk] = sa[1023]; p.toffsets[p.nbk] = sb[1023]; }
}
// ---- K3: grid (ceil(n / 256), W)
__global__ void __launch_bounds__(256) k_p2_scatter(Pip2 p) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint32_t key = i < p.n ? p.keys[(size_t)blockIdx.y * p.n + i] : P2_NONE;
  const bool valid = key != P2_NONE;
  const uint32_t pos = p2_agg_inc(p.cursor, valid ? (key & 0x7FFFFFFFu) : 0u, valid);
  if (valid) p.sorted[pos] = (uint32_t)i | (key & 0x80000000u);
}
// ---- K4: 64-thread blocks, lane per task
__device__ __forceinline__ void p2_accum_body(const Pip2 &p, size_t blk) {
  const size_t t = blk * 64 + threadIdx.x;
  if (t >= p.toffsets[p.nbk]) return;
  const uint32_t b = p.task_bucket[t], slice = (uint32_t)t - p.toffsets[b];
  const uint32_t lo = p.offsets[b] + slice * P2_TASK, end = p.offsets[b + 1], hi = lo + P2_TASK < end ? lo + P2_TASK : end;
  Jac acc = jac_inf();
  // Two dependent loads per entry (index, then a random 64-byte row) against a ~1 650-instruction addition: the index
  // of entry e + 2 and the row of entry e + 1 are requested before the addition of entry e starts (with the row of
  // e + 1 waiting on an index fetched in the same iteration the launch ran at 60 percent of the addition rate).
  uint32_t cur[16], vcur = 0, vnxt = 0;
  if (lo < hi) {
    vcur = p.sorted[lo];
    const AffDev *src = &p.pts[vcur & 0x7FFFFFFFu];
#pragma unroll
    for (int j = 0; j < 16; j++) cur[j] = src->w[j];
    if (lo + 1 < hi) vnxt = p.sorted[lo + 1];
  }
  for (uint32_t e = lo; e < hi; e++) {
    uint32_t nxt[16], vnn = 0;
    if (e + 1 < hi) {
      const AffDev *src = &p.pts[vnxt & 0x7FFFFFFFu];
#pragma unroll
      for (int j = 0; j < 16; j++) nxt[j] = src->w[j];
      if (e + 2 < hi) vnn = p.sorted[e + 2];
    }
    Aff q;
    q.x = unpack<FP>(cur);
    q.y = unpack<FP>(cur + 8);
    if (vcur & 0x80000000u) q.y = neg(q.y);
    acc = jac_madd_nzq(acc, q);
#pragma unroll
    for (int j = 0; j < 16; j++) cur[j] = nxt[j];
    vcur = vnxt;
    vnxt = vnn;
  }
  raw_store(&p.partial[t], acc);
}
// ---- K5: block per window, 256 threads; thread t owns the L = half / 256 buckets t L .. t L + L - 1 (L >= 1: half >= 256,
// smaller windows use fewer threads)
// ---- K4b: block per heavy bucket (the grid is an upper bound; excess blocks exit): strided sum of its task partials + LDS
// tree -> buckets[b].  Low-entropy digits put thousands of entries into ONE bucket (the top window holds 252 mod c bits:
// 2 - 500 distinct digits; equal or small scalars do it in every window).
constexpr int P2R_TPB = 256;
__global__ void __launch_bounds__(P2R_TPB) k_p2_heavy(Pip2 p) {
  __shared__ int32_t smem[27 * (P2R_TPB / 2)];
  const uint32_t nh = p.heavy[0] < p.max_heavy ? p.heavy[0] : (uint32_t)p.max_heavy;
  if (blockIdx.x >= nh) return;
  const uint32_t b = p.heavy[1 + blockIdx.x], lo = p.toffsets[b], hi = p.toffsets[b + 1];
  Jac acc = jac_inf();
  for (uint32_t t = lo + threadIdx.x; t < hi; t += P2R_TPB) acc = jac_add(acc, raw_load(&p.partial[t]));
  acc = block_sum<P2R_TPB>(acc, smem);
  if (threadIdx.x == 0) raw_store(&p.buckets[b], acc);
}
// ---- K5: block per window, 256 threads; thread t owns the L = half / 256 buckets t L .. t L + L - 1 (L >= 1: half >= 256,
// smaller windows use fewer threads)
__global__ void __launch_bounds__(P2R_TPB) k_p2_reduce(Pip2 p) {
  __shared__ int32_t smem[27 * (P2R_TPB / 2)];
  const int w = blockIdx.x, tid = threadIdx.x;
  const int L = p.half >= P2R_TPB ? p.half / P2R_TPB : 1;
  const bool active = tid * L < p.half;
  const size_t b0 = (size_t)w * p.half + (size_t)tid * L;
  const bool heavy_done = p.heavy[0] <= p.max_heavy;     // (more heavy buckets than K4b blocks: the owner lanes sum them)
  Jac run = jac_inf(), ws = jac_inf();
  if (active) {
    for (int j = L - 1; j >= 0; j--) {
      const uint32_t lo = p.toffsets[b0 + j], hi = p.toffsets[b0 + j + 1];
      Jac tot;
      if (hi - lo > P2_HEAVY && heavy_done) tot = raw_load(&p.buckets[b0 + j]);
      else {
        tot = raw_load(&p.partial[lo]);
        for (uint32_t t = lo + 1; t < hi; t++) tot = jac_add(tot, raw_load(&p.partial[t]));
      }
      run = jac_add(run, tot);
      ws = jac_add(ws, run);                    // ws = sum (j + 1) * B_j over the lane's buckets
    }
  }
  // global weight of the lane's bucket j is (j + 1) + tid * L: add (tid * L) * run
  const unsigned mulby = (unsigned)tid * (unsigned)L;
  Jac sm = jac_inf();
  for (int bit = p.bits - 1; bit >= 0; bit--) {
    sm = jac_dbl(sm);
    if ((mulby >> bit) & 1) sm = jac_add(sm, run);
  }
  Jac acc = block_sum<P2R_TPB>(jac_add(ws, sm), smem);
  if (tid == 0) raw_store(&p.win[w], acc);
}
// ---- K6: ONE quad.  out_raw and / or out_xy (boundary bytes; costs one inversion)
struct P2Final { const JacRaw *win; int W, c; const JacRaw *extra; int nextra; JacRaw *out_raw; Words8 *out_xy; };
__device__ __forceinline__ void p2_final_body(const P2Final &f) {
  // ONE wave, row form (ec29_row.cuh): the 252 doublings at ~270 instructions each instead of the quad's 675
  const RowK K = rowk_init();
  JacR acc = jacr_from_limbs(K, f.win[f.W - 1].v);
#pragma unroll 1
  for (int w = f.W - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < f.c; d++) acc = rdbl(K, acc);
    acc = radd(K, acc, jacr_addend_from_limbs(K, f.win[w].v));
  }
#pragma unroll 1
  for (int e = 0; e < f.nextra; e++) acc = radd(K, acc, jacr_addend_from_limbs(K, f.extra[e].v));
  Jac r = jacr_gather(K, acc);
  if (threadIdx.x == 0) {
    if (f.out_raw) raw_store(f.out_raw, r);
    if (f.out_xy) {
      if (!jac_is_inf(r) && is_zero_exact(r.Z)) r = jac_inf();
      uint32_t w[16];
      aff_to_boundary(w, jac_to_aff(r));   // Fermat: the one-lane binary GCD measured 150 us slower (its compare / shift steps are serial too)
#pragma unroll
      for (int j = 0; j < 8; j++) { f.out_xy[0].w[j] = w[j]; f.out_xy[1].w[j] = w[8 + j]; }
    }
  }
}
__global__ void __launch_bounds__(64) k_p2_final(P2Final f) { p2_final_body(f); }

// ---- stand-alone launches (bpgpu_msm* of one mid-size instance) ----------------------------------------------------
__global__ void __launch_bounds__(256) k_p2_digits(Pip2 p) { p2_digits_body(p, blockIdx.x); }
__global__ void __launch_bounds__(64) k_p2_accum(Pip2 p) { p2_accum_body(p, blockIdx.x); }
// front: [boundary points -> device affine, validated | scalars canonical? | histogram reset]
struct P2Front { const Words8 *xy; AffDev *pts; const Words8 *sc; size_t n; int *bad; uint32_t *counts; size_t ncounts; unsigned pb; };
__global__ void __launch_bounds__(256) k_p2_front(P2Front a) {
  if (blockIdx.x < a.pb) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 8; j++) { w[j] = a.xy[2 * i].w[j]; w[8 + j] = a.xy[2 * i + 1].w[j]; }
    Aff q;
    if (!aff_from_boundary(q, w)) { atomicOr(a.bad, 1); q.x = fe_zero<FP>(); q.y = fe_zero<FP>(); }
    aff_store(&a.pts[i], q);
    uint32_t s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = a.sc[i].w[j];
    if (!words_lt_mod<FN>(s)) atomicOr(a.bad, 1);
  } else {
    const size_t i = (size_t)(blockIdx.x - a.pb) * 256 + threadIdx.x;
    if (i < a.ncounts) a.counts[i] = 0;
  }
}
// out_xy (boundary bytes) = sum_i scalars[i] * points[i] from the ABI encodings in HBM: seven launches.
// pts_tmp: n AffDev of scratch for the converted points.
void pippenger2_boundary(hipStream_t st, const Words8 *points_xy, const Words8 *scalars, size_t n, int c, Words8 *out_xy,
                         AffDev *pts_tmp, void *scratch, int *bad) {
  Pip2 p = p2_plan(pts_tmp, (const uint32_t *)scalars, n, c, scratch, bad);
  P2Front fr{points_xy, pts_tmp, scalars, n, bad, p.counts, p.nbk + 1, (unsigned)((n + 255) / 256)};
  hipLaunchKernelGGL(k_p2_front, dim3(fr.pb + (unsigned)((p.nbk + 1 + 255) / 256)), dim3(256), 0, st, fr);
  hipLaunchKernelGGL(k_p2_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(k_p2_scan, dim3(1), dim3(1024), 0, st, p);
  hipLaunchKernelGGL(k_p2_scatter, dim3((unsigned)((n + 255) / 256), p.W), dim3(256), 0, st, p);
  hipLaunchKernelGGL(k_p2_accum, dim3((unsigned)((p.max_tasks + 63) / 64)), dim3(64), 0, st, p);
  hipLaunchKernelGGL(k_p2_heavy, dim3((unsigned)p.max_heavy), dim3(P2R_TPB), 0, st, p);
  hipLaunchKernelGGL(k_p2_reduce, dim3(p.W), dim3(P2R_TPB), 0, st, p);
  P2Final f{p.win, p.W, p.c, nullptr, 0, nullptr, out_xy};
  hipLaunchKernelGGL(k_p2_final, dim3(1), dim3(64), 0, st, f);
}
// out = sum_i scalars[i] * pts[i]; pts already validated / converted (points_from_boundary).  The counts are zeroed here.
void pippenger2(hipStream_t st, const AffDev *pts, const uint32_t *scalars, size_t n, int c, JacRaw *out, void *scratch, int *bad) {
  Pip2 p = p2_plan(pts, scalars, n, c, scratch, bad);
  (void)hipMemsetAsync(p.counts, 0, (p.nbk + 1) * 4, st);
  hipLaunchKernelGGL(k_p2_digits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  hipLaunchKernelGGL(k_p2_scan, dim3(1), dim3(1024), 0, st, p);
  hipLaunchKernelGGL(k_p2_scatter, dim3((unsigned)((n + 255) / 256), p.W), dim3(256), 0, st, p);
  hipLaunchKernelGGL(k_p2_accum, dim3((unsigned)((p.max_tasks + 63) / 64)), dim3(64), 0, st, p);
  hipLaunchKernelGGL(k_p2_heavy, dim3((unsigned)p.max_heavy), dim3(P2R_TPB), 0, st, p);
  hipLaunchKernelGGL(k_p2_reduce, dim3(p.W), dim3(P2R_TPB), 0, st, p);
  P2Final f{p.win, p.W, p.c, nullptr, 0, out, nullptr};
  hipLaunchKernelGGL(k_p2_final, dim3(1), dim3(64), 0, st, f);
}

// ---- the combined batch check in eight launches ---------------------------------------------------------------------
// sum_p rho_p * mega_check_p as one point (BASELINE.json configs[1] "single big MSM"; built from r1cs/verifier.rs:457-553):
//   1 front    [proof points: validate + convert | inversion pass of the scalar assembly | zero the histogram]
//   2 scalars  k_verify_scalars (k_scalar.hip): per-proof MSM scalars
//   3 K1       [digits of rho_p * s_{p,j} | generator scalars sum_p rho_p * s_{p,g} (one block per generator)]
//   4 K2 scan  5 K3 scatter
//   6 K4       [bucket accumulation | the ONE fixed-base MSM over the generators (64 lanes, table lookups)]
//   7 K5 reduce
//   8 K6       Horner over the windows + the fixed-base partial -> boundary bytes
struct CombFront { const Words8 *points; AffDev *pts; size_t npts; int *bad; VsPrepArgs prep; uint32_t *counts; size_t ncounts; unsigned pb, vb; };
__global__ void __launch_bounds__(256) k_comb_front(CombFront a) {
  if (blockIdx.x < a.pb) {                       // proof points
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.npts) return;
    uint32_t w[16];
#pragma unroll
    for (int j = 0; j < 8; j++) { w[j] = a.points[2 * i].w[j]; w[8 + j] = a.points[2 * i + 1].w[j]; }
    Aff q;
    if (!aff_from_boundary(q, w)) { atomicOr(a.bad, 1); q.x = fe_zero<FP>(); q.y = fe_zero<FP>(); }
    aff_store(&a.pts[i], q);
  } else if (blockIdx.x < a.pb + a.vb) {         // inversion pass, lane per proof
    vs_prep_lane(a.prep, (size_t)(blockIdx.x - a.pb) * 256 + threadIdx.x);
  } else {                                       // histogram reset
    const size_t i = (size_t)(blockIdx.x - a.pb - a.vb) * 256 + threadIdx.x;
    if (i < a.ncounts) a.counts[i] = 0;
  }
}
struct CombColsum { const Words8 *x, *w; size_t nb, cnt; Words8 *out; };
__device__ __forceinline__ void comb_colsum_body(const CombColsum &a, size_t i, int32_t *sm /* NL * 4 */) {
  Fn acc = fe_zero<FN>();
  int c = 0;
  for (size_t p = threadIdx.x; p < a.nb; p += 256) {
    acc = add(acc, mul(load_plain(&a.x[p * a.cnt + i]), load_plain(&a.w[p])));
    if ((++c & 15) == 0) acc = fn_reduce(acc);
  }
  acc = wave_sum(fn_reduce(acc));
  if ((threadIdx.x & 63) == 0) raw_put(sm + (threadIdx.x >> 6) * NL, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    Fn t = raw_get(sm);
    for (int wv = 1; wv < 4; wv++) t = add(t, raw_get(sm + wv * NL));
    store_plain(&a.out[i], t);
  }
}
__global__ void __launch_bounds__(256) k_comb_k1(Pip2 p, unsigned digit_blocks, CombColsum cs) {
  __shared__ int32_t sm[NL * 4];
  if (blockIdx.x < digit_blocks) p2_digits_body(p, blockIdx.x);
  else comb_colsum_body(cs, blockIdx.x - digit_blocks, sm);
}
template <int C>
__global__ void __launch_bounds__(64) k_comb_k4(Pip2 p, unsigned accum_blocks, FixedSmallArgs f) {
  if (blockIdx.x < accum_blocks) p2_accum_body(p, blockIdx.x);
  else fixed_small_body<C, 64>(f.table, f.n, f.cap, f.scalars, f.sc_stride, f.out, f.nb, blockIdx.x - accum_blocks);
}
size_t verify_combined2_scratch_bytes(size_t nb, size_t nvar, size_t nfix) {
  const size_t tot = nb * nvar;
  return al(tot * sizeof(AffDev)) + al(nfix * 32) + al(2 * sizeof(JacRaw)) + pippenger2_scratch_bytes(tot, pippenger2_window(tot));
}
bool verify_combined2_supported(size_t nb, size_t nvar, int c, size_t np) {
  const size_t total = (2 + 2 * np) * (252 / c + 1);
  return pippenger2_supported(nb * nvar) && total <= 65536 && (c == 8 || c == 16 || c == 20);
}
void verify_combined2(hipStream_t st, const CombinedArgs &a) {
  const size_t tot = a.d.nb * a.nvar, nfix = 2 + 2 * a.d.padded_n;
  const int cw = pippenger2_window(tot);
  uint8_t *q = (uint8_t *)a.scratch;
  AffDev *dpts = (AffDev *)q; q += al(tot * sizeof(AffDev));
  Words8 *dfsum = (Words8 *)q; q += al(nfix * 32);
  JacRaw *dfixed = (JacRaw *)q; q += al(2 * sizeof(JacRaw));
  Pip2 p = p2_plan(dpts, (const uint32_t *)a.var_sc, tot, cw, q, a.bad);
  p.rho = (const uint32_t *)a.rho; p.rho_div = a.nvar;
  // 1 front
  int32_t *aux = nullptr;
  size_t aux_stride = 0;
  const bool fuse_prep = verify_scalars_aux(a.circ, a.d, a.zpow_scratch, &aux, &aux_stride);
  const bool fast = fuse_prep && verify_scalars_fast_shape(a.circ, a.d);   // wave-sized proofs: the serial part of the assembly in the front launch's lanes
  VsPrepArgs prep{a.d, a.challenges, aux, aux_stride};
  if (fast) { prep.proof_scalars = a.proof_scalars; prep.fixed_sc = a.fixed_sc; prep.var_sc = a.var_sc; }
  CombFront cf{a.points, dpts, tot, a.bad, prep, p.counts, p.nbk + 1,
               (unsigned)((tot + 255) / 256), fuse_prep ? (unsigned)((a.d.nb + 255) / 256) : 0u};
  { ProfMark pm(a.prof, a.prof_ctx, 12, st);
    hipLaunchKernelGGL(k_comb_front, dim3(cf.pb + cf.vb + (unsigned)((p.nbk + 1 + 255) / 256)), dim3(256), 0, st, cf);
    // 2 scalars (canonicity of the challenges / proof scalars is checked inside; rho in K1)
    verify_scalars(st, a.circ, a.d, a.challenges, a.proof_scalars, a.fixed_sc, a.var_sc, nullptr, a.zpow_scratch, a.bad, nullptr, fuse_prep, fast);
    // 3 K1
    CombColsum cs{a.fixed_sc, a.rho, a.d.nb, nfix, dfsum};
    const unsigned db = (unsigned)((tot + 255) / 256);
    hipLaunchKernelGGL(k_comb_k1, dim3(db + (unsigned)nfix), dim3(256), 0, st, p, db, cs); }
  { ProfMark pm(a.prof, a.prof_ctx, 13, st);
    hipLaunchKernelGGL(k_p2_scan, dim3(1), dim3(1024), 0, st, p);
    hipLaunchKernelGGL(k_p2_scatter, dim3((unsigned)((tot + 255) / 256), p.W), dim3(256), 0, st, p);
    // 6 K4 | fixed
    FixedSmallArgs f{a.table, a.d.padded_n, a.cap, (const uint32_t *)dfsum, nfix * 8, dfixed, 1};
    const unsigned ab = (unsigned)((p.max_tasks + 63) / 64);
    if (a.c == 8) hipLaunchKernelGGL((k_comb_k4<8>), dim3(ab + 1), dim3(64), 0, st, p, ab, f);
    else if (a.c == 16) hipLaunchKernelGGL((k_comb_k4<16>), dim3(ab + 1), dim3(64), 0, st, p, ab, f);
    else hipLaunchKernelGGL((k_comb_k4<20>), dim3(ab + 1), dim3(64), 0, st, p, ab, f);
    hipLaunchKernelGGL(k_p2_heavy, dim3((unsigned)p.max_heavy), dim3(P2R_TPB), 0, st, p);
  hipLaunchKernelGGL(k_p2_reduce, dim3(p.W), dim3(P2R_TPB), 0, st, p); }
  { ProfMark pm(a.prof, a.prof_ctx, 15, st);
    P2Final fin{p.win, p.W, p.c, dfixed, 1, nullptr, a.partial_xy};
    hipLaunchKernelGGL(k_p2_final, dim3(1), dim3(64), 0, st, fin); }
}

}  // namespace bpk
