// k_pip.hip -- bucket-method (Pippenger) multi-scalar multiplication for large variable-base MSMs
// (prover commitments and IPP rounds over folded generators, the C4 98 347-term
// mega_check, combined batch checks).  Replaces StarkPoint::msm_iter for n >~ 1k
// (call sites: r1cs/verifier.rs:516, r1cs/prover.rs:465-564, inner_product_proof.rs:90-172).
//
// Pipeline (all on one stream):
//   1 k_pip_digits    lane per term: signed c-bit digits via the +K recoding (no serial carry),
//                     histogram of bucket ids with integer atomics
//   2 k_pip_scan      exclusive scan of the W * 2^(c-1) bucket counts (single block)
//   3 k_pip_scatter   counting-sort scatter (order inside a bucket is irrelevant: the sum commutes)
//   4 k_pip_bucket    lane per bucket: gather its points (64 B rows) and accumulate with mixed adds
//   5 k_pip_window    block per window: segmented running sums  S_w = sum_d d * B_d,  LDS tree sum
//   6 k_pip_final     Horner over the windows: sum_w 2^(c w) S_w (252 doublings, one lane per instance)
// Integer work only; step 5's doubling chain (<= 252 sequential doublings, ~1 ms) is the latency
// floor of any variable-base MSM on this machine.
#include "ec_dev.cuh"
#include "ec29_quad.cuh"
#include "ec29_row.cuh"

using namespace bp;

namespace bpk {

struct PipParams {
  int c, W, half;          // window bits, windows, buckets per window
  uint32_t K[9];           // sum_w 2^(c-1) 2^(c w)
};

__device__ __forceinline__ int pip_digit(const uint32_t sp[9], int c, int w) {
  const int bit = c * w, k = bit >> 5, sft = bit & 31;
  uint64_t two = (uint64_t)sp[k] | (k + 1 < 9 ? (uint64_t)sp[k + 1] << 32 : 0);
  return (int)((two >> sft) & ((1u << c) - 1)) - (1 << (c - 1));
}

// `ninst` independent instances of n terms each (term-major per instance); bucket ids carry the instance
__global__ void __launch_bounds__(256) k_pip_digits(PipParams pp, const uint32_t *scalars, size_t n, size_t ninst,
                                                    uint32_t *keys, uint32_t *counts) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * ninst) return;
  const size_t inst = i / n, r = i - inst * n;
  uint32_t sp[9];
  uint64_t carry = 0;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    uint64_t t = (uint64_t)(j < 8 ? scalars[i * 8 + j] : 0u) + pp.K[j] + carry;
    sp[j] = (uint32_t)t;
    carry = t >> 32;
  }
  for (int w = 0; w < pp.W; w++) {
    int d = pip_digit(sp, pp.c, w);
    uint32_t key = 0xFFFFFFFFu;
    if (d != 0) {
      uint32_t b = (uint32_t)((inst * pp.W + w) * pp.half) + (uint32_t)((d < 0 ? -d : d) - 1);
      key = b | (d < 0 ? 0x80000000u : 0u);
      if (counts) atomicAdd(&counts[b], 1u);
    }
    keys[((size_t)inst * pp.W + w) * n + r] = key;
  }
}
// exclusive scan of `nb` counts into offsets[nb + 1] (cursor = optional copy), three phases:
// per-block sums of 2048-element tiles, single-block scan of the tile sums, per-tile scan + offset
constexpr int SCAN_TILE = 2048;
__global__ void __launch_bounds__(256) k_pip_scan_tiles(const uint32_t *counts, size_t nb, uint32_t *tile_sum) {
  __shared__ uint32_t sm[4];
  size_t base = (size_t)blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
  for (int j = threadIdx.x; j < SCAN_TILE; j += 256) { size_t i = base + j; if (i < nb) s += counts[i]; }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ void __launch_bounds__(1024) k_pip_scan_top(uint32_t *tile_sum, size_t ntiles, uint32_t *total_out) {
  __shared__ uint32_t part[1024];
  const int tid = threadIdx.x;
  size_t per = (ntiles + 1023) / 1024, lo = tid * per, hi = lo + per < ntiles ? lo + per : ntiles;
  uint32_t s = 0;
  for (size_t j = lo; j < hi; j++) s += tile_sum[j];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0;
  for (size_t j = lo; j < hi; j++) { uint32_t c = tile_sum[j]; tile_sum[j] = run; run += c; }   // exclusive tile offsets
  if (tid == 1023) *total_out = part[1023];
}
__global__ void __launch_bounds__(256) k_pip_scan_apply(const uint32_t *counts, const uint32_t *tile_off, size_t nb,
                                                        uint32_t *offsets, uint32_t *cursor) {
  __shared__ uint32_t sm[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
  uint32_t v[8], s = 0;
  for (int j = 0; j < 8; j++) { v[j] = base + j < nb ? counts[base + j] : 0; s += v[j]; }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    uint32_t t = threadIdx.x >= (unsigned)off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = tile_off[blockIdx.x] + (threadIdx.x ? sm[threadIdx.x - 1] : 0);
  for (int j = 0; j < 8; j++) {
    if (base + j < nb) { offsets[base + j] = run; if (cursor) cursor[base + j] = run; }
    run += v[j];
  }
}
// the same with the scan of the tile sums done by every block for itself (each adds up the sums of the tiles before its own: a few
// thousand words from L2) -- two launches instead of three while the tile sums are few
__global__ void __launch_bounds__(256) k_pip_scan_apply2(const uint32_t *counts, const uint32_t *tile_sum, size_t nb, size_t ntiles,
                                                         uint32_t *offsets, uint32_t *cursor) {
  __shared__ uint32_t sm[256];
  __shared__ uint32_t before[4];
  uint32_t pre = 0;
  for (size_t t = threadIdx.x; t < blockIdx.x; t += 256) pre += tile_sum[t];
  for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off, 64);
  if ((threadIdx.x & 63) == 0) before[threadIdx.x >> 6] = pre;
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
  uint32_t v[8], s = 0;
  for (int j = 0; j < 8; j++) { v[j] = base + j < nb ? counts[base + j] : 0; s += v[j]; }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    uint32_t t = threadIdx.x >= (unsigned)off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  const uint32_t tile_off = before[0] + before[1] + before[2] + before[3];
  uint32_t run = tile_off + (threadIdx.x ? sm[threadIdx.x - 1] : 0);
  for (int j = 0; j < 8; j++) {
    if (base + j < nb) { offsets[base + j] = run; if (cursor) cursor[base + j] = run; }
    run += v[j];
  }
  if (blockIdx.x + 1 == ntiles && threadIdx.x == 255) offsets[nb] = tile_off + sm[255];   // the total
}
static void pip_scan(hipStream_t st, const uint32_t *counts, uint32_t *offsets, uint32_t *cursor, size_t nb, uint32_t *tile_tmp) {
  size_t ntiles = (nb + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(k_pip_scan_tiles, dim3(ntiles), dim3(256), 0, st, counts, nb, tile_tmp);
  if (ntiles <= 4096) {
    hipLaunchKernelGGL(k_pip_scan_apply2, dim3(ntiles), dim3(256), 0, st, counts, tile_tmp, nb, ntiles, offsets, cursor);
    return;
  }
  hipLaunchKernelGGL(k_pip_scan_top, dim3(1), dim3(1024), 0, st, tile_tmp, ntiles, offsets + nb);
  hipLaunchKernelGGL(k_pip_scan_apply, dim3(ntiles), dim3(256), 0, st, counts, tile_tmp, nb, offsets, cursor);
}
__global__ void __launch_bounds__(256) k_pip_scatter(PipParams pp, const uint32_t *keys, size_t n, size_t ninst,
                                                     uint32_t *cursor, uint32_t *sorted) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * ninst * (size_t)pp.W) return;
  uint32_t key = keys[t];
  if (key == 0xFFFFFFFFu) return;
  uint32_t b = key & 0x7FFFFFFFu;
  uint32_t pos = atomicAdd(&cursor[b], 1u);
  size_t inst = t / (n * (size_t)pp.W), r = t % n;
  sorted[pos] = (uint32_t)(inst * n + r) | (key & 0x80000000u);   // global term index
}
// ---- LDS-staged two-level counting sort (large MSMs) ------------------------------------------------------------------
// The scatter above issues one global atomic and one random 4-byte write per (term, window): 1.5 ms of a 5 ms 2^20-term
// MSM, plus 0.4 ms of histogram atomics in k_pip_digits.  Here the keys of a (instance, window) segment are first split
// into 256 coarse bins (top 8 bits of the bucket id) with per-tile histograms and ranks kept in LDS -- a tile writes
// ~8 consecutive entries per bin -- and every (segment, bin) is then counting-sorted by one block entirely in LDS
// (<= 128 fine buckets), which also yields the per-bucket counts.  Order inside a bucket is irrelevant (the sum commutes).
// Tile = 8 192 keys: a tile writes ~32 consecutive entries per bin (128-byte runs; with 2 048-key tiles the 32-byte runs made
// every store a partial line).
constexpr int RS_BINS = 256, RS_TPB = 256, RS_PER = 32, RS_TILE = RS_TPB * RS_PER;
// The TOP window of a 252-bit scalar holds 252 mod c bits: its digits fill only the first 2^(252 mod c) buckets, which with the
// common shift all land in the first coarse bins (one block then sorted a whole window's entries: 0.2 ms of a 2^17-term MSM).
// It gets its own, smaller shift, so that its entries spread over the 256 bins as every other window's do.
__global__ void __launch_bounds__(RS_TPB) k_pip_coarse_hist(const uint32_t *keys, size_t n, int W, int half, int shift, int shift_top,
                                                            size_t tiles, uint32_t *gh) {
  __shared__ uint32_t h[RS_BINS];
  const size_t seg = blockIdx.y, tile = blockIdx.x;
  h[threadIdx.x] = 0;
  __syncthreads();
  if ((int)(seg % (size_t)W) == W - 1) shift = shift_top;
  const uint32_t base = (uint32_t)(seg * (size_t)half);
#pragma unroll 8
  for (int j = 0; j < RS_PER; j++) {
    size_t r = tile * RS_TILE + (size_t)j * RS_TPB + threadIdx.x;
    if (r < n) {
      uint32_t key = keys[seg * n + r];
      if (key != 0xFFFFFFFFu) atomicAdd(&h[((key & 0x7FFFFFFFu) - base) >> shift], 1u);
    }
  }
  __syncthreads();
  gh[(seg * RS_BINS + threadIdx.x) * tiles + tile] = h[threadIdx.x];
}
// The tile's entries are first placed bin by bin in LDS (ranks from LDS atomics), then written out in that order: consecutive
// lanes store consecutive addresses of a bin's run, so a wave's store is two or three full 128-byte segments instead of 64
// scattered 4-byte writes (0.27 -> 0.1 ms at 2^20 terms).
__global__ void __launch_bounds__(RS_TPB) k_pip_coarse_scatter(const uint32_t *keys, size_t n, int W, int half, int shift, int shift_top,
                                                               size_t tiles, const uint32_t *goff, uint32_t *cval, uint8_t *cfine) {
  __shared__ uint32_t cnt[RS_BINS], loc[RS_BINS], gbase[RS_BINS];
  __shared__ uint32_t sval[RS_TILE];
  __shared__ uint8_t sfine[RS_TILE], sbin[RS_TILE];
  const size_t seg = blockIdx.y, tile = blockIdx.x;
  const int tid = threadIdx.x;
  cnt[tid] = 0;
  gbase[tid] = goff[(seg * RS_BINS + tid) * tiles + tile];
  __syncthreads();
  if ((int)(seg % (size_t)W) == W - 1) shift = shift_top;
  const uint32_t base = (uint32_t)(seg * (size_t)half), fmask = (1u << shift) - 1;
#pragma unroll 8
  for (int j = 0; j < RS_PER; j++) {
    size_t r = tile * RS_TILE + (size_t)j * RS_TPB + tid;
    if (r < n) {
      uint32_t key = keys[seg * n + r];
      if (key != 0xFFFFFFFFu) atomicAdd(&cnt[((key & 0x7FFFFFFFu) - base) >> shift], 1u);
    }
  }
  __syncthreads();
  // exclusive prefix of the 256 bin counts (Hillis-Steele over one value per thread)
  uint32_t incl = cnt[tid];
  loc[tid] = incl;
  __syncthreads();
#pragma unroll 1
  for (int off = 1; off < RS_BINS; off <<= 1) {
    uint32_t v = tid >= off ? loc[tid - off] : 0u;
    __syncthreads();
    incl += v;
    loc[tid] = incl;
    __syncthreads();
  }
  const uint32_t total = loc[RS_BINS - 1];
  __syncthreads();
  loc[tid] = incl - cnt[tid];
  cnt[tid] = 0;
  __syncthreads();
#pragma unroll 8
  for (int j = 0; j < RS_PER; j++) {
    size_t r = tile * RS_TILE + (size_t)j * RS_TPB + tid;
    if (r < n) {
      uint32_t key = keys[seg * n + r];
      if (key != 0xFFFFFFFFu) {
        uint32_t lb = (key & 0x7FFFFFFFu) - base, bin = lb >> shift;
        uint32_t slot = loc[bin] + atomicAdd(&cnt[bin], 1u);
        sval[slot] = (uint32_t)((seg / (size_t)W) * n + r) | (key & 0x80000000u);   // global term index | sign (as k_pip_scatter)
        sfine[slot] = (uint8_t)(lb & fmask);
        sbin[slot] = (uint8_t)bin;
      }
    }
  }
  __syncthreads();
  for (uint32_t i = tid; i < total; i += RS_TPB) {
    const uint32_t bin = sbin[i], d = gbase[bin] + (i - loc[bin]);
    cval[d] = sval[i];
    cfine[d] = sfine[i];
  }
}
// one block per (segment, bin): counts of its 2^shift buckets and the entries placed bucket by bucket -- in LDS when the bin's
// entries fit (then written out contiguously), straight in global memory otherwise (skewed scalars)
constexpr uint32_t FS_CAP = 6144;     // 31 KB of LDS per block: five blocks per CU (a uniform 2^20-term window puts ~4 100 entries in a bin)
__global__ void __launch_bounds__(RS_TPB) k_pip_fine_sort(const uint32_t *goff, size_t tiles, size_t nseg, int W, int half, int shift,
                                                          int shift_top, const uint32_t *cval, const uint8_t *cfine, uint32_t *counts,
                                                          uint32_t *sorted) {
  __shared__ uint32_t h[128], start[128];
  __shared__ uint32_t stage[FS_CAP];
  __shared__ uint32_t fwords[FS_CAP / 4 + 2];
  const size_t seg = blockIdx.y, bin = blockIdx.x;
  if ((int)(seg % (size_t)W) == W - 1) shift = shift_top;     // (the counts of the top window's unused buckets were zeroed by the caller)
  const int fb = 1 << shift, t = threadIdx.x;
  const size_t gi = (seg * RS_BINS + bin) * tiles;
  const uint32_t lo = goff[gi], hi = goff[gi + tiles];     // goff has nseg * RS_BINS * tiles + 1 entries
  const uint32_t cnt = hi - lo;
  const bool staged = cnt <= FS_CAP;
  if (t < 128) h[t] = 0;
  // the fine digits of the bin, fetched with word loads into LDS (a byte per lane per load kept the launch waiting on memory)
  const uint32_t a0 = lo & ~3u;
  const uint8_t *fbytes = (const uint8_t *)fwords + (lo - a0);
  if (staged) {
    const uint32_t nwords = (hi - a0 + 3) / 4;
    const uint32_t *src = (const uint32_t *)(cfine + a0);
    for (uint32_t w = t; w < nwords; w += RS_TPB) fwords[w] = src[w];
  }
  __syncthreads();
  if (staged) { for (uint32_t i = t; i < cnt; i += RS_TPB) atomicAdd(&h[fbytes[i]], 1u); }
  else { for (uint32_t e = lo + t; e < hi; e += RS_TPB) atomicAdd(&h[cfine[e]], 1u); }
  __syncthreads();
  {   // exclusive prefix of the <= 128 bucket counts
    uint32_t incl = t < 128 ? h[t] : 0u;
    if (t < 128) start[t] = incl;
    __syncthreads();
#pragma unroll 1
    for (int off = 1; off < 128; off <<= 1) {
      uint32_t v = (t < 128 && t >= off) ? start[t - off] : 0u;
      __syncthreads();
      incl += v;
      if (t < 128) start[t] = incl;
      __syncthreads();
    }
    if (t < 128) start[t] = incl - h[t];
    __syncthreads();
  }
  if (t < fb) counts[seg * (size_t)half + bin * (size_t)fb + t] = h[t];
  if (staged) {
#pragma unroll 4
    for (uint32_t i = t; i < cnt; i += RS_TPB) stage[atomicAdd(&start[fbytes[i]], 1u)] = cval[lo + i];
    __syncthreads();
    for (uint32_t i = t; i < cnt; i += RS_TPB) sorted[lo + i] = stage[i];
  } else {
    for (uint32_t e = lo + t; e < hi; e += RS_TPB) sorted[lo + atomicAdd(&start[cfine[e]], 1u)] = cval[e];
  }
}

// Load balance: a bucket list is cut into tasks of at most PIP_TASK entries (the partial top window
// has only 2^(252 mod c) non-empty buckets holding n / 2^(252 mod c) points each; equal or
// low-entropy scalars are worse).  tcount[b] = max(1, ceil(len / PIP_TASK)); its scan gives task ids.
// `task` is chosen per call: 16 entries for sparse buckets, 64 when the regular buckets hold a few dozen points each (2^20 terms at
// c = 16: 32 +- 6) -- then almost every bucket IS one task, its lane writes the bucket itself, and the merge pass (0.19 ms of a
// 2.1 ms MSM at 2^20 terms, a lane per bucket adding two or three partials) only sees the top window's heavy buckets.
constexpr uint32_t PIP_TASK = 16, PIP_TASK_MAX = 64;
__global__ void __launch_bounds__(256) k_pip_taskcount(const uint32_t *counts, uint32_t *tcount, size_t nb, uint32_t task) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  uint32_t c = counts[b];
  tcount[b] = c ? (c + task - 1) / task : 1;
}
__global__ void __launch_bounds__(256) k_pip_taskdesc(const uint32_t *toffsets, size_t nb, uint32_t *task_bucket) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  for (uint32_t t = toffsets[b]; t < toffsets[b + 1]; t++) task_bucket[t] = (uint32_t)b;
}
// the same table, lane per TASK (binary search for its bucket; every bucket owns at least one task, so toffsets is strictly
// increasing): when the top window's few buckets hold hundreds of tasks each, the lane-per-bucket loop above is a serial tail
// (76 us of a 2^17-term MSM)
__global__ void __launch_bounds__(256) k_pip_taskdesc_search(const uint32_t *toffsets, size_t nb, uint32_t *task_bucket) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= toffsets[nb]) return;
  uint32_t lo = 0, hi = (uint32_t)nb;                      // toffsets[lo] <= t < toffsets[hi]
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (toffsets[mid] <= t) lo = mid; else hi = mid;
  }
  task_bucket[t] = lo;
}
// Tasks of equal length side by side.  Bucket sizes scatter around n / 2^(c-1) (2^20 terms at c = 16: 32 +- 6), so in
// bucket order the 64 tasks of a wave held anything from 1 to 16 entries and the wave ran for the longest: a third of the
// lane-time of the biggest launch of a large MSM was idle.  A counting sort of the task ids by length (17 classes, longest
// first, block-local histograms in LDS) makes the waves uniform; partial[] stays indexed by task id, so the merge is unchanged.
constexpr int TL_TPB = 256, TL_CLASSES = PIP_TASK_MAX + 1, TL_CURSOR = 128;   // (cursor: offset of the scatter cursors behind the histogram, in words)
__device__ __forceinline__ uint32_t pip_task_len(uint32_t t, const uint32_t *offsets, const uint32_t *toffsets, const uint32_t *task_bucket, uint32_t task) {
  const uint32_t b = task_bucket[t], slice = t - toffsets[b];
  const uint32_t lo = offsets[b] + slice * task, end = offsets[b + 1];
  const uint32_t len = lo < end ? end - lo : 0;             // (an empty bucket still owns one, empty, task)
  return len < task ? len : task;
}
// a block walks a tile of TL_TILE tasks: 17 global atomics per 4 096 tasks (one block per 256 tasks spent 60 us of a
// 2^20-term MSM queueing on the 17 counters)
constexpr int TL_TILE = 4096;
__global__ void __launch_bounds__(TL_TPB) k_pip_tasklen_hist(const uint32_t *offsets, const uint32_t *toffsets, const uint32_t *task_bucket,
                                                             size_t nbk, uint32_t *hist, uint32_t task) {
  __shared__ uint32_t h[TL_CLASSES];
  if (threadIdx.x < TL_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const size_t ntasks = toffsets[nbk], t0 = (size_t)blockIdx.x * TL_TILE;
  for (size_t t = t0 + threadIdx.x; t < t0 + TL_TILE && t < ntasks; t += TL_TPB)
    atomicAdd(&h[pip_task_len((uint32_t)t, offsets, toffsets, task_bucket, task)], 1u);
  __syncthreads();
  if (threadIdx.x < TL_CLASSES && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// cursor: TL_CLASSES zeroed counters; perm[position] = task id, classes in descending length
__global__ void __launch_bounds__(TL_TPB) k_pip_task_scatter(const uint32_t *offsets, const uint32_t *toffsets, const uint32_t *task_bucket,
                                                             size_t nbk, const uint32_t *hist, uint32_t *cursor, uint32_t *perm, uint32_t task) {
  __shared__ uint32_t h[TL_CLASSES], base[TL_CLASSES];
  if (threadIdx.x < TL_CLASSES) h[threadIdx.x] = 0;
  __syncthreads();
  const size_t ntasks = toffsets[nbk], t0 = (size_t)blockIdx.x * TL_TILE;
  for (size_t t = t0 + threadIdx.x; t < t0 + TL_TILE && t < ntasks; t += TL_TPB)
    atomicAdd(&h[pip_task_len((uint32_t)t, offsets, toffsets, task_bucket, task)], 1u);
  __syncthreads();
  if (threadIdx.x < TL_CLASSES) {
    uint32_t before = 0;
    for (int k = TL_CLASSES - 1; k > (int)threadIdx.x; k--) before += hist[k];
    base[threadIdx.x] = before + (h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0u);
    h[threadIdx.x] = 0;                                     // second pass: ranks inside the block's reservation
  }
  __syncthreads();
  for (size_t t = t0 + threadIdx.x; t < t0 + TL_TILE && t < ntasks; t += TL_TPB) {
    const uint32_t len = pip_task_len((uint32_t)t, offsets, toffsets, task_bucket, task);
    perm[base[len] + atomicAdd(&h[len], 1u)] = (uint32_t)t;
  }
}
__global__ void __launch_bounds__(64) k_pip_bucket_bounded(const AffDev *pts, const uint32_t *offsets, const uint32_t *sorted,
                                                           const uint32_t *toffsets, const uint32_t *task_bucket, size_t nbk,
                                                           const uint32_t *perm, JacRaw *partial, uint32_t task, JacRaw *buckets) {
  const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= toffsets[nbk]) return;
  const size_t t = perm ? perm[id] : id;
  uint32_t b = task_bucket[t], slice = (uint32_t)t - toffsets[b];
  uint32_t lo = offsets[b] + slice * task, end = offsets[b + 1], hi = lo + task < end ? lo + task : end;
  Jac acc = jac_inf();
  // Two dependent loads per entry (index, then a random 64-byte row) against a ~1 650-instruction addition: the index
  // of entry e + 2 and the row of entry e + 1 are requested before the addition of entry e starts (with the row of
  // e + 1 waiting on an index fetched in the same iteration the launch ran at 60 percent of the addition rate).
  uint32_t cur[16], vcur = 0, vnxt = 0;
  if (lo < hi) {
    vcur = sorted[lo];
    const AffDev *src = &pts[vcur & 0x7FFFFFFFu];
#pragma unroll
    for (int j = 0; j < 16; j++) cur[j] = src->w[j];
    if (lo + 1 < hi) vnxt = sorted[lo + 1];
  }
  for (uint32_t e = lo; e < hi; e++) {
    uint32_t nxt[16], vnn = 0;
    if (e + 1 < hi) {
      const AffDev *src = &pts[vnxt & 0x7FFFFFFFu];
#pragma unroll
      for (int j = 0; j < 16; j++) nxt[j] = src->w[j];
      if (e + 2 < hi) vnn = sorted[e + 2];
    }
    Aff q;
    q.x = unpack<FP>(cur);
    q.y = unpack<FP>(cur + 8);
    if (vcur & 0x80000000u) q.y = neg(q.y);
    acc = jac_madd(acc, q);
#pragma unroll
    for (int j = 0; j < 16; j++) cur[j] = nxt[j];
    vcur = vnxt;
    vnxt = vnn;
  }
  // a bucket that is ONE task is finished here (k_pip_merge skips it)
  raw_store(toffsets[b + 1] - toffsets[b] == 1 ? &buckets[b] : &partial[t], acc);
}
// bucket = sum of its task partials.  Lane per bucket; a bucket with more than PIP_HEAVY partials (the partial top
// window, equal or low-entropy scalars such as the 0/1 bits of a range proof) is queued for k_pip_merge_heavy
// instead of being summed serially by one lane (1.9 ms of a 5 ms 2^17-term MSM before).
constexpr uint32_t PIP_HEAVY = 16;
__global__ void __launch_bounds__(64) k_pip_merge(const uint32_t *toffsets, const JacRaw *partial, size_t nbuckets,
                                                  JacRaw *buckets, uint32_t *heavy_list, uint32_t *heavy_count) {
  size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  uint32_t lo = toffsets[b], hi = toffsets[b + 1];
  if (hi - lo == 1) return;                               // written by its one task's lane (k_pip_bucket_bounded)
  if (hi - lo > PIP_HEAVY) { heavy_list[atomicAdd(heavy_count, 1u)] = (uint32_t)b; return; }
  Jac acc = raw_load(&partial[lo]);
  for (uint32_t t = lo + 1; t < hi; t++) acc = jac_add(acc, raw_load(&partial[t]));
  raw_store(&buckets[b], acc);
}
// block per queued bucket (the grid is the upper bound total_tasks / PIP_HEAVY; excess blocks exit)
constexpr int PH_TPB = 64;    // one wave per queued bucket (128 threads: 7 tree levels for ~32 partials and twice the waves: 172 us at 2^20 terms)
__global__ void __launch_bounds__(PH_TPB) k_pip_merge_heavy(const uint32_t *toffsets, const JacRaw *partial,
                                                            const uint32_t *heavy_list, const uint32_t *heavy_count,
                                                            JacRaw *buckets) {
  if (blockIdx.x >= *heavy_count) return;
  const uint32_t b = heavy_list[blockIdx.x];
  const uint32_t lo = toffsets[b], hi = toffsets[b + 1];
  Jac acc = jac_inf();
  for (uint32_t t = lo + threadIdx.x; t < hi; t += PH_TPB) acc = jac_add(acc, raw_load(&partial[t]));
  static_assert(PH_TPB == 64, "one wave: shuffle butterfly, no LDS round trips");
#pragma unroll 1
  for (int off = 32; off > 0; off >>= 1) {
    Jac q;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      q.X.v[t] = __shfl_xor(acc.X.v[t], off, 64);
      q.Y.v[t] = __shfl_xor(acc.Y.v[t], off, 64);
      q.Z.v[t] = __shfl_xor(acc.Z.v[t], off, 64);
    }
    acc = jac_add(acc, q);
  }
  if (threadIdx.x == 0) raw_store(&buckets[b], acc);
}
// S_w = sum_{d=1}^{half} d * B_d per (window, instance), split into `chunks` blocks of half / chunks buckets each
// (a single block per window walks 256 buckets per lane at c = 16: 3.8 ms of a 9 ms 2^20-term MSM); the chunk
// partials already carry their global weights and are summed by segmented_sum
constexpr int PW_TPB = 128;
__global__ void __launch_bounds__(PW_TPB) k_pip_window(PipParams pp, const JacRaw *buckets, JacRaw *win_out, int chunks) {
  __shared__ int32_t smem[27 * (PW_TPB / 2)];
  const int w = blockIdx.x, tid = threadIdx.x, ch = blockIdx.z;
  const size_t inst = blockIdx.y;
  const int per_chunk = pp.half / chunks;
  const int L = per_chunk / PW_TPB;               // buckets per lane (per_chunk >= PW_TPB)
  const JacRaw *B = buckets + ((size_t)inst * pp.W + w) * pp.half + (size_t)ch * per_chunk + (size_t)tid * L;
  Jac run = jac_inf(), ws = jac_inf();
  for (int j = L - 1; j >= 0; j--) {
    run = jac_add(run, raw_load(&B[j]));
    ws = jac_add(ws, run);                         // ws = sum (j + 1) * B_j over the segment
  }
  // global weight of local bucket j is (j + 1) + ch * per_chunk + tid * L: add that offset times `run`
  unsigned mulby = (unsigned)ch * (unsigned)per_chunk + (unsigned)tid * (unsigned)L;
  Jac sm = jac_inf();
  for (int bit = 15; bit >= 0; bit--) {
    sm = jac_dbl(sm);
    if ((mulby >> bit) & 1) sm = jac_add(sm, run);
  }
  Jac acc = block_sum<PW_TPB>(jac_add(ws, sm), smem);
  if (tid == 0) raw_store(&win_out[((size_t)inst * pp.W + w) * chunks + ch], acc);
}
// ---- the same weighted sum without any per-lane scalar multiplication (the default): two stages of additions only.
// Stage A, one WAVE per chunk of 64 L buckets: lane l sums its L buckets (run_l, ws_l as above); the weight of lane l's run inside
// the chunk is l L, and sum_l l run_l = sum_{l >= 1} Suf_l with Suf the inclusive suffix sums of run over the lanes (six
// shuffle-add steps); so the chunk contributes  W_ch = sum_l [ws_l + L (l >= 1 ? Suf_l : 0)]  (log2 L doublings per lane, one
// butterfly) and  R_ch = Suf_0.  Stage B, one block of 64 QUADS per window (ec29_quad.cuh; the chunks' offsets):
// S = sum_ch W_ch + per_chunk sum_{ch >= 1} SufR_ch  -- a suffix scan over the chunks, log2(per_chunk) doublings, one tree.
// 31 dependent lane operations + ~25 quad operations against 16 + 32 + 7 with the double-and-add offset: 0.40 -> 0.22 ms at 2^20 terms.
__device__ __forceinline__ Jac jac_shfl_down(const Jac &a, int off) {
  Jac r;
#pragma unroll
  for (int t = 0; t < NL; t++) {
    r.X.v[t] = __shfl_down(a.X.v[t], off, 64);
    r.Y.v[t] = __shfl_down(a.Y.v[t], off, 64);
    r.Z.v[t] = __shfl_down(a.Z.v[t], off, 64);
  }
  return r;
}
__global__ void __launch_bounds__(64) k_pip_window_a(PipParams pp, const JacRaw *buckets, JacRaw *part, int chunks) {
  const int w = blockIdx.x, l = threadIdx.x, ch = blockIdx.z;
  const size_t seg = (size_t)blockIdx.y * pp.W + w;
  const int per_chunk = pp.half / chunks, L = per_chunk / 64;      // L = 2^k >= 1
  const JacRaw *B = buckets + seg * pp.half + (size_t)ch * per_chunk + (size_t)l * L;
  Jac run = jac_inf(), ws = jac_inf();
#pragma unroll 1
  for (int j = L - 1; j >= 0; j--) {
    run = jac_add(run, raw_load(&B[j]));
    ws = jac_add(ws, run);
  }
  Jac suf = run;
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    Jac v = jac_shfl_down(suf, off);
    suf = jac_add(suf, jac_select(l + off < 64, v, jac_inf()));
  }
  Jac y = jac_select(l >= 1, suf, jac_inf());
#pragma unroll 1
  for (int i = 1; i < L; i <<= 1) y = jac_dbl(y);
  Jac v = jac_add(ws, y);
#pragma unroll 1
  for (int off = 32; off > 0; off >>= 1) {
    Jac q;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      q.X.v[t] = __shfl_xor(v.X.v[t], off, 64);
      q.Y.v[t] = __shfl_xor(v.Y.v[t], off, 64);
      q.Z.v[t] = __shfl_xor(v.Z.v[t], off, 64);
    }
    v = jac_add(v, q);
  }
  if (l == 0) {
    raw_store(&part[seg * 2 * chunks + ch], v);
    raw_store(&part[seg * 2 * chunks + chunks + ch], suf);
  }
}
__global__ void __launch_bounds__(256) k_pip_window_b(PipParams pp, const JacRaw *part, JacRaw *win, int chunks) {
  __shared__ int32_t sm[27 * 64];
  const size_t seg = blockIdx.x;
  const int role = threadIdx.x & 3, qd = threadIdx.x >> 2;
  const bool live = qd < chunks;
  const int per_chunk = pp.half / chunks;
  auto put = [&](int slot, const Jac &a) {
#pragma unroll
    for (int t = 0; t < NL; t++) { sm[t * 64 + slot] = a.X.v[t]; sm[(NL + t) * 64 + slot] = a.Y.v[t]; sm[(2 * NL + t) * 64 + slot] = a.Z.v[t]; }
  };
  auto get = [&](int slot) {
    Jac a;
#pragma unroll
    for (int t = 0; t < NL; t++) { a.X.v[t] = sm[t * 64 + slot]; a.Y.v[t] = sm[(NL + t) * 64 + slot]; a.Z.v[t] = sm[(2 * NL + t) * 64 + slot]; }
    return a;
  };
  const JacT Wq = jact_from_jac(live ? raw_load(&part[seg * 2 * chunks + qd]) : jac_inf());
  JacT suf = jact_from_jac(live ? raw_load(&part[seg * 2 * chunks + chunks + qd]) : jac_inf());
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {          // inclusive suffix sums of R over the chunks (Hillis-Steele through LDS)
    if (role == 0) put(qd, jact_to_jac(suf));
    __syncthreads();
    const Jac o = qd + off < 64 ? get(qd + off) : jac_inf();
    __syncthreads();
    suf = q4_add(suf, jact_from_jac(o), role);
  }
  JacT y = jact_select(qd >= 1, suf, jact_inf());
#pragma unroll 1
  for (int i = 1; i < per_chunk; i <<= 1) y = q4_dbl(y, role);
  JacT v = q4_add(Wq, y, role);
#pragma unroll 1
  for (int s2 = 32; s2 > 0; s2 >>= 1) {
    if (role == 0 && qd >= s2 && qd < 2 * s2) put(qd - s2, jact_to_jac(v));
    __syncthreads();
    if (qd < s2) v = q4_add(v, jact_from_jac(get(qd)), role);
    __syncthreads();
  }
  if (threadIdx.x == 0) raw_store(&win[seg], jact_to_jac(v));
}
static int pip_window_ab_chunks(int half) {
  int chunks = half / 256;                           // L = 4 buckets per lane ...
  if (chunks > 64) chunks = 64;                      // ... more when the window has more than 64 x 256 buckets
  return chunks < 1 ? 1 : chunks;
}
static int pip_window_chunks(int half, size_t ninst, int W) {
  int chunks = 1;
  while (half / (chunks * 2) >= PW_TPB * 4 && ninst * (size_t)W * (size_t)(chunks * 2) <= 8192) chunks *= 2;   // >= 4 buckets per lane
  return chunks;
}
// Horner over the windows, one QUAD per instance (ec29_quad.cuh: a doubling is 3 field multiplications deep instead of
// 9): sum_w 2^(c w) S_w with 252 doublings in all (per-window doubling would cost c W^2 / 2 of them).  One lane per
// instance took 0.75 - 0.9 ms however small the MSM: the latency floor of every bucket-method call.
__global__ void __launch_bounds__(64) k_pip_final(const JacRaw *win, int W, int c, size_t ninst, JacRaw *out, size_t out_stride) {
  const int role = threadIdx.x & 3;
  size_t inst = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 2;
  const bool live = inst < ninst;
  if (!live) inst = ninst - 1;          // whole quads stay active
  JacT acc = jact_from_jac(raw_load(&win[inst * W + W - 1]));
#pragma unroll 1
  for (int w = W - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < c; d++) acc = q4_dbl(acc, role);
    acc = q4_add(acc, jact_from_jac(raw_load(&win[inst * W + w])), role);
  }
  if (live && role == 0) raw_store(&out[inst * out_stride], jact_to_jac(acc));
}
// The same with a WAVE per instance (ec29_row.cuh: a doubling ~270 instructions deep instead of 675): for the few instances of
// an ordinary call this tail is a lone chain on an idle chip -- 0.15 ms instead of 0.37.
__global__ void __launch_bounds__(64) k_pip_final_row(const JacRaw *win, int W, int c, JacRaw *out, size_t out_stride) {
  const RowK K = rowk_init();
  const size_t inst = blockIdx.x;
  JacR acc = jacr_from_limbs(K, win[inst * W + W - 1].v);
#pragma unroll 1
  for (int w = W - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < c; d++) acc = rdbl(K, acc);
    acc = radd(K, acc, jacr_addend_from_limbs(K, win[inst * W + w].v));
  }
  jacr_store(K, out[inst * out_stride].v, acc);
}

// window choice: minimise  n * W (bucket adds) + W * 2^(c-1) * ~3 (running sums), c in [8, 16]
int pippenger_window(size_t n) {
  int best = 8;
  double bc = 1e300;
  for (int c = 8; c <= 16; c++) {
    double W = 252 / c + 1, cost = (double)n * W + W * (double)(1u << (c - 1)) * 3.0;
    if (cost < bc) { bc = cost; best = c; }
  }
  return best;
}
static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
// the two-level sort pays from ~2^15 terms per instance on (fixed cost: two more scans and launches) and needs
// >= 256 buckets per coarse split (below: the atomic scatter)
static bool pip_two_level(size_t ninst, size_t n, int c) {
  (void)ninst;
  return c >= 12 && c <= 16 && n >= ((size_t)1 << 15);
}
static size_t pip_max_tasks(size_t n, size_t W, size_t nbk) { return n * W / PIP_TASK + nbk + 1; }
size_t pippenger_scratch_bytes_batch(size_t ninst, size_t n, int c) {
  size_t W = 252 / c + 1, half = (size_t)1 << (c - 1), nbk = ninst * W * half, tot = ninst * n, mt = pip_max_tasks(tot, W, nbk);
  size_t base = al(tot * W * 4) * 2 + al((nbk + 1) * 4) * 5 + al(mt * 4) + al(mt * sizeof(JacRaw)) + al(nbk * sizeof(JacRaw)) +
                al(ninst * W * sizeof(JacRaw)) * 129 + al(mt / PIP_HEAVY * 4 + 8) + al((nbk / SCAN_TILE + 2) * 4) + al(mt * 4) + al(1280);
  if (pip_two_level(ninst, n, c)) {
    size_t ngh = ninst * W * RS_BINS * ((n + RS_TILE - 1) / RS_TILE);
    base += al((ngh + 1) * 4) * 2 + al((ngh / SCAN_TILE + 2) * 4) + al(tot * W * 4) + al(tot * W);
  }
  return base;
}
size_t pippenger_scratch_bytes(size_t n, int c) { return pippenger_scratch_bytes_batch(1, n, c); }
// ninst instances of n terms: pts / scalars hold instance-major arrays; out[inst * out_stride]
void pippenger_batch(hipStream_t st, const AffDev *pts, const uint32_t *scalars, size_t ninst, size_t n, int c, JacRaw *out,
                     size_t out_stride, void *scratch) {
  if (!ninst) return;
  PipParams pp;
  pp.c = c; pp.W = 252 / c + 1; pp.half = 1 << (c - 1);
  for (int j = 0; j < 9; j++) pp.K[j] = 0;
  for (int w = 0; w < pp.W; w++) { int bit = c * w + c - 1; pp.K[bit >> 5] |= 1u << (bit & 31); }
  size_t W = pp.W, nbk = ninst * W * (size_t)pp.half, tot = ninst * n;
  uint8_t *p = (uint8_t *)scratch;
  uint32_t *keys = (uint32_t *)p; p += al(tot * W * 4);
  uint32_t *sorted = (uint32_t *)p; p += al(tot * W * 4);
  uint32_t *counts = (uint32_t *)p; p += al((nbk + 1) * 4);
  uint32_t *zblk = (uint32_t *)p; p += al(1280);          // right behind `counts`: cleared with it by ONE fill at the head of the stream of launches
  uint32_t *offsets = (uint32_t *)p; p += al((nbk + 1) * 4);
  uint32_t *cursor = (uint32_t *)p; p += al((nbk + 1) * 4);
  uint32_t *tcount = (uint32_t *)p; p += al((nbk + 1) * 4);
  uint32_t *toffsets = (uint32_t *)p; p += al((nbk + 1) * 4);
  const size_t mt = pip_max_tasks(tot, W, nbk);
  uint32_t *task_bucket = (uint32_t *)p; p += al(mt * 4);
  JacRaw *partial = (JacRaw *)p; p += al(mt * sizeof(JacRaw));
  JacRaw *buckets = (JacRaw *)p; p += al(nbk * sizeof(JacRaw));
  JacRaw *win = (JacRaw *)p; p += al(ninst * W * sizeof(JacRaw));
  JacRaw *win_part = (JacRaw *)p; p += al(ninst * W * sizeof(JacRaw)) * 128;    // <= 64 chunks per window, two points each
  uint32_t *heavy = (uint32_t *)p; p += al(mt / PIP_HEAVY * 4 + 8);              // [0] = count, [2..] = bucket ids
  uint32_t *tile_tmp = (uint32_t *)p; p += al((nbk / SCAN_TILE + 2) * 4);
  uint32_t *task_perm = (uint32_t *)p; p += al(mt * 4);
  uint32_t *tl_hist = zblk;                                                      // [0, TL_CLASSES) histogram, [TL_CURSOR, ..) cursors (1 024 bytes)
  uint32_t *heavy_cnt = zblk + 256;                                              // count of heavy buckets (heavy[2..] = their ids)
  (void)hipMemsetAsync(counts, 0, al((nbk + 1) * 4) + 1280, st);   // bucket counts (the top windows' buckets beyond 2^top_bits are written by nobody), task-length histogram, heavy count
  // entries per task: a few dozen points per regular bucket AND enough buckets to fill the chip with one lane each (4 waves per
  // SIMD) -> a bucket is one task, no merge pass for it.  (2^17 terms, 82 k buckets: long tasks leave one wave per SIMD -- 0.92
  // against 0.83 ms -- so the 16-entry tasks stay there.)
  const uint32_t task = n / (size_t)pp.half >= 12 && nbk >= ((size_t)1 << 18) ? PIP_TASK_MAX : PIP_TASK;
  if (pip_two_level(ninst, n, c)) {
    // LDS-staged two-level counting sort: digits (no atomics) -> coarse histograms per tile -> scan -> coarse scatter ->
    // per-(segment, bin) fine sort, which also produces the bucket counts
    const int shift = c - 1 - 8, top_bits = 252 - c * (pp.W - 1), shift_top = top_bits > 8 ? top_bits - 8 : 0;   // top digits are in [0, 2^top_bits]
    const size_t nseg = ninst * W, tiles = (n + RS_TILE - 1) / RS_TILE, ngh = nseg * RS_BINS * tiles;
    uint32_t *gh = (uint32_t *)p; p += al((ngh + 1) * 4);
    uint32_t *goff = (uint32_t *)p; p += al((ngh + 1) * 4);
    uint32_t *gtile = (uint32_t *)p; p += al((ngh / SCAN_TILE + 2) * 4);
    uint32_t *cval = (uint32_t *)p; p += al(tot * W * 4);
    uint8_t *cfine = p; p += al(tot * W);
    hipLaunchKernelGGL(k_pip_digits, dim3((tot + 255) / 256), dim3(256), 0, st, pp, scalars, n, ninst, keys, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_pip_coarse_hist, dim3(tiles, nseg), dim3(RS_TPB), 0, st, keys, n, pp.W, pp.half, shift, shift_top, tiles, gh);
    pip_scan(st, gh, goff, nullptr, ngh, gtile);
    hipLaunchKernelGGL(k_pip_coarse_scatter, dim3(tiles, nseg), dim3(RS_TPB), 0, st, keys, n, pp.W, pp.half, shift, shift_top, tiles, goff, cval, cfine);
    hipLaunchKernelGGL(k_pip_fine_sort, dim3(RS_BINS, nseg), dim3(RS_TPB), 0, st, goff, tiles, nseg, pp.W, pp.half, shift, shift_top, cval, cfine, counts, sorted);
    pip_scan(st, counts, offsets, nullptr, nbk, tile_tmp);
  } else {
    if (tot) hipLaunchKernelGGL(k_pip_digits, dim3((tot + 255) / 256), dim3(256), 0, st, pp, scalars, n, ninst, keys, counts);
    pip_scan(st, counts, offsets, cursor, nbk, tile_tmp);
    if (tot) hipLaunchKernelGGL(k_pip_scatter, dim3((tot * W + 255) / 256), dim3(256), 0, st, pp, keys, n, ninst, cursor, sorted);
  }
  hipLaunchKernelGGL(k_pip_taskcount, dim3((nbk + 255) / 256), dim3(256), 0, st, counts, tcount, nbk, task);
  pip_scan(st, tcount, toffsets, nullptr, nbk, tile_tmp);
  {
    const int top_bits = 252 - c * (pp.W - 1);
    const size_t tasks_per_top_bucket = (n >> top_bits) / task;     // uniform scalars: n / 2^top_bits entries per top bucket
    if (tasks_per_top_bucket > 64) hipLaunchKernelGGL(k_pip_taskdesc_search, dim3((mt + 255) / 256), dim3(256), 0, st, toffsets, nbk, task_bucket);
    else hipLaunchKernelGGL(k_pip_taskdesc, dim3((nbk + 255) / 256), dim3(256), 0, st, toffsets, nbk, task_bucket);
  }
  // the task count is data dependent: launch the upper bound, excess lanes exit on the device-side count
  const bool sort_tasks = tot * W >= ((size_t)1 << 18);     // two short launches: worth it from ~16 k tasks on
  if (sort_tasks) {
    hipLaunchKernelGGL(k_pip_tasklen_hist, dim3((mt + TL_TILE - 1) / TL_TILE), dim3(TL_TPB), 0, st, offsets, toffsets, task_bucket, nbk, tl_hist, task);
    hipLaunchKernelGGL(k_pip_task_scatter, dim3((mt + TL_TILE - 1) / TL_TILE), dim3(TL_TPB), 0, st, offsets, toffsets, task_bucket, nbk,
                       tl_hist, tl_hist + TL_CURSOR, task_perm, task);
  }
  hipLaunchKernelGGL(k_pip_bucket_bounded, dim3((mt + 63) / 64), dim3(64), 0, st, pts, offsets, sorted, toffsets, task_bucket,
                     nbk, sort_tasks ? task_perm : (const uint32_t *)nullptr, partial, task, buckets);
  hipLaunchKernelGGL(k_pip_merge, dim3((nbk + 63) / 64), dim3(64), 0, st, toffsets, partial, nbk, buckets, heavy + 2, heavy_cnt);
  hipLaunchKernelGGL(k_pip_merge_heavy, dim3(mt / PIP_HEAVY + 1), dim3(PH_TPB), 0, st, toffsets, partial, heavy + 2, heavy_cnt, buckets);
  if (pp.half >= 64) {
    const int chunks = pip_window_ab_chunks(pp.half);
    hipLaunchKernelGGL(k_pip_window_a, dim3(pp.W, ninst, chunks), dim3(64), 0, st, pp, buckets, win_part, chunks);
    hipLaunchKernelGGL(k_pip_window_b, dim3(ninst * W), dim3(256), 0, st, pp, win_part, win, chunks);
  } else {
    const int chunks = pip_window_chunks(pp.half, ninst, pp.W);
    if (chunks > 1) {
      hipLaunchKernelGGL(k_pip_window, dim3(pp.W, ninst, chunks), dim3(PW_TPB), 0, st, pp, buckets, win_part, chunks);
      segmented_sum(st, win_part, win, ninst * W, chunks);
    } else {
      hipLaunchKernelGGL(k_pip_window, dim3(pp.W, ninst, 1), dim3(PW_TPB), 0, st, pp, buckets, win, 1);
    }
  }
  if (ninst <= 1536) hipLaunchKernelGGL(k_pip_final_row, dim3((unsigned)ninst), dim3(64), 0, st, win, pp.W, pp.c, out, out_stride);
  else hipLaunchKernelGGL(k_pip_final, dim3((ninst * 4 + 63) / 64), dim3(64), 0, st, win, pp.W, pp.c, ninst, out, out_stride);
}
void pippenger(hipStream_t st, const AffDev *pts, const uint32_t *scalars, size_t n, int c, JacRaw *out, void *scratch) {
  pippenger_batch(st, pts, scalars, 1, n, c, out, 1, scratch);
}

}  // namespace bpk
