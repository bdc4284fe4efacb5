// k_codec.hip -- wire codec of curve points: 32-byte compressed <-> 64-byte affine.
// StarkPoint::to_bytes / from_bytes live in the absent crate mpc-stark; the encoding restated here is arkworks'
// compressed short-Weierstrass form (x little-endian, bit 7 of byte 31 = "y is the larger of (y, -y)", bit 6 =
// infinity) -- parity unpinned.  Call sites: r1cs/proof.rs:82-207,
// inner_product_proof.rs:379-455.
//
// Decompression needs a square root in F_p with p - 1 = 2^192 (2^59 + 17): plain Tonelli-Shanks would take
// O(192^2) squarings.  Here the discrete logarithm of a^t in the 2-Sylow subgroup <c> is found by a recursive
// Pohlig-Hellman over 24 eight-bit digits (halving the digit range each level: 480 squarings + 52 table
// multiplications + 24 hashed lookups), then sqrt(a) = a^((t+1)/2) * c^(-e/2): ~100 k instructions per point.
#include "ec_dev.cuh"
#include "fe29_sqrt.cuh"

using namespace bp;

namespace bpk {

// T[j][d] = c^(-d * 2^(8j)), raw canonical Montgomery limbs; hash: 65536 bytes, slot -> digit of an element of <c^(2^184)>
__global__ void __launch_bounds__(256) k_sqrt_tables(int32_t *T, uint8_t *hash) {
  const int d = threadIdx.x, j = blockIdx.x;
  Fp e = sqrt_table_entry(j, d);
#pragma unroll
  for (int t = 0; t < NL; t++) T[((size_t)j * 256 + d) * NL + t] = e.v[t];
  if (j == SQ_DIG - 1) hash[sqrt_hash(e)] = (uint8_t)((256 - d) & 255);   // c^(-d 2^184) = (c^(2^184))^(256 - d)
}
// in: n x 32 B compressed; out: n x 64 B affine boundary form (zeros = identity); ok[i] = 1 iff the encoding is valid
__global__ void __launch_bounds__(64) k_points_decompress(const Words8 *in, Words8 *out, int32_t *ok, size_t n,
                                                          const int32_t *T, const uint8_t *hash) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  if (!live) i = n - 1;
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = in[i].w[j];
  const uint32_t flags = w[7] >> 30;
  w[7] &= 0x3FFFFFFFu;
  uint32_t xy[16];
#pragma unroll
  for (int j = 0; j < 16; j++) xy[j] = 0;
  bool good = flags != 3 && words_lt_mod<FP>(w);
  const bool inf = flags == 1;
  Fp x = to_mont(unpack<FP>(w));
  constexpr int32_t CB[NL] = CURVE_B_MONT;
  Fp B;
#pragma unroll
  for (int j = 0; j < NL; j++) B.v[j] = CB[j];
  Fp rhs = add(add(mul(sqr(x), x), x), B);   // x^3 + a x + b, a = 1
  Fp y;
  bool is_sq = fp_sqrt(y, rhs, T, hash);     // every lane walks the same instruction stream
  if (good && !inf) {
    good = is_sq;
    uint32_t yw[8];
    pack(yw, from_mont(y));
    constexpr uint32_t HALF[8] = FP_HALF_W;
    uint32_t hw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) hw[j] = HALF[j];
    if (words_gt(yw, hw) != (flags == 2)) pack(yw, from_mont(neg(y)));
#pragma unroll
    for (int j = 0; j < 8; j++) { xy[j] = w[j]; xy[8 + j] = yw[j]; }
    // (0, sqrt(b)) is a legitimate point; the all-zero boundary form is reserved for the identity and cannot
    // collide with it because b is not zero
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < 8; j++) { out[2 * i].w[j] = good ? xy[j] : 0u; out[2 * i + 1].w[j] = good ? xy[8 + j] : 0u; }
    ok[i] = good ? 1 : 0;
  }
}
// in: n x 64 B affine boundary form (validated elsewhere); out: n x 32 B
__global__ void __launch_bounds__(256) k_points_compress(const Words8 *xy, Words8 *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x[8], y[8], o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { x[j] = xy[2 * i].w[j]; y[j] = xy[2 * i + 1].w[j]; o |= x[j] | y[j]; }
  constexpr uint32_t HALF[8] = FP_HALF_W;
  uint32_t hw[8];
#pragma unroll
  for (int j = 0; j < 8; j++) hw[j] = HALF[j];
  if (o == 0) x[7] = 0x40000000u;
  else if (words_gt(y, hw)) x[7] |= 0x80000000u;
#pragma unroll
  for (int j = 0; j < 8; j++) out[i].w[j] = x[j];
}

size_t sqrt_table_bytes() { return (size_t)SQ_DIG * 256 * NL * 4 + 65536; }
void sqrt_tables_build(hipStream_t st, void *tab) {
  hipMemsetAsync((uint8_t *)tab + (size_t)SQ_DIG * 256 * NL * 4, 0, 65536, st);
  hipLaunchKernelGGL(k_sqrt_tables, dim3(SQ_DIG), dim3(256), 0, st, (int32_t *)tab, (uint8_t *)tab + (size_t)SQ_DIG * 256 * NL * 4);
}
void points_decompress(hipStream_t st, const Words8 *in, Words8 *out_xy, int32_t *ok, size_t n, const void *tab) {
  if (!n) return;
  hipLaunchKernelGGL(k_points_decompress, dim3((n + 63) / 64), dim3(64), 0, st, in, out_xy, ok, n, (const int32_t *)tab,
                     (const uint8_t *)tab + (size_t)SQ_DIG * 256 * NL * 4);
}
void points_compress(hipStream_t st, const Words8 *xy, Words8 *out, size_t n) {
  if (!n) return;
  hipLaunchKernelGGL(k_points_compress, dim3((n + 255) / 256), dim3(256), 0, st, xy, out, n);
}

}  // namespace bpk
