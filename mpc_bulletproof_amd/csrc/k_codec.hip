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

// R1CSProof::from_bytes (r1cs/proof.rs:128-207) for nb proofs of one length: one lane per (proof, point slot).
// Writes the compressed points in the operand order of bpgpu_r1cs_verify_batch
// (A_I1 A_O1 S1 A_I2 A_O2 S2 | V_0..V_{m-1} | T_1 T_3 T_4 T_5 T_6 | L_0.. | R_0..), the five scalars as canonical
// little-endian words (big-endian on the wire, read modulo n: from_be_bytes_mod_order), fmt_ok[p] = version byte as
// the length implies.
__global__ void __launch_bounds__(256) k_wire_unpack(const uint8_t *proofs, size_t proof_len, const uint8_t *commitments,
                                                     size_t nb, size_t m, size_t k, int two_phase, Words8 *comp,
                                                     Words8 *scalars, int32_t *fmt_ok) {
  const size_t nvar = 11 + m + 2 * k, per = nvar + 5;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * per) return;
  const size_t p = t / per, s = t - p * per;
  const uint8_t *pr = proofs + p * proof_len;
  const size_t head = two_phase ? 11 : 8;   // compressed points before the three scalars
  if (s == 0) fmt_ok[p] = pr[0] == (two_phase ? 1 : 0) ? 1 : 0;
  const uint8_t *body = pr + 1;
  if (s < nvar) {
    const uint8_t *src = nullptr;
    bool identity = false;
    if (s < 3) src = body + 32 * s;
    else if (s < 6) { if (two_phase) src = body + 32 * s; else identity = true; }
    else if (s < 6 + m) src = commitments + (p * m + (s - 6)) * 32;
    else if (s < 11 + m) src = body + 32 * ((two_phase ? 6 : 3) + (s - 6 - m));
    else if (s < 11 + m + k) src = body + 32 * (head + 3 + 2 * (s - 11 - m));            // L_i
    else src = body + 32 * (head + 3 + 2 * (s - 11 - m - k) + 1);                        // R_i
    uint32_t w[8];
    for (int j = 0; j < 8; j++) {
      w[j] = identity ? 0u : ((uint32_t)src[4 * j] | (uint32_t)src[4 * j + 1] << 8 | (uint32_t)src[4 * j + 2] << 16 | (uint32_t)src[4 * j + 3] << 24);
    }
    if (identity) w[7] = 0x40000000u;
    for (int j = 0; j < 8; j++) comp[p * nvar + s].w[j] = w[j];
  } else {
    const size_t q = s - nvar;   // t_x t_x_blinding e_blinding a b
    const uint8_t *src = q < 3 ? body + 32 * (head + q) : body + 32 * (head + 3 + 2 * k + (q - 3));
    uint32_t w[8];
    for (int j = 0; j < 8; j++) {   // big-endian bytes -> little-endian words
      const uint8_t *b = src + 28 - 4 * j;
      w[j] = (uint32_t)b[3] | (uint32_t)b[2] << 8 | (uint32_t)b[1] << 16 | (uint32_t)b[0] << 24;
    }
    // any 256-bit value modulo n: one Montgomery multiplication by R^2 brings it into range
    uint32_t o[8];
    pack(o, from_mont(to_mont(unpack<FN>(w))));
    for (int j = 0; j < 8; j++) scalars[p * 5 + q].w[j] = o[j];
  }
}
// ok[p] &= fmt_ok[p] & all(dec_ok[p][*])
__global__ void __launch_bounds__(256) k_wire_and_ok(int32_t *ok, const int32_t *fmt_ok, const int32_t *dec_ok, size_t nb, size_t nvar) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nb) return;
  int good = fmt_ok[p];
  for (size_t v = 0; v < nvar; v++) good &= dec_ok[p * nvar + v];
  if (!good) ok[p] = 0;
}
void wire_unpack(hipStream_t st, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments, size_t nb, size_t m,
                 size_t k, int two_phase, Words8 *comp, Words8 *scalars, int32_t *fmt_ok) {
  size_t tot = nb * (16 + m + 2 * k);
  if (!tot) return;
  hipLaunchKernelGGL(k_wire_unpack, dim3((tot + 255) / 256), dim3(256), 0, st, proofs, proof_len, commitments, nb, m, k,
                     two_phase, comp, scalars, fmt_ok);
}
void wire_and_ok(hipStream_t st, int32_t *ok, const int32_t *fmt_ok, const int32_t *dec_ok, size_t nb, size_t nvar) {
  if (!nb) return;
  hipLaunchKernelGGL(k_wire_and_ok, dim3((nb + 255) / 256), dim3(256), 0, st, ok, fmt_ok, dec_ok, nb, nvar);
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
