// fe29_sqrt.cuh -- square roots in F_p for point decompression (k_codec.hip); also compiled for the CPU by
// tests/csrc/fe29_host_test.cpp.  p - 1 = 2^192 t with t = 2^59 + 17: c = 3^t generates the 2-Sylow subgroup.
// sqrt(a) = a^((t+1)/2) * c^(-e/2) where a^t = c^e; e is found by a recursive Pohlig-Hellman over 24 eight-bit
// digits (the digit range halves at each level: 480 squarings + 52 table multiplications + 24 hashed lookups)
// against the table T[j][d] = c^(-d 2^(8j)) and a collision-free 16-bit hash of the 256 elements of <c^(2^184)>.
#pragma once
#include "fe29.cuh"

namespace bp {

constexpr int SQ_DIG = 24;   // 192 bits of 2-adicity in 8-bit digits

BP_HD Fp sqrt_table_entry(int j, int d) {   // canonical limbs of c^(-d 2^(8j))
  Fp base;
  constexpr int32_t C[NL] = FP_SQRT_CINV_MONT;
  for (int t = 0; t < NL; t++) base.v[t] = C[t];
  for (int i = 0; i < 8 * j; i++) base = sqr(base);
  Fp acc = fe_one<FP>();
  for (int i = 7; i >= 0; i--) { acc = sqr(acc); if ((d >> i) & 1) acc = mul(acc, base); }
  return canon(acc);
}
BP_HD uint32_t sqrt_hash(const Fp &canonical) {
  return (((uint32_t)canonical.v[0] ^ ((uint32_t)canonical.v[1] << 3)) * FP_SQRT_HASH_K) >> 16;
}
BP_HD Fp sqrt_tab_get(const int32_t *T, int j, int d) {
  Fp r;
  const int32_t *s = T + ((size_t)j * 256 + d) * NL;
  for (int t = 0; t < NL; t++) r.v[t] = s[t];
  return r;
}
// b in the subgroup of order 2^(8 ND) -> its ND digits relative to the generator c^(2^(8 (24 - ND)))
template <int ND> BP_HD void sylow_dlog(Fp b, uint8_t *dig, const int32_t *T, const uint8_t *hash) {
  if constexpr (ND == 1) {
    dig[0] = hash[sqrt_hash(canon(b))];
  } else {
    constexpr int LO = ND / 2, HI = ND - LO;
    Fp bh = b;
#pragma unroll 1
    for (int i = 0; i < 8 * HI; i++) bh = sqr(bh);
    sylow_dlog<LO>(bh, dig, T, hash);
#pragma unroll 1
    for (int i = 0; i < LO; i++) b = mul(b, sqrt_tab_get(T, SQ_DIG - ND + i, dig[i]));
    sylow_dlog<HI>(b, dig + LO, T, hash);
  }
}
// false when a is a non-residue (out is then meaningless)
BP_HD bool fp_sqrt(Fp &out, const Fp &a, const int32_t *T, const uint8_t *hash) {
  if (is_zero_exact(a)) { out = fe_zero<FP>(); return true; }
  Fp a8 = sqr(sqr(sqr(a)));   // w = a^((t-1)/2), (t - 1) / 2 = 2^58 + 8
  Fp w = a8;
#pragma unroll 1
  for (int i = 0; i < 55; i++) w = sqr(w);
  w = mul(w, a8);
  Fp x0 = mul(a, w);          // a^((t+1)/2)
  Fp b = mul(x0, w);          // a^t, in the 2-Sylow subgroup
  uint8_t dig[SQ_DIG + 1];
  sylow_dlog<SQ_DIG>(b, dig, T, hash);
  dig[SQ_DIG] = 0;
  bool ok = (dig[0] & 1) == 0;   // odd exponent <=> non-residue
  Fp r = x0;
#pragma unroll 1
  for (int j = 0; j < SQ_DIG; j++) {   // * c^(-e/2), e/2 digit by digit
    int f = (dig[j] >> 1) | ((dig[j + 1] & 1) << 7);
    r = mul(r, sqrt_tab_get(T, j, f));
  }
  out = r;
  return ok && is_zero_exact(sub(sqr(r), a));
}
BP_HD bool words_gt(const uint32_t a[8], const uint32_t b[8]) {
  for (int j = 7; j >= 0; j--) { if (a[j] != b[j]) return a[j] > b[j]; }
  return false;
}

}  // namespace bp
