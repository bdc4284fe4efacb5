// CPU build (g++ -fsanitize=undefined) of the device arithmetic headers fe29.cuh / ec29.cuh, so the
// exact source the HIP kernels use is unit-tested (and overflow-checked by UBSan) without a GPU.
// Test infrastructure only -- never linked into the product.
#include <cstring>
#include "../../mpc_bulletproof_amd/csrc/ec29.cuh"
using namespace bp;

static void load_words(uint32_t w[8], const uint8_t *b) { memcpy(w, b, 32); }
template <class F> static bool load_fe(Fe<F> &out, const uint8_t *b) {
  uint32_t w[8];
  load_words(w, b);
  if (!words_lt_mod<F>(w)) return false;
  out = to_mont(unpack<F>(w));
  return true;
}
template <class F> static void store_fe(uint8_t *b, const Fe<F> &x) {
  uint32_t w[8];
  pack(w, from_mont(x));
  memcpy(b, w, 32);
}
template <class F> static int binop(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  Fe<F> x, y, r;
  if (!load_fe(x, a) || !load_fe(y, b)) return -1;
  switch (op) {
    case 0: r = add(x, y); break;
    case 1: r = sub(x, y); break;
    case 2: r = mul(x, y); break;
    case 3: r = sqr(x); break;
    case 4: r = inv_fermat(x); break;
    case 5: r = neg(x); break;
    case 6: r = mul_small<8>(x); break;
    // lazy chains: products of un-canonical operands
    case 7: r = mul(sub(x, y), add(x, y)); break;                  // x^2 - y^2
    case 8: r = sqr(sub(sub(x, y), y)); break;                     // (x - 2y)^2
    case 9: r = mul(norm(add_nr(add_nr(x, x), x)), sub(y, x)); break;  // 3x (y - x)
    case 10: r = inv_gcd(x); break;
    case 11: r = inv_ds(x); break;
    case 12: r = mul(inv_ds(sub(x, y)), sub(x, y)); break;        // lazy operand in, lazy result straight into a product: 1 (or 0)
    default: return -2;
  }
  store_fe(out, r);
  return 0;
}
static bool load_pt(Aff &p, const uint8_t *b) { uint32_t w[16]; memcpy(w, b, 64); return aff_from_boundary(p, w); }
static void store_pt(uint8_t *b, const Jac &j) { uint32_t w[16]; aff_to_boundary(w, jac_to_aff(j)); memcpy(b, w, 64); }

// scalar multiplication with signed fixed windows of C bits through recode_add_k / recode_digit
template <int C> static int smul(const uint8_t *s, const uint8_t *pt, uint8_t *out) {
  Aff p;
  uint32_t sw[8], sp[9];
  memcpy(sw, s, 32);
  if (!words_lt_mod<FN>(sw) || !load_pt(p, pt)) return -1;
  recode_add_k<C>(sp, sw);
  Jac tab[1 << (C - 1)];
  tab[0] = jac_from_aff(p);
  for (int i = 1; i < (1 << (C - 1)); i++) tab[i] = jac_madd(tab[i - 1], p);
  Jac acc = jac_inf();
  for (int w = num_windows<C>() - 1; w >= 0; w--) {
    for (int d = 0; d < C; d++) acc = jac_dbl(acc);
    int dig = recode_digit<C>(sp, w);
    if (dig > 0) acc = jac_add(acc, tab[dig - 1]);
    else if (dig < 0) acc = jac_add(acc, jac_neg(tab[-dig - 1]));
  }
  store_pt(out, acc);
  return 0;
}
extern "C" {
int h29_fp(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) { return binop<FP>(op, a, b, out); }
int h29_fn(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) { return binop<FN>(op, a, b, out); }

// op 0: madd(a, b)  1: add(jac a, jac b') with b' rescaled by a random-ish Z  2: dbl(a)
// op 3: the same sum through the extended-Jacobian accumulator: ((identity + a) + b) by xyzz_madd, back through xyzz_to_jac
// op 4: (a rescaled to a non-trivial ZZ / ZZZ) + b by xyzz_madd_nzq (b must not be the identity)
int h29_point(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  Aff p, q;
  if (!load_pt(p, a) || !load_pt(q, b)) return -1;
  Jac pj = jac_from_aff(p), r;
  if (op == 0) r = jac_madd(pj, q);
  else if (op == 1) {
    // give both operands non-trivial Z: (X z^2, Y z^3, z)
    Jac qj = jac_from_aff(q);
    Fp z = to_mont(unpack<FP>((const uint32_t[8]){0x12345, 7, 9, 0, 0, 0, 0, 0}));
    Fp z2 = sqr(z), z3 = mul(z2, z);
    if (!jac_is_inf(qj)) { qj.X = mul(qj.X, z2); qj.Y = mul(qj.Y, z3); qj.Z = mul(qj.Z, z); }
    if (!jac_is_inf(pj)) { Fp w = add(z, z2), w2 = sqr(w), w3 = mul(w2, w); pj.X = mul(pj.X, w2); pj.Y = mul(pj.Y, w3); pj.Z = mul(pj.Z, w); }
    r = jac_add(pj, qj);
  } else if (op == 3) {
    Xyzz acc = xyzz_madd(xyzz_madd(xyzz_inf(), p), q);
    r = xyzz_to_jac(acc);
  } else if (op == 4) {
    if (aff_is_inf(q)) return -3;
    Jac t = pj;
    Fp z = to_mont(unpack<FP>((const uint32_t[8]){0x54321, 11, 5, 0, 0, 0, 0, 0}));
    if (!jac_is_inf(t)) { Fp z2 = sqr(z), z3 = mul(z2, z); t.X = mul(t.X, z2); t.Y = mul(t.Y, z3); t.Z = mul(t.Z, z); }
    r = xyzz_to_jac(xyzz_madd_nzq(xyzz_from_jac(t), q));
  } else r = jac_dbl(pj);
  store_pt(out, r);
  return 0;
}
int h29_scalar_mul(int c, const uint8_t *s, const uint8_t *pt, uint8_t *out) {
  switch (c) {
    case 2: return smul<2>(s, pt, out);
    case 4: return smul<4>(s, pt, out);
    case 5: return smul<5>(s, pt, out);
    case 7: return smul<7>(s, pt, out);
    default: return -2;
  }
}
// digits of recode for window size c, to check sum d_w 2^(cw) == s
int h29_recode(int c, const uint8_t *s, int *digits) {
  uint32_t sw[8], sp[9];
  memcpy(sw, s, 32);
  int n = 0;
#define DO(C) { recode_add_k<C>(sp, sw); n = num_windows<C>(); for (int w = 0; w < n; w++) digits[w] = recode_digit<C>(sp, w); }
  switch (c) { case 4: DO(4) break; case 8: DO(8) break; case 12: DO(12) break; case 13: DO(13) break; case 16: DO(16) break; default: return -1; }
  return n;
}
}

// ---- square roots (fe29_sqrt.cuh): tables built once on the CPU
#include "../../mpc_bulletproof_amd/csrc/fe29_sqrt.cuh"
static int32_t g_sqrt_T[SQ_DIG * 256 * NL];
static uint8_t g_sqrt_hash[65536];
static bool g_sqrt_ready = false;
extern "C" {
// -> 1: root written (canonical bytes), 0: non-residue, -1: malformed; -2: the hash table collides
int h29_sqrt(const uint8_t *a, uint8_t *out) {
  if (!g_sqrt_ready) {
    memset(g_sqrt_hash, 0xff, sizeof g_sqrt_hash);
    for (int j = 0; j < SQ_DIG; j++)
      for (int d = 0; d < 256; d++) {
        Fp e = sqrt_table_entry(j, d);
        for (int t = 0; t < NL; t++) g_sqrt_T[((size_t)j * 256 + d) * NL + t] = e.v[t];
        if (j == SQ_DIG - 1) {
          uint32_t h = sqrt_hash(e);
          if (g_sqrt_hash[h] != 0xff) return -2;
          g_sqrt_hash[h] = (uint8_t)((256 - d) & 255);
        }
      }
    g_sqrt_ready = true;
  }
  Fp x, r;
  if (!load_fe(x, a)) return -1;
  if (!fp_sqrt(r, x, g_sqrt_T, g_sqrt_hash)) return 0;
  store_fe(out, r);
  return 1;
}
}
