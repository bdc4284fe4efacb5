/* bpo_algebra.c -- CPU oracle: F_p / F_n Montgomery arithmetic, Stark-curve group law, MSM.
 * TEST INFRASTRUCTURE ONLY (see bpo.h).  Restates the external crate mpc-stark 0.2
 * (Scalar, StarkPoint, msm, batch_inverse; call sites listed in SURVEY.md K1-K10). */
#include "bpo.h"
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

const fctx BPO_FP = {
    {0x0000000000000001ULL, 0x0000000000000000ULL, 0x0000000000000000ULL, 0x0800000000000011ULL},
    0xffffffffffffffffULL,
    {{0xfffffd737e000401ULL, 0x00000001330fffffULL, 0xffffffffff6f8000ULL, 0x07ffd4ab5e008810ULL}},
    {{0xffffffffffffffe1ULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL, 0x07fffffffffffdf0ULL}}};
const fctx BPO_FN = {
    {0x1e66a241adc64d2fULL, 0xb781126dcae7b232ULL, 0xffffffffffffffffULL, 0x0800000000000010ULL},
    0xbb6b3c4ce8bde631ULL,
    {{0x6021b3f1ea1c688dULL, 0x509cf64d14ce60b9ULL, 0xbaf0ab4cf78bbabbULL, 0x07d9e57c2333766eULL}},
    {{0x51925a0bf4fca74fULL, 0xc75ec4b46df16beeULL, 0x0000000000000008ULL, 0x07fffffffffffdf1ULL}}};

/* curve: y^2 = x^3 + x + b, generator (Gx, Gy); Montgomery form */
static const fe CURVE_B = {{0x359ddd67b59a21caULL, 0x6725f2237aab9006ULL, 0xab8a1e002a41f947ULL, 0x013931651774247fULL}};
const aff BPO_G = {
    {{0xc9019623cf0273ddULL, 0x51a9bf65d4403deaULL, 0x0429bf5184041c7bULL, 0x033840300bf6cec1ULL}},
    {{0x569d0da34235308aULL, 0x0939e3442869bbe7ULL, 0xfbd89a97cf4b33adULL, 0x05a0e71610f55329ULL}},
    0};

/* ---------------------------------------------------------------- helpers */
static inline int geq(const uint64_t a[4], const uint64_t m[4]) {
  for (int i = 3; i >= 0; i--) {
    if (a[i] > m[i]) return 1;
    if (a[i] < m[i]) return 0;
  }
  return 1;
}
static inline uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - b[i] - (uint64_t)br;
    r[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  return (uint64_t)br;
}
static inline uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}

void fe_add(const fctx *c, fe *r, const fe *a, const fe *b) {
  uint64_t t[4];
  add4(t, a->v, b->v); /* both < m < 2^252: no carry out */
  if (geq(t, c->m)) sub4(t, t, c->m);
  memcpy(r->v, t, 32);
}
void fe_sub(const fctx *c, fe *r, const fe *a, const fe *b) {
  uint64_t t[4];
  if (sub4(t, a->v, b->v)) add4(t, t, c->m);
  memcpy(r->v, t, 32);
}
void fe_neg(const fctx *c, fe *r, const fe *a) {
  if (fe_is_zero(a)) { memset(r, 0, sizeof *r); return; }
  uint64_t t[4];
  sub4(t, c->m, a->v);
  memcpy(r->v, t, 32);
}
/* CIOS Montgomery multiplication, 4 x 64-bit limbs */
void fe_mul(const fctx *c, fe *r, const fe *a, const fe *b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 cy = 0;
    for (int j = 0; j < 4; j++) {
      cy += (u128)a->v[j] * b->v[i] + t[j];
      t[j] = (uint64_t)cy;
      cy >>= 64;
    }
    cy += t[4];
    t[4] = (uint64_t)cy;
    t[5] = (uint64_t)(cy >> 64);
    uint64_t mq = t[0] * c->n0;
    cy = (u128)mq * c->m[0] + t[0];
    cy >>= 64;
    for (int j = 1; j < 4; j++) {
      cy += (u128)mq * c->m[j] + t[j];
      t[j - 1] = (uint64_t)cy;
      cy >>= 64;
    }
    cy += t[4];
    t[3] = (uint64_t)cy;
    t[4] = t[5] + (uint64_t)(cy >> 64);
  }
  if (t[4] || geq(t, c->m)) sub4(t, t, c->m);
  memcpy(r->v, t, 32);
}
static void fe_pow(const fctx *c, fe *r, const fe *a, const uint64_t e[4]) {
  fe acc = c->one, base = *a;
  for (int i = 0; i < 256; i++) {
    if ((e[i / 64] >> (i % 64)) & 1) fe_mul(c, &acc, &acc, &base);
    fe_mul(c, &base, &base, &base);
  }
  *r = acc;
}
void fe_inv(const fctx *c, fe *r, const fe *a) {
  uint64_t e[4], two[4] = {2, 0, 0, 0};
  sub4(e, c->m, two);
  fe_pow(c, r, a, e);
}
void fe_from_u64(const fctx *c, fe *r, uint64_t x) {
  fe t = {{x, 0, 0, 0}};
  fe_mul(c, r, &t, &c->r2);
}
int fe_from_le(const fctx *c, fe *r, const uint8_t b[32]) {
  fe t;
  for (int i = 0; i < 4; i++) {
    uint64_t w = 0;
    for (int j = 7; j >= 0; j--) w = (w << 8) | b[8 * i + j];
    t.v[i] = w;
  }
  if (geq(t.v, c->m)) return -1;
  fe_mul(c, r, &t, &c->r2);
  return 0;
}
/* 512-bit little-endian integer mod m: lo + hi * 2^256, each half reduced first */
void fe_from_le_wide(const fctx *c, fe *r, const uint8_t b[64]) {
  fe lo, hi;
  for (int h = 0; h < 2; h++) {
    fe *d = h ? &hi : &lo;
    for (int i = 0; i < 4; i++) {
      uint64_t w = 0;
      for (int j = 7; j >= 0; j--) w = (w << 8) | b[32 * h + 8 * i + j];
      d->v[i] = w;
    }
    /* value < 2^256 < 32*m : subtract until canonical */
    while (geq(d->v, c->m)) sub4(d->v, d->v, c->m);
  }
  /* to Montgomery: lo*R, hi*R ; then hi*R * (R mod m as Montgomery = R*R) -> hi*R*R ... */
  fe lom, him;
  fe_mul(c, &lom, &lo, &c->r2);          /* lo * R */
  fe_mul(c, &him, &hi, &c->r2);          /* hi * R */
  fe_mul(c, &him, &him, &c->r2);         /* hi * R * R^2 / R = hi * R^2 = (hi * 2^256) * R */
  fe_add(c, r, &lom, &him);
}
void fe_to_int(const fctx *c, uint64_t out[4], const fe *a) {
  fe one = {{1, 0, 0, 0}}, t;
  fe_mul(c, &t, a, &one);
  memcpy(out, t.v, 32);
}
void fe_to_le(const fctx *c, uint8_t b[32], const fe *a) {
  uint64_t t[4];
  fe_to_int(c, t, a);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(t[i] >> (8 * j));
}
int fe_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
int fe_eq(const fe *a, const fe *b) { return memcmp(a->v, b->v, 32) == 0; }

/* Scalar::batch_inverse -- Montgomery's trick.  Inputs non-zero (challenges). */
void sc_batch_inverse(sc *v, size_t n) {
  if (!n) return;
  sc *pref = (sc *)malloc(n * sizeof(sc));
  sc acc = SC->one;
  for (size_t i = 0; i < n; i++) {
    pref[i] = acc;
    fe_mul(SC, &acc, &acc, &v[i]);
  }
  sc ai;
  fe_inv(SC, &ai, &acc);
  for (size_t i = n; i-- > 0;) {
    sc t;
    fe_mul(SC, &t, &ai, &pref[i]);
    fe_mul(SC, &ai, &ai, &v[i]);
    v[i] = t;
  }
  free(pref);
}
/* src/inner_product_proof.rs:463-472 */
void sc_inner_product(sc *out, const sc *a, const sc *b, size_t n) {
  sc acc, t;
  memset(&acc, 0, sizeof acc);
  for (size_t i = 0; i < n; i++) {
    fe_mul(SC, &t, &a[i], &b[i]);
    fe_add(SC, &acc, &acc, &t);
  }
  *out = acc;
}

/* ---------------------------------------------------------------- curve */
#define FP (&BPO_FP)
#define M(r, a, b) fe_mul(FP, r, a, b)
#define A(r, a, b) fe_add(FP, r, a, b)
#define S(r, a, b) fe_sub(FP, r, a, b)

void jac_set_inf(jac *r) { memset(r, 0, sizeof *r); r->X = FP->one; r->Y = FP->one; }
int jac_is_inf(const jac *a) { return fe_is_zero(&a->Z); }
void jac_from_aff(jac *r, const aff *a) {
  if (a->inf) { jac_set_inf(r); return; }
  r->X = a->x; r->Y = a->y; r->Z = FP->one;
}
void jac_to_aff(aff *r, const jac *a) {
  if (jac_is_inf(a)) { memset(r, 0, sizeof *r); r->inf = 1; return; }
  fe zi, zi2, zi3;
  fe_inv(FP, &zi, &a->Z);
  M(&zi2, &zi, &zi);
  M(&zi3, &zi2, &zi);
  M(&r->x, &a->X, &zi2);
  M(&r->y, &a->Y, &zi3);
  r->inf = 0;
}
void jac_neg(jac *r, const jac *a) { *r = *a; fe_neg(FP, &r->Y, &a->Y); }

/* dbl-2007-bl with a = 1 */
void jac_dbl(jac *r, const jac *p) {
  if (jac_is_inf(p) || fe_is_zero(&p->Y)) { jac_set_inf(r); return; }
  fe XX, YY, YYYY, ZZ, s, m, t, y3, z3;
  M(&XX, &p->X, &p->X);
  M(&YY, &p->Y, &p->Y);
  M(&YYYY, &YY, &YY);
  M(&ZZ, &p->Z, &p->Z);
  A(&s, &p->X, &YY); M(&s, &s, &s); S(&s, &s, &XX); S(&s, &s, &YYYY); A(&s, &s, &s); /* S = 2((X+YY)^2-XX-YYYY) */
  A(&m, &XX, &XX); A(&m, &m, &XX); M(&t, &ZZ, &ZZ); A(&m, &m, &t);                  /* M = 3XX + a ZZ^2 */
  M(&t, &m, &m); S(&t, &t, &s); S(&t, &t, &s);                                        /* T = M^2 - 2S */
  A(&z3, &p->Y, &p->Z); M(&z3, &z3, &z3); S(&z3, &z3, &YY); S(&z3, &z3, &ZZ);        /* Z3 = (Y+Z)^2-YY-ZZ */
  S(&y3, &s, &t); M(&y3, &y3, &m);
  A(&YYYY, &YYYY, &YYYY); A(&YYYY, &YYYY, &YYYY); A(&YYYY, &YYYY, &YYYY);
  S(&y3, &y3, &YYYY);
  r->X = t; r->Y = y3; r->Z = z3;
}
/* add-2007-bl, made complete */
void jac_add(jac *r, const jac *p, const jac *q) {
  if (jac_is_inf(p)) { *r = *q; return; }
  if (jac_is_inf(q)) { *r = *p; return; }
  fe Z1Z1, Z2Z2, U1, U2, S1, S2, H, I, J, rr, V, t, x3, y3, z3;
  M(&Z1Z1, &p->Z, &p->Z); M(&Z2Z2, &q->Z, &q->Z);
  M(&U1, &p->X, &Z2Z2); M(&U2, &q->X, &Z1Z1);
  M(&S1, &p->Y, &q->Z); M(&S1, &S1, &Z2Z2);
  M(&S2, &q->Y, &p->Z); M(&S2, &S2, &Z1Z1);
  S(&H, &U2, &U1);
  S(&rr, &S2, &S1);
  if (fe_is_zero(&H)) {
    if (fe_is_zero(&rr)) { jac_dbl(r, p); return; }
    jac_set_inf(r); return;
  }
  A(&I, &H, &H); M(&I, &I, &I);
  M(&J, &H, &I);
  A(&rr, &rr, &rr);
  M(&V, &U1, &I);
  M(&x3, &rr, &rr); S(&x3, &x3, &J); S(&x3, &x3, &V); S(&x3, &x3, &V);
  S(&y3, &V, &x3); M(&y3, &y3, &rr); M(&t, &S1, &J); A(&t, &t, &t); S(&y3, &y3, &t);
  A(&z3, &p->Z, &q->Z); M(&z3, &z3, &z3); S(&z3, &z3, &Z1Z1); S(&z3, &z3, &Z2Z2); M(&z3, &z3, &H);
  r->X = x3; r->Y = y3; r->Z = z3;
}
void jac_madd(jac *r, const jac *p, const aff *q) {
  jac t;
  jac_from_aff(&t, q);
  jac_add(r, p, &t);
}
int jac_eq(const jac *a, const jac *b) {
  int ia = jac_is_inf(a), ib = jac_is_inf(b);
  if (ia || ib) return ia && ib;
  fe za, zb, t1, t2;
  M(&za, &a->Z, &a->Z); M(&zb, &b->Z, &b->Z);
  M(&t1, &a->X, &zb); M(&t2, &b->X, &za);
  if (!fe_eq(&t1, &t2)) return 0;
  M(&za, &za, &a->Z); M(&zb, &zb, &b->Z);
  M(&t1, &a->Y, &zb); M(&t2, &b->Y, &za);
  return fe_eq(&t1, &t2);
}
void jac_mul(jac *r, const jac *p, const sc *k) {
  uint64_t e[4];
  fe_to_int(SC, e, k);
  jac acc;
  jac_set_inf(&acc);
  for (int i = 255; i >= 0; i--) {
    jac_dbl(&acc, &acc);
    if ((e[i / 64] >> (i % 64)) & 1) jac_add(&acc, &acc, p);
  }
  *r = acc;
}
int aff_from_bytes(aff *r, const uint8_t b[64]) {
  int z = 1;
  for (int i = 0; i < 64; i++) if (b[i]) { z = 0; break; }
  memset(r, 0, sizeof *r);
  if (z) { r->inf = 1; return 0; }
  if (fe_from_le(FP, &r->x, b) || fe_from_le(FP, &r->y, b + 32)) return -1;
  fe l, rr, t;
  M(&l, &r->y, &r->y);
  M(&rr, &r->x, &r->x); M(&rr, &rr, &r->x); A(&rr, &rr, &r->x); t = CURVE_B; A(&rr, &rr, &t);
  return fe_eq(&l, &rr) ? 0 : -1;
}
void aff_to_bytes(uint8_t b[64], const aff *a) {
  if (a->inf) { memset(b, 0, 64); return; }
  fe_to_le(FP, b, &a->x);
  fe_to_le(FP, b + 32, &a->y);
}
void jac_to_bytes(uint8_t b[64], const jac *a) {
  aff t;
  jac_to_aff(&t, a);
  aff_to_bytes(b, &t);
}
void batch_to_aff(aff *out, const jac *in, size_t n) {
  if (!n) return;
  fe *pref = (fe *)malloc(n * sizeof(fe));
  fe acc = FP->one;
  for (size_t i = 0; i < n; i++) {
    pref[i] = acc;
    if (!jac_is_inf(&in[i])) M(&acc, &acc, &in[i].Z);
  }
  fe ai;
  fe_inv(FP, &ai, &acc);
  for (size_t i = n; i-- > 0;) {
    if (jac_is_inf(&in[i])) { memset(&out[i], 0, sizeof(aff)); out[i].inf = 1; continue; }
    fe zi, zi2, zi3;
    M(&zi, &ai, &pref[i]);
    M(&ai, &ai, &in[i].Z);
    M(&zi2, &zi, &zi); M(&zi3, &zi2, &zi);
    M(&out[i].x, &in[i].X, &zi2);
    M(&out[i].y, &in[i].Y, &zi3);
    out[i].inf = 0;
  }
  free(pref);
}

/* ---------------------------------------------------------------- MSM */
void msm_naive(jac *r, const sc *s, const aff *p, size_t n) {
  jac acc, t, pj;
  jac_set_inf(&acc);
  for (size_t i = 0; i < n; i++) {
    jac_from_aff(&pj, &p[i]);
    jac_mul(&t, &pj, &s[i]);
    jac_add(&acc, &acc, &t);
  }
  *r = acc;
}
/* Pippenger bucket method as in ark-ec 0.4 VariableBaseMSM (which mpc-stark's
 * StarkPoint::msm delegates to [memory; crate absent]): window c = 3 for n < 32 else
 * ln(n) + 2, unsigned digits, per-window bucket running sums, windows combined
 * high-to-low with c doublings. */
static unsigned ln_without_floats(size_t a) {
  unsigned lg = 0;
  while ((a >> lg) > 1) lg++;
  return lg * 69 / 100;
}
void msm_pippenger(jac *r, const sc *s, const aff *p, size_t n) {
  if (n == 0) { jac_set_inf(r); return; }
  unsigned c = n < 32 ? 3 : ln_without_floats(n) + 2;
  const unsigned bits = 252;
  unsigned nwin = (bits + c - 1) / c;
  size_t nb = ((size_t)1 << c) - 1;
  uint64_t(*e)[4] = malloc(n * sizeof *e);
  for (size_t i = 0; i < n; i++) fe_to_int(SC, e[i], &s[i]);
  jac *buckets = (jac *)malloc(nb * sizeof(jac));
  jac total;
  jac_set_inf(&total);
  for (int w = (int)nwin - 1; w >= 0; w--) {
    for (unsigned d = 0; d < c; d++) jac_dbl(&total, &total);
    for (size_t b = 0; b < nb; b++) jac_set_inf(&buckets[b]);
    unsigned lo = (unsigned)w * c;
    for (size_t i = 0; i < n; i++) {
      uint64_t dig = e[i][lo / 64] >> (lo % 64);
      if (lo % 64 + c > 64 && lo / 64 + 1 < 4) dig |= e[i][lo / 64 + 1] << (64 - lo % 64);
      dig &= nb;
      if (dig && !p[i].inf) jac_madd(&buckets[dig - 1], &buckets[dig - 1], &p[i]);
    }
    jac run, sum;
    jac_set_inf(&run);
    jac_set_inf(&sum);
    for (size_t b = nb; b-- > 0;) {
      jac_add(&run, &run, &buckets[b]);
      jac_add(&sum, &sum, &run);
    }
    jac_add(&total, &total, &sum);
  }
  free(buckets);
  free(e);
  *r = total;
}
void msm(jac *r, const sc *s, const aff *p, size_t n) {
  if (n < 8) msm_naive(r, s, p, n);
  else msm_pippenger(r, s, p, n);
}
