/* bpo_transcript.c -- CPU oracle: keccak256, hash_to_scalar, hash-chain transcript, generator
 * chain, util.rs helpers, injectable RNG.  TEST INFRASTRUCTURE ONLY (see bpo.h).
 *
 * The transcript is a stand-in for merlin's HashChainTranscript (git fork, source absent):
 * PARITY UNPINNED.  Definition (shared with oracle/pymodel.py):
 *   state_0         = keccak256(pad_label("bp-hashchain-v1") || pad_label(label))
 *   append_message  : state = keccak256(state || 0x00 || pad_label(l) || u32le(len) || msg)
 *   challenge_bytes : state = keccak256(state || 0x01 || pad_label(l)); output = state
 *   pad_label(l)    = l right-padded with zeros to a multiple of 32 bytes (min 32)
 */
#include "bpo.h"
#include <stdlib.h>
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int RHO[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PI[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

static void keccak_f(uint64_t st[25]) {
  for (int r = 0; r < 24; r++) {
    uint64_t bc[5], t;
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      t = bc[(i + 4) % 5] ^ ROL(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    t = st[1];
    for (int i = 0; i < 24; i++) {
      int j = PI[i];
      uint64_t b = st[j];
      st[j] = ROL(t, RHO[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= RC[r];
  }
}
/* Original Keccak-256 (padding 0x01): merlin fork `keccak256`, util.rs:255 */
void bpo_keccak256(const uint8_t *in, size_t len, uint8_t out[32]) {
  uint64_t st[25];
  uint8_t blk[136];
  memset(st, 0, sizeof st);
  const size_t rate = 136;
  while (len >= rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w = 0;
      for (int j = 7; j >= 0; j--) w = (w << 8) | in[8 * i + j];
      st[i] ^= w;
    }
    keccak_f(st);
    in += rate;
    len -= rate;
  }
  memset(blk, 0, rate);
  memcpy(blk, in, len);
  blk[len] ^= 0x01;
  blk[rate - 1] ^= 0x80;
  for (size_t i = 0; i < rate / 8; i++) {
    uint64_t w = 0;
    for (int j = 7; j >= 0; j--) w = (w << 8) | blk[8 * i + j];
    st[i] ^= w;
  }
  keccak_f(st);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(st[i] >> (8 * j));
}
/* BlindVec v1 (bpo.h): the device-drawn blinding vectors of the R1CS prover, restated */
void blind_vector(sc *out, const uint8_t key[32], int v, size_t count) {
  for (size_t j = 0; 2 * j < count; j++) {
    uint64_t st[25];
    memset(st, 0, sizeof st);
    for (int i = 0; i < 4; i++) {
      uint64_t w = 0;
      for (int b = 7; b >= 0; b--) w = (w << 8) | key[8 * i + b];
      st[i] = w;
    }
    st[4] = (uint64_t)v;
    st[5] = (uint64_t)j;
    st[6] = 0x01;                       /* pad: 0x01 right after the 48 message bytes ... */
    st[16] = 0x8000000000000000ULL;     /* ... 0x80 at byte 135 (rate 136) */
    keccak_f(st);
    for (int h = 0; h < 2 && 2 * j + h < count; h++) {
      uint8_t wide[64];
      for (int i = 0; i < 8; i++)
        for (int b = 0; b < 8; b++) wide[8 * i + b] = (uint8_t)(st[8 * h + i] >> (8 * b));
      fe_from_le_wide(SC, &out[2 * j + h], wide);
    }
  }
}
/* util.rs:252-267 */
void hash_to_scalar(sc *r, const uint8_t low[32]) {
  uint8_t buf[64];
  memcpy(buf, low, 32);
  bpo_keccak256(low, 32, buf + 32);
  fe_from_le_wide(SC, r, buf);
}

static size_t pad_label(uint8_t *dst, const uint8_t *label, size_t len) {
  size_t k = (len + 31) / 32 * 32;
  if (k < 32) k = 32;
  memset(dst, 0, k);
  memcpy(dst, label, len);
  return k;
}
void tr_init(transcript *t, const uint8_t *label, size_t len) {
  uint8_t buf[32 + 256];
  size_t o = pad_label(buf, (const uint8_t *)"bp-hashchain-v1", 15);
  o += pad_label(buf + o, label, len > 200 ? 200 : len);
  bpo_keccak256(buf, o, t->state);
}
void tr_append_message(transcript *t, const char *label, const uint8_t *msg, size_t len) {
  size_t ll = strlen(label);
  uint8_t *buf = (uint8_t *)malloc(32 + 1 + ll + 64 + 8 + len);
  size_t o = 0;
  memcpy(buf, t->state, 32); o = 32;
  buf[o++] = 0x00;
  o += pad_label(buf + o, (const uint8_t *)label, ll);
  for (int j = 0; j < 4; j++) buf[o++] = (uint8_t)((uint64_t)len >> (8 * j));   /* u32le, as merlin frames lengths */
  memcpy(buf + o, msg, len); o += len;
  bpo_keccak256(buf, o, t->state);
  free(buf);
}
void tr_append_u64(transcript *t, const char *label, uint64_t x) {
  uint8_t b[8];
  for (int j = 0; j < 8; j++) b[j] = (uint8_t)(x >> (8 * j));
  tr_append_message(t, label, b, 8);
}
void tr_challenge_bytes(transcript *t, const char *label, uint8_t out[32]) {
  size_t ll = strlen(label);
  uint8_t buf[32 + 1 + 256];
  memcpy(buf, t->state, 32);
  buf[32] = 0x01;
  size_t o = 33 + pad_label(buf + 33, (const uint8_t *)label, ll);
  bpo_keccak256(buf, o, t->state);
  memcpy(out, t->state, 32);
}
static void dom_sep(transcript *t, const char *s) {
  uint8_t buf[64];
  size_t k = pad_label(buf, (const uint8_t *)s, strlen(s));
  tr_append_message(t, "dom-sep", buf, k);
}
/* src/transcript.rs:70-85 */
void tr_innerproduct_domain_sep(transcript *t, uint64_t n) { dom_sep(t, "ipp v1"); tr_append_u64(t, "n", n); }
void tr_r1cs_domain_sep(transcript *t) { dom_sep(t, "r1cs v1"); }
void tr_r1cs_1phase_domain_sep(transcript *t) { dom_sep(t, "r1cs-1phase"); }
void tr_r1cs_2phase_domain_sep(transcript *t) { dom_sep(t, "r1cs-2phase"); }
/* src/transcript.rs:87-92: little-endian canonical bytes */
void tr_append_scalar(transcript *t, const char *label, const sc *s) {
  uint8_t b[32];
  fe_to_le(SC, b, s);
  tr_append_message(t, label, b, 32);
}
/* src/transcript.rs:94-97 + util.rs:274-289 */
void tr_append_point(transcript *t, const char *label, const aff *p) {
  uint8_t b[64];
  aff_to_bytes(b, p);
  tr_append_message(t, label, b, 64);
}
/* src/transcript.rs:101-113 */
int tr_validate_and_append_point(transcript *t, const char *label, const aff *p) {
  if (p->inf) return -1;
  tr_append_point(t, label, p);
  return 0;
}
/* src/transcript.rs:115-120 */
void tr_challenge_scalar(transcript *t, const char *label, sc *out) {
  uint8_t b[32];
  tr_challenge_bytes(t, label, b);
  hash_to_scalar(out, b);
}

/* src/generators.rs:82-89 (new), :112-124 (next), :217-232 (labels 'G'/'H' + LE party index) */
void gens_chain(aff *out, sc *dlog, char which, uint32_t party, size_t count) {
  uint8_t lab[64], padded[64], state[32];
  memcpy(lab, "GeneratorsChain", 15);
  lab[15] = (uint8_t)which;
  for (int j = 0; j < 4; j++) lab[16 + j] = (uint8_t)(party >> (8 * j));
  size_t k = pad_label(padded, lab, 20);
  bpo_keccak256(padded, k, state);
  jac g, t;
  jac_from_aff(&g, &BPO_G);
  jac *tmp = (jac *)malloc(count * sizeof(jac));
  for (size_t i = 0; i < count; i++) {
    uint8_t nx[32];
    bpo_keccak256(state, 32, nx);
    memcpy(state, nx, 32);
    sc s;
    hash_to_scalar(&s, state);
    if (dlog) dlog[i] = s;
    jac_mul(&t, &g, &s);
    tmp[i] = t;
  }
  batch_to_aff(out, tmp, count);
  free(tmp);
}

uint64_t sm_next(splitmix *r) {
  r->s += 0x9E3779B97F4A7C15ULL;
  uint64_t z = r->s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* 4 x u64 little-endian limbs reduced mod n (SURVEY.md 8d synthetic inputs) */
void sm_scalar(splitmix *r, sc *out) {
  uint8_t b[64];
  memset(b, 0, 64);
  for (int i = 0; i < 4; i++) {
    uint64_t w = sm_next(r);
    for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(w >> (8 * j));
  }
  fe_from_le_wide(SC, out, b);
}

/* util.rs:73-76 */
void exp_iter(sc *out, const sc *x, size_t n) {
  sc cur = SC->one;
  for (size_t i = 0; i < n; i++) {
    out[i] = cur;
    fe_mul(SC, &cur, &cur, x);
  }
}
/* util.rs:237-239 */
void sum_of_powers_slow(sc *out, const sc *x, size_t n) {
  sc acc, cur = SC->one;
  memset(&acc, 0, sizeof acc);
  for (size_t i = 0; i < n; i++) {
    fe_add(SC, &acc, &acc, &cur);
    fe_mul(SC, &cur, &cur, x);
  }
  *out = acc;
}
/* util.rs:218-234 */
void sum_of_powers(sc *out, const sc *x, size_t n) {
  if (n & (n - 1)) { sum_of_powers_slow(out, x, n); return; }
  if (n == 0 || n == 1) { fe_from_u64(SC, out, n); return; }
  size_t m = n;
  sc result, factor = *x, t;
  fe_add(SC, &result, &SC->one, x);
  while (m > 2) {
    fe_mul(SC, &factor, &factor, &factor);
    fe_mul(SC, &t, &factor, &result);
    fe_add(SC, &result, &result, &t);
    m /= 2;
  }
  *out = result;
}
