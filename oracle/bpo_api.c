/* bpo_api.c -- flat byte-oriented entry points of the CPU oracle for ctypes (tests/, smoke(),
 * bench.py cpu_baseline).  TEST INFRASTRUCTURE ONLY (see bpo.h).
 *
 * Encodings (same as the product's C-ABI, include/bpgpu.h): scalar = 32-byte little-endian
 * canonical integer < n; point = affine x||y, 32-byte LE each, 64 zero bytes = identity.
 * Flat proof ("flat v0", NOT the reference wire format r1cs/proof.rs:82-109, whose 32-byte point
 * compression lives in the absent mpc-stark crate):
 *   u32 k | u32 0 | A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 (64 B each) |
 *   t_x t_x_blinding e_blinding (32 B each) | L_0..L_{k-1} | R_0..R_{k-1} | a | b
 */
#include "bpo.h"
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

static int load_scalars(sc *out, const uint8_t *in, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (fe_from_le(SC, &out[i], in + 32 * i)) return -1;
  return 0;
}
static void store_scalars(uint8_t *out, const sc *in, size_t n) {
  for (size_t i = 0; i < n; i++) fe_to_le(SC, out + 32 * i, &in[i]);
}
static int load_points(aff *out, const uint8_t *in, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (aff_from_bytes(&out[i], in + 64 * i)) return -1;
  return 0;
}
static void store_points(uint8_t *out, const aff *in, size_t n) {
  for (size_t i = 0; i < n; i++) aff_to_bytes(out + 64 * i, &in[i]);
}

/* ------------------------------------------------------------------ scalars */
API int bpo_sc_binop(int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out) {
  for (size_t i = 0; i < n; i++) {
    sc x, y, r;
    if (fe_from_le(SC, &x, a + 32 * i) || fe_from_le(SC, &y, b + 32 * i)) return -1;
    if (op == 0) fe_add(SC, &r, &x, &y);
    else if (op == 1) fe_sub(SC, &r, &x, &y);
    else fe_mul(SC, &r, &x, &y);
    fe_to_le(SC, out + 32 * i, &r);
  }
  return 0;
}
API int bpo_sc_inv(const uint8_t *a, size_t n, uint8_t *out) {
  for (size_t i = 0; i < n; i++) {
    sc x, r;
    if (fe_from_le(SC, &x, a + 32 * i)) return -1;
    fe_inv(SC, &r, &x);
    fe_to_le(SC, out + 32 * i, &r);
  }
  return 0;
}
API int bpo_sc_batch_inverse(uint8_t *v, size_t n) {
  sc *t = (sc *)malloc((n ? n : 1) * sizeof(sc));
  if (load_scalars(t, v, n)) { free(t); return -1; }
  sc_batch_inverse(t, n);
  store_scalars(v, t, n);
  free(t);
  return 0;
}
API int bpo_inner_product(const uint8_t *a, const uint8_t *b, size_t n, uint8_t out[32]) {
  sc *x = (sc *)malloc((n ? n : 1) * sizeof(sc)), *y = (sc *)malloc((n ? n : 1) * sizeof(sc)), r;
  int rc = load_scalars(x, a, n) || load_scalars(y, b, n);
  if (!rc) { sc_inner_product(&r, x, y, n); fe_to_le(SC, out, &r); }
  free(x); free(y);
  return rc ? -1 : 0;
}
API void bpo_hash_to_scalar(const uint8_t low[32], uint8_t out[32]) {
  sc r;
  hash_to_scalar(&r, low);
  fe_to_le(SC, out, &r);
}
API int bpo_exp_iter(const uint8_t x[32], size_t n, uint8_t *out) {
  sc xs, *t = (sc *)malloc((n ? n : 1) * sizeof(sc));
  if (fe_from_le(SC, &xs, x)) { free(t); return -1; }
  exp_iter(t, &xs, n);
  store_scalars(out, t, n);
  free(t);
  return 0;
}
API int bpo_sum_of_powers(const uint8_t x[32], size_t n, int slow, uint8_t out[32]) {
  sc xs, r;
  if (fe_from_le(SC, &xs, x)) return -1;
  if (slow) sum_of_powers_slow(&r, &xs, n); else sum_of_powers(&r, &xs, n);
  fe_to_le(SC, out, &r);
  return 0;
}
API void bpo_random_scalars(uint64_t seed, size_t n, uint8_t *out) {
  splitmix r = {seed};
  for (size_t i = 0; i < n; i++) { sc s; sm_scalar(&r, &s); fe_to_le(SC, out + 32 * i, &s); }
}

/* ------------------------------------------------------------------ points */
API int bpo_point_add(const uint8_t a[64], const uint8_t b[64], uint8_t out[64]) {
  aff x, y;
  if (aff_from_bytes(&x, a) || aff_from_bytes(&y, b)) return -1;
  jac j;
  jac_from_aff(&j, &x);
  jac_madd(&j, &j, &y);
  jac_to_bytes(out, &j);
  return 0;
}
API int bpo_point_mul(const uint8_t s[32], const uint8_t p[64], uint8_t out[64]) {
  aff x; sc k;
  if (aff_from_bytes(&x, p) || fe_from_le(SC, &k, s)) return -1;
  jac j, r;
  jac_from_aff(&j, &x);
  jac_mul(&r, &j, &k);
  jac_to_bytes(out, &r);
  return 0;
}
/* algo: 0 auto, 1 naive, 2 pippenger */
API int bpo_msm(const uint8_t *scalars, const uint8_t *points, size_t n, int algo, uint8_t out[64]) {
  sc *s = (sc *)malloc((n ? n : 1) * sizeof(sc));
  aff *p = (aff *)malloc((n ? n : 1) * sizeof(aff));
  int rc = load_scalars(s, scalars, n) || load_points(p, points, n);
  if (!rc) {
    jac r;
    if (algo == 1) msm_naive(&r, s, p, n); else if (algo == 2) msm_pippenger(&r, s, p, n); else msm(&r, s, p, n);
    jac_to_bytes(out, &r);
  }
  free(s); free(p);
  return rc ? -1 : 0;
}
/* nb independent MSMs of n terms each (cpu_baseline of the batched verification MSM) */
API int bpo_msm_batch(const uint8_t *scalars, const uint8_t *points, size_t nb, size_t n, uint8_t *out) {
  for (size_t b = 0; b < nb; b++)
    if (bpo_msm(scalars + 32 * n * b, points + 64 * n * b, n, 0, out + 64 * b)) return -1;
  return 0;
}
API void bpo_gens(int which, uint32_t party, size_t n, uint8_t *out_points, uint8_t *out_dlogs) {
  aff *p = (aff *)malloc((n ? n : 1) * sizeof(aff));
  sc *d = (sc *)malloc((n ? n : 1) * sizeof(sc));
  gens_chain(p, d, (char)which, party, n);
  store_points(out_points, p, n);
  if (out_dlogs) store_scalars(out_dlogs, d, n);
  free(p); free(d);
}
API void bpo_generator(uint8_t out[64]) { aff_to_bytes(out, &BPO_G); }

/* ------------------------------------------------------------------ IPP */
API int bpo_fold_witness(size_t n, const uint8_t u[32], const uint8_t u_inv[32], const uint8_t *a,
                         const uint8_t *b, const uint8_t *G, const uint8_t *H, uint8_t *a_out,
                         uint8_t *b_out, uint8_t *G_out, uint8_t *H_out) {
  sc us, ui, *as = (sc *)malloc(2 * n * sizeof(sc)), *bs = (sc *)malloc(2 * n * sizeof(sc));
  aff *Gs = (aff *)malloc(2 * n * sizeof(aff)), *Hs = (aff *)malloc(2 * n * sizeof(aff));
  sc *ao = (sc *)malloc(n * sizeof(sc)), *bo = (sc *)malloc(n * sizeof(sc));
  aff *Go = (aff *)malloc(n * sizeof(aff)), *Ho = (aff *)malloc(n * sizeof(aff));
  int rc = fe_from_le(SC, &us, u) || fe_from_le(SC, &ui, u_inv) || load_scalars(as, a, 2 * n) ||
           load_scalars(bs, b, 2 * n) || load_points(Gs, G, 2 * n) || load_points(Hs, H, 2 * n);
  if (!rc) {
    fold_witness(&us, &ui, n, as, as + n, bs, bs + n, Gs, Gs + n, Hs, Hs + n, ao, bo, Go, Ho);
    store_scalars(a_out, ao, n); store_scalars(b_out, bo, n);
    store_points(G_out, Go, n); store_points(H_out, Ho, n);
  }
  free(as); free(bs); free(Gs); free(Hs); free(ao); free(bo); free(Go); free(Ho);
  return rc ? -1 : 0;
}
API int bpo_verification_scalars(const uint8_t *challenges, size_t k, size_t n, uint8_t *u_sq,
                                 uint8_t *u_inv_sq, uint8_t *s) {
  sc *ch = (sc *)malloc((k ? k : 1) * sizeof(sc)), *a = (sc *)malloc((k ? k : 1) * sizeof(sc));
  sc *b = (sc *)malloc((k ? k : 1) * sizeof(sc)), *ss = (sc *)malloc(n * sizeof(sc));
  int rc = load_scalars(ch, challenges, k);
  if (!rc) {
    verification_scalars(ch, k, n, a, b, ss);
    store_scalars(u_sq, a, k); store_scalars(u_inv_sq, b, k); store_scalars(s, ss, n);
  }
  free(ch); free(a); free(b); free(ss);
  return rc ? -1 : 0;
}
API int bpo_ipp_create(const uint8_t *label, size_t label_len, size_t n, const uint8_t Q[64],
                       const uint8_t *Gf, const uint8_t *Hf, const uint8_t *G, const uint8_t *H,
                       const uint8_t *a, const uint8_t *b, uint8_t *L_out, uint8_t *R_out,
                       uint8_t a_out[32], uint8_t b_out[32], uint8_t *challenges_out) {
  sc *gf = (sc *)malloc(n * sizeof(sc)), *hf = (sc *)malloc(n * sizeof(sc));
  sc *as = (sc *)malloc(n * sizeof(sc)), *bs = (sc *)malloc(n * sizeof(sc)), *ch = (sc *)malloc(64 * sizeof(sc));
  aff *Gs = (aff *)malloc(n * sizeof(aff)), *Hs = (aff *)malloc(n * sizeof(aff)), q;
  int rc = load_scalars(gf, Gf, n) || load_scalars(hf, Hf, n) || load_scalars(as, a, n) ||
           load_scalars(bs, b, n) || load_points(Gs, G, n) || load_points(Hs, H, n) || aff_from_bytes(&q, Q);
  if (!rc) {
    transcript t;
    tr_init(&t, label, label_len);
    ipp_proof p;
    ipp_create(&p, &t, &q, gf, hf, Gs, Hs, as, bs, n, ch);
    store_points(L_out, p.L, p.k); store_points(R_out, p.R, p.k);
    fe_to_le(SC, a_out, &p.a); fe_to_le(SC, b_out, &p.b);
    if (challenges_out) store_scalars(challenges_out, ch, p.k);
    ipp_free(&p);
  }
  free(gf); free(hf); free(as); free(bs); free(ch); free(Gs); free(Hs);
  return rc ? -1 : 0;
}
API int bpo_ipp_verify(const uint8_t *label, size_t label_len, size_t n, const uint8_t *Gf,
                       const uint8_t *Hf, const uint8_t P[64], const uint8_t Q[64], const uint8_t *G,
                       const uint8_t *H, const uint8_t *L, const uint8_t *R, size_t k,
                       const uint8_t a[32], const uint8_t b[32]) {
  sc *gf = (sc *)malloc(n * sizeof(sc)), *hf = (sc *)malloc(n * sizeof(sc));
  aff *Gs = (aff *)malloc(n * sizeof(aff)), *Hs = (aff *)malloc(n * sizeof(aff)), p, q;
  ipp_proof pr;
  pr.k = k;
  pr.L = (aff *)malloc((k ? k : 1) * sizeof(aff));
  pr.R = (aff *)malloc((k ? k : 1) * sizeof(aff));
  int rc = load_scalars(gf, Gf, n) || load_scalars(hf, Hf, n) || load_points(Gs, G, n) ||
           load_points(Hs, H, n) || aff_from_bytes(&p, P) || aff_from_bytes(&q, Q) ||
           load_points(pr.L, L, k) || load_points(pr.R, R, k) || fe_from_le(SC, &pr.a, a) ||
           fe_from_le(SC, &pr.b, b);
  if (rc) rc = -3;
  else {
    transcript t;
    tr_init(&t, label, label_len);
    rc = ipp_verify(&pr, n, &t, gf, hf, &p, &q, Gs, Hs);
  }
  ipp_free(&pr);
  free(gf); free(hf); free(Gs); free(Hs);
  return rc;
}

/* ------------------------------------------------------------------ R1CS sessions */
enum { K_RANGE = 0, K_SHUFFLE = 1, K_EXAMPLE = 2, K_DUMMY = 3, K_RANGE_MULTI = 4 };   /* 4: param = n_bits | nvals << 16 */

static size_t proof_flat_size(size_t k) { return 8 + 11 * 64 + 3 * 32 + 2 * k * 64 + 64; }
API size_t bpo_proof_flat_size(size_t k) { return proof_flat_size(k); }

static void proof_to_flat(uint8_t *o, const r1cs_proof *p) {
  uint32_t k = (uint32_t)p->ipp.k;
  memset(o, 0, 8);
  for (int j = 0; j < 4; j++) o[j] = (uint8_t)(k >> (8 * j));
  o += 8;
  const aff *pts[11] = {&p->A_I1, &p->A_O1, &p->S1, &p->A_I2, &p->A_O2, &p->S2, &p->T_1, &p->T_3, &p->T_4, &p->T_5, &p->T_6};
  for (int i = 0; i < 11; i++, o += 64) aff_to_bytes(o, pts[i]);
  fe_to_le(SC, o, &p->t_x); o += 32;
  fe_to_le(SC, o, &p->t_x_blinding); o += 32;
  fe_to_le(SC, o, &p->e_blinding); o += 32;
  store_points(o, p->ipp.L, k); o += 64 * k;
  store_points(o, p->ipp.R, k); o += 64 * k;
  fe_to_le(SC, o, &p->ipp.a); o += 32;
  fe_to_le(SC, o, &p->ipp.b);
}
static int proof_from_flat(r1cs_proof *p, const uint8_t *o, size_t len) {
  memset(p, 0, sizeof *p);
  if (len < 8) return -1;
  uint32_t k = 0;
  for (int j = 0; j < 4; j++) k |= (uint32_t)o[j] << (8 * j);
  if (k >= 32 || len != proof_flat_size(k)) return -1;
  o += 8;
  aff *pts[11] = {&p->A_I1, &p->A_O1, &p->S1, &p->A_I2, &p->A_O2, &p->S2, &p->T_1, &p->T_3, &p->T_4, &p->T_5, &p->T_6};
  for (int i = 0; i < 11; i++, o += 64) if (aff_from_bytes(pts[i], o)) return -1;
  if (fe_from_le(SC, &p->t_x, o) || fe_from_le(SC, &p->t_x_blinding, o + 32) || fe_from_le(SC, &p->e_blinding, o + 64)) return -1;
  o += 96;
  p->ipp.k = k;
  p->ipp.L = (aff *)malloc((k ? k : 1) * sizeof(aff));
  p->ipp.R = (aff *)malloc((k ? k : 1) * sizeof(aff));
  if (load_points(p->ipp.L, o, k) || load_points(p->ipp.R, o + 64 * k, k)) { ipp_free(&p->ipp); return -1; }
  o += 128 * k;
  if (fe_from_le(SC, &p->ipp.a, o) || fe_from_le(SC, &p->ipp.b, o + 32)) { ipp_free(&p->ipp); return -1; }
  return 0;
}

static void start_transcript(transcript *t, int kind, size_t param, const uint8_t *label, size_t label_len) {
  tr_init(t, label, label_len);
  if (kind == K_SHUFFLE) { /* tests/r1cs.rs:80-81 */
    tr_append_message(t, "dom-sep", (const uint8_t *)"ShuffleProof", 12);
    tr_append_u64(t, "k", param);
  }
}

/* 1: provers made by bpo_r1cs_prove draw their blinding vectors as "BlindVec v1" keys (bpo.h splitmix.vec_keys) */
static int g_vec_keys = 0;
API void bpo_set_vector_keys(int on) { g_vec_keys = on != 0; }
API void bpo_blind_vector(const uint8_t key[32], int v, size_t count, uint8_t *out) {
  sc *s = (sc *)malloc((count ? count : 1) * sizeof(sc));
  blind_vector(s, key, v, count);
  for (size_t i = 0; i < count; i++) fe_to_le(SC, out + 32 * i, &s[i]);
  free(s);
}
/* Prove one gadget instance.  values: range [v]; shuffle inputs[k] then outputs[k];
 * example [a1,a2,b1,b2,c1,c2]; dummy [] (the public value is drawn from the seed).
 * The SplitMix64 stream `seed` first yields the commitment blinding factors in commit order
 * (as tests/r1cs.rs:86,91,251,684 draw them), then the prover's blinding factors. */
API int bpo_r1cs_prove(int kind, size_t param, const uint8_t *label, size_t label_len,
                       const uint64_t *values, size_t nvalues, uint64_t seed, size_t gens_capacity,
                       uint8_t *proof_out, size_t *proof_len, uint8_t *commitments_out, size_t *m_out) {
  transcript t;
  start_transcript(&t, kind, param, label, label_len);
  cs_t cs;
  cs_init(&cs, 1, &t);
  splitmix rng = {seed, g_vec_keys};
  sc v, bl;
  if (kind == K_RANGE) {
    if (nvalues != 1) return -3;
    fe_from_u64(SC, &v, values[0]); sm_scalar(&rng, &bl);
    var_t var = cs_commit_prover(&cs, &v, &bl, NULL);
    gadget_range_proof(&cs, var, 1, values[0], param);
  } else if (kind == K_RANGE_MULTI) {   /* several values range-proved in ONE constraint system (BASELINE config 3 shape) */
    size_t nb_ = param & 0xffff, nv = param >> 16;
    if (nvalues != nv) return -3;
    for (size_t i = 0; i < nv; i++) {
      fe_from_u64(SC, &v, values[i]); sm_scalar(&rng, &bl);
      var_t var = cs_commit_prover(&cs, &v, &bl, NULL);
      gadget_range_proof(&cs, var, 1, values[i], nb_);
    }
  } else if (kind == K_SHUFFLE) {
    size_t k = param;
    if (nvalues != 2 * k) return -3;
    var_t *vars = (var_t *)malloc(2 * k * sizeof(var_t));
    for (size_t i = 0; i < 2 * k; i++) {
      fe_from_u64(SC, &v, values[i]); sm_scalar(&rng, &bl);
      vars[i] = cs_commit_prover(&cs, &v, &bl, NULL);
    }
    gadget_shuffle(&cs, vars, vars + k, k);
    free(vars);
  } else if (kind == K_EXAMPLE) {
    if (nvalues != 6) return -3;
    var_t vars[5];
    for (int i = 0; i < 5; i++) {
      fe_from_u64(SC, &v, values[i]); sm_scalar(&rng, &bl);
      vars[i] = cs_commit_prover(&cs, &v, &bl, NULL);
    }
    gadget_example(&cs, vars, values[5]);
  } else if (kind == K_DUMMY) {
    sm_scalar(&rng, &v);
    gadget_dummy(&cs, &v, param);
  } else return -3;
  size_t cap = gens_capacity;
  aff *G = (aff *)malloc((cap ? cap : 1) * sizeof(aff)), *H = (aff *)malloc((cap ? cap : 1) * sizeof(aff));
  gens_chain(G, NULL, 'G', 0, cap);
  gens_chain(H, NULL, 'H', 0, cap);
  r1cs_proof p;
  int rc = cs_prove(&cs, G, H, cap, &rng, &p);
  if (rc == BPO_OK) {
    proof_to_flat(proof_out, &p);
    *proof_len = proof_flat_size(p.ipp.k);
    store_points(commitments_out, cs.V, cs.nv);
    *m_out = cs.nv;
  }
  r1cs_proof_free(&p);
  free(G); free(H);
  cs_free(&cs);
  return rc;
}

typedef struct {
  int rc;
  cs_t cs;
  transcript t;
  verify_trace tr;
  size_t nnz;
} vsession;

/* Verify one gadget instance and keep everything the GPU parity tests need.
 * values: example [c2]; others unused.  commitments: m x 64 bytes. */
API void *bpo_verify_open(int kind, size_t param, const uint8_t *label, size_t label_len,
                          const uint64_t *values, size_t nvalues, const uint8_t *commitments,
                          size_t m, const uint8_t *proof, size_t proof_len, size_t gens_capacity) {
  vsession *s = (vsession *)calloc(1, sizeof *s);
  start_transcript(&s->t, kind, param, label, label_len);
  cs_init(&s->cs, 0, &s->t);
  aff *V = (aff *)malloc((m ? m : 1) * sizeof(aff));
  r1cs_proof p;
  memset(&p, 0, sizeof p);
  s->rc = -3;
  if (load_points(V, commitments, m) == 0 && proof_from_flat(&p, proof, proof_len) == 0) {
    var_t *vars = (var_t *)malloc((m ? m : 1) * sizeof(var_t));
    for (size_t i = 0; i < m; i++) vars[i] = cs_commit_verifier(&s->cs, &V[i]);
    int ok = 1;
    if (kind == K_RANGE && m == 1) gadget_range_proof(&s->cs, vars[0], 0, 0, param);
    else if (kind == K_RANGE_MULTI && m == (param >> 16)) { for (size_t i = 0; i < m; i++) gadget_range_proof(&s->cs, vars[i], 0, 0, param & 0xffff); }
    else if (kind == K_SHUFFLE && m == 2 * param) gadget_shuffle(&s->cs, vars, vars + param, param);
    else if (kind == K_EXAMPLE && m == 5 && nvalues == 1) gadget_example(&s->cs, vars, values[0]);
    else if (kind == K_DUMMY && m == 1) {
      /* benches/r1cs.rs:24-33 with the prover's commitment reused as the public input */
      var_t var = vars[0], out[3];
      for (size_t i = 0; i < param; i++) {
        lincomb l, r;
        lc_init(&l); lc_add_term_i64(&l, var, 1);
        lc_init(&r); lc_add_term_i64(&r, var, 1);
        cs_multiply(&s->cs, &l, &r, out);
        var = out[2];
      }
    } else ok = 0;
    free(vars);
    if (ok) {
      size_t cap = gens_capacity;
      aff *G = (aff *)malloc((cap ? cap : 1) * sizeof(aff)), *H = (aff *)malloc((cap ? cap : 1) * sizeof(aff));
      gens_chain(G, NULL, 'G', 0, cap);
      gens_chain(H, NULL, 'H', 0, cap);
      s->rc = cs_verify(&s->cs, &p, G, H, cap, &s->tr);
      free(G); free(H);
      for (size_t r = 0; r < s->cs.nc; r++) s->nnz += s->cs.constraints[r].n;
    }
  }
  r1cs_proof_free(&p);
  free(V);
  return s;
}
API int bpo_verify_rc(void *h) { return ((vsession *)h)->rc; }
/* out: n1, n2, padded_n, k, m, nterms, q (constraints), nnz */
API void bpo_verify_dims(void *h, uint64_t out[8]) {
  vsession *s = (vsession *)h;
  out[0] = s->tr.n1; out[1] = s->tr.n2; out[2] = s->tr.padded_n; out[3] = s->tr.k; out[4] = s->tr.m;
  out[5] = s->tr.nterms; out[6] = s->cs.nc; out[7] = s->nnz;
}
/* (6 + k) x 32: y, z, u, x, w, r, u_1..u_k  (only valid if the trace was produced) */
API void bpo_verify_challenges(void *h, uint8_t *out) {
  vsession *s = (vsession *)h;
  if (!s->tr.scalars) return;
  const sc *c[6] = {&s->tr.y, &s->tr.z, &s->tr.u, &s->tr.x, &s->tr.w, &s->tr.r};
  for (int i = 0; i < 6; i++) fe_to_le(SC, out + 32 * i, c[i]);
  store_scalars(out + 192, s->tr.ipp_u, s->tr.k);
}
API void bpo_verify_msm(void *h, uint8_t *scalars, uint8_t *points) {
  vsession *s = (vsession *)h;
  if (!s->tr.scalars) return;
  store_scalars(scalars, s->tr.scalars, s->tr.nterms);
  store_points(points, s->tr.points, s->tr.nterms);
}
API void bpo_verify_mega(void *h, uint8_t out[64]) {
  vsession *s = (vsession *)h;
  aff_to_bytes(out, &s->tr.mega_check);
}
/* CSR of the constraint rows: row_ptr[q+1], kind[nnz] (VAR_*), idx[nnz], coeff[nnz*32] */
API void bpo_verify_csr(void *h, uint32_t *row_ptr, uint32_t *kind, uint32_t *idx, uint8_t *coeff) {
  vsession *s = (vsession *)h;
  size_t o = 0;
  for (size_t r = 0; r < s->cs.nc; r++) {
    row_ptr[r] = (uint32_t)o;
    const lincomb *l = &s->cs.constraints[r];
    for (size_t i = 0; i < l->n; i++, o++) {
      kind[o] = l->t[i].var.kind;
      idx[o] = l->t[i].var.idx;
      fe_to_le(SC, coeff + 32 * o, &l->t[i].coeff);
    }
  }
  row_ptr[s->cs.nc] = (uint32_t)o;
}
API void bpo_verify_flatten(void *h, const uint8_t z[32], uint8_t *wL, uint8_t *wR, uint8_t *wO,
                            uint8_t *wV, uint8_t wc[32]) {
  vsession *s = (vsession *)h;
  size_t n = s->cs.num_vars, m = s->cs.nv;
  sc zs, c, *a = (sc *)malloc((n ? n : 1) * sizeof(sc)), *b = (sc *)malloc((n ? n : 1) * sizeof(sc));
  sc *o = (sc *)malloc((n ? n : 1) * sizeof(sc)), *v = (sc *)malloc((m ? m : 1) * sizeof(sc));
  fe_from_le(SC, &zs, z);
  cs_flattened_constraints(&s->cs, &zs, a, b, o, v, &c);
  store_scalars(wL, a, n); store_scalars(wR, b, n); store_scalars(wO, o, n); store_scalars(wV, v, m);
  fe_to_le(SC, wc, &c);
  free(a); free(b); free(o); free(v);
}
API void bpo_verify_close(void *h) {
  vsession *s = (vsession *)h;
  verify_trace_free(&s->tr);
  cs_free(&s->cs);
  free(s);
}
/* convenience: 0 = Ok, -1 = VerificationError, -2 = InvalidGeneratorsLength, -3 = malformed */
API int bpo_r1cs_verify(int kind, size_t param, const uint8_t *label, size_t label_len,
                        const uint64_t *values, size_t nvalues, const uint8_t *commitments, size_t m,
                        const uint8_t *proof, size_t proof_len, size_t gens_capacity) {
  void *h = bpo_verify_open(kind, param, label, label_len, values, nvalues, commitments, m, proof, proof_len, gens_capacity);
  int rc = bpo_verify_rc(h);
  bpo_verify_close(h);
  return rc;
}

/* Verifier::verify over nb proofs of one gadget with the generators built once (what a host that
 * keeps BulletproofGens around does; benches/r1cs.rs:69).  proofs: nb x proof_len, commitments:
 * nb x m x 64.  ok[i] = 1 accept / 0 reject.  Used as the timed CPU baseline ("port"). */
API int bpo_r1cs_verify_many(int kind, size_t param, const uint8_t *label, size_t label_len,
                             const uint64_t *values, size_t nvalues, const uint8_t *commitments, size_t m,
                             const uint8_t *proofs, size_t proof_len, size_t nb, size_t gens_capacity,
                             int32_t *ok) {
  size_t cap = gens_capacity;
  aff *G = (aff *)malloc((cap ? cap : 1) * sizeof(aff)), *H = (aff *)malloc((cap ? cap : 1) * sizeof(aff));
  gens_chain(G, NULL, 'G', 0, cap);
  gens_chain(H, NULL, 'H', 0, cap);
  aff *V = (aff *)malloc((m ? m : 1) * sizeof(aff));
  var_t *vars = (var_t *)malloc((m ? m : 1) * sizeof(var_t));
  int rc_all = 0;
  for (size_t i = 0; i < nb; i++) {
    ok[i] = 0;
    transcript t;
    start_transcript(&t, kind, param, label, label_len);
    cs_t cs;
    cs_init(&cs, 0, &t);
    r1cs_proof p;
    memset(&p, 0, sizeof p);
    if (load_points(V, commitments + i * m * 64, m) == 0 && proof_from_flat(&p, proofs + i * proof_len, proof_len) == 0) {
      for (size_t j = 0; j < m; j++) vars[j] = cs_commit_verifier(&cs, &V[j]);
      int good = 1;
      if (kind == K_RANGE && m == 1) gadget_range_proof(&cs, vars[0], 0, 0, param);
      else if (kind == K_RANGE_MULTI && m == (param >> 16)) { for (size_t j = 0; j < m; j++) gadget_range_proof(&cs, vars[j], 0, 0, param & 0xffff); }
      else if (kind == K_SHUFFLE && m == 2 * param) gadget_shuffle(&cs, vars, vars + param, param);
      else if (kind == K_EXAMPLE && m == 5 && nvalues == 1) gadget_example(&cs, vars, values[0]);
      else good = 0;
      if (good) ok[i] = cs_verify(&cs, &p, G, H, cap, NULL) == BPO_OK;
      else rc_all = -3;
    } else rc_all = -3;
    r1cs_proof_free(&p);
    cs_free(&cs);
  }
  free(G); free(H); free(V); free(vars);
  return rc_all;
}
