/* bpo_ipp.c -- CPU oracle: inner-product argument, restating src/inner_product_proof.rs.
 * TEST INFRASTRUCTURE ONLY (see bpo.h). */
#include "bpo.h"
#include <stdlib.h>
#include <string.h>

void ipp_free(ipp_proof *p) {
  free(p->L);
  free(p->R);
  p->L = p->R = NULL;
  p->k = 0;
}

/* StarkPoint::msm(&[s0, s1], &[P0, P1]) as called at inner_product_proof.rs:226-227 */
static void msm2(aff *out, const sc *s0, const aff *p0, const sc *s1, const aff *p1) {
  sc s[2] = {*s0, *s1};
  aff p[2] = {*p0, *p1};
  jac r;
  msm_naive(&r, s, p, 2);
  jac_to_aff(out, &r);
}

/* inner_product_proof.rs:202-248 (the serial and the rayon branch compute the same values) */
void fold_witness(const sc *u, const sc *u_inv, size_t n,
                  const sc *a_L, const sc *a_R, const sc *b_L, const sc *b_R,
                  const aff *G_L, const aff *G_R, const aff *H_L, const aff *H_R,
                  sc *a_out, sc *b_out, aff *G_out, aff *H_out) {
  for (size_t i = 0; i < n; i++) {
    sc t1, t2;
    fe_mul(SC, &t1, &a_L[i], u); fe_mul(SC, &t2, u_inv, &a_R[i]); fe_add(SC, &a_out[i], &t1, &t2);
    fe_mul(SC, &t1, &b_L[i], u_inv); fe_mul(SC, &t2, u, &b_R[i]); fe_add(SC, &b_out[i], &t1, &t2);
    aff g, h;
    msm2(&g, u_inv, &G_L[i], u, &G_R[i]);
    msm2(&h, u, &H_L[i], u_inv, &H_R[i]);
    G_out[i] = g;
    H_out[i] = h;
  }
}

/* inner_product_proof.rs:49-193 */
void ipp_create(ipp_proof *out, transcript *t, const aff *Q, const sc *G_factors,
                const sc *H_factors, aff *G, aff *H, sc *a, sc *b, size_t n, sc *challenges_out) {
  size_t lg_n = 0;
  while (((size_t)1 << lg_n) < n) lg_n++;
  out->k = lg_n;
  out->L = (aff *)calloc(lg_n ? lg_n : 1, sizeof(aff));
  out->R = (aff *)calloc(lg_n ? lg_n : 1, sizeof(aff));
  tr_innerproduct_domain_sep(t, n);

  sc *ms = (sc *)malloc((2 * n + 1) * sizeof(sc));
  aff *mp = (aff *)malloc((2 * n + 1) * sizeof(aff));
  sc *a2 = (sc *)malloc(n * sizeof(sc)), *b2 = (sc *)malloc(n * sizeof(sc));
  aff *G2 = (aff *)malloc(n * sizeof(aff)), *H2 = (aff *)malloc(n * sizeof(aff));
  int first = 1;
  size_t round = 0;
  while (n != 1) {
    n /= 2;
    sc *a_L = a, *a_R = a + n, *b_L = b, *b_R = b + n;
    aff *G_L = G, *G_R = G + n, *H_L = H, *H_R = H + n;
    sc c_L, c_R;
    sc_inner_product(&c_L, a_L, b_R, n);
    sc_inner_product(&c_R, a_R, b_L, n);
    jac Lj, Rj;
    /* L: scalars a_L (.G_factors[n..2n]) || b_R (.H_factors[0..n]) || c_L ; points G_R || H_L || Q */
    for (size_t i = 0; i < n; i++) {
      if (first) {
        fe_mul(SC, &ms[i], &a_L[i], &G_factors[n + i]);
        fe_mul(SC, &ms[n + i], &b_R[i], &H_factors[i]);
      } else {
        ms[i] = a_L[i];
        ms[n + i] = b_R[i];
      }
      mp[i] = G_R[i];
      mp[n + i] = H_L[i];
    }
    ms[2 * n] = c_L;
    mp[2 * n] = *Q;
    msm(&Lj, ms, mp, 2 * n + 1);
    for (size_t i = 0; i < n; i++) {
      if (first) {
        fe_mul(SC, &ms[i], &a_R[i], &G_factors[i]);
        fe_mul(SC, &ms[n + i], &b_L[i], &H_factors[n + i]);
      } else {
        ms[i] = a_R[i];
        ms[n + i] = b_L[i];
      }
      mp[i] = G_L[i];
      mp[n + i] = H_R[i];
    }
    ms[2 * n] = c_R;
    msm(&Rj, ms, mp, 2 * n + 1);
    jac_to_aff(&out->L[round], &Lj);
    jac_to_aff(&out->R[round], &Rj);
    tr_append_point(t, "L", &out->L[round]);
    tr_append_point(t, "R", &out->R[round]);
    sc u, u_inv;
    tr_challenge_scalar(t, "u", &u);
    fe_inv(SC, &u_inv, &u);
    if (challenges_out) challenges_out[round] = u;
    if (first) {
      /* inner_product_proof.rs:125-134: scale every generator by its factor */
      jac pj, rj;
      for (size_t i = 0; i < 2 * n; i++) {
        jac_from_aff(&pj, &G[i]); jac_mul(&rj, &pj, &G_factors[i]); jac_to_aff(&G[i], &rj);
        jac_from_aff(&pj, &H[i]); jac_mul(&rj, &pj, &H_factors[i]); jac_to_aff(&H[i], &rj);
      }
      first = 0;
    }
    fold_witness(&u, &u_inv, n, a_L, a_R, b_L, b_R, G_L, G_R, H_L, H_R, a2, b2, G2, H2);
    memcpy(a, a2, n * sizeof(sc));
    memcpy(b, b2, n * sizeof(sc));
    memcpy(G, G2, n * sizeof(aff));
    memcpy(H, H2, n * sizeof(aff));
    round++;
  }
  out->a = a[0];
  out->b = b[0];
  free(ms); free(mp); free(a2); free(b2); free(G2); free(H2);
}

/* inner_product_proof.rs:259-278 */
int ipp_challenges(const ipp_proof *p, size_t n, transcript *t, sc *challenges) {
  size_t lg_n = p->k;
  if (lg_n >= 32) return BPO_ERR_VERIFICATION;
  if (n != ((size_t)1 << lg_n)) return BPO_ERR_VERIFICATION;
  tr_innerproduct_domain_sep(t, n);
  for (size_t i = 0; i < lg_n; i++) {
    if (tr_validate_and_append_point(t, "L", &p->L[i])) return BPO_ERR_VERIFICATION;
    if (tr_validate_and_append_point(t, "R", &p->R[i])) return BPO_ERR_VERIFICATION;
    tr_challenge_scalar(t, "u", &challenges[i]);
  }
  return BPO_OK;
}

/* inner_product_proof.rs:280-309 */
void verification_scalars(const sc *challenges, size_t k, size_t n, sc *u_sq, sc *u_inv_sq, sc *s) {
  sc *inv = (sc *)malloc((k ? k : 1) * sizeof(sc));
  memcpy(inv, challenges, k * sizeof(sc));
  sc_batch_inverse(inv, k);
  sc allinv = SC->one;
  for (size_t i = 0; i < k; i++) fe_mul(SC, &allinv, &allinv, &inv[i]);
  for (size_t i = 0; i < k; i++) {
    fe_mul(SC, &u_sq[i], &challenges[i], &challenges[i]);
    fe_mul(SC, &u_inv_sq[i], &inv[i], &inv[i]);
  }
  s[0] = allinv;
  for (size_t i = 1; i < n; i++) {
    size_t lg_i = 0;
    while ((i >> (lg_i + 1)) != 0) lg_i++;
    size_t kk = (size_t)1 << lg_i;
    fe_mul(SC, &s[i], &s[i - kk], &u_sq[(k - 1) - lg_i]);
  }
  free(inv);
}

/* inner_product_proof.rs:317-372 */
int ipp_verify(const ipp_proof *p, size_t n, transcript *t, const sc *G_factors,
               const sc *H_factors, const aff *P, const aff *Q, const aff *G, const aff *H) {
  size_t k = p->k;
  sc *ch = (sc *)malloc((k ? k : 1) * sizeof(sc));
  int rc = ipp_challenges(p, n, t, ch);
  if (rc) { free(ch); return rc; }
  sc *u_sq = (sc *)malloc((k ? k : 1) * sizeof(sc)), *u_inv_sq = (sc *)malloc((k ? k : 1) * sizeof(sc));
  sc *s = (sc *)malloc(n * sizeof(sc));
  verification_scalars(ch, k, n, u_sq, u_inv_sq, s);
  size_t nt = 1 + 2 * n + 2 * k;
  sc *ms = (sc *)malloc(nt * sizeof(sc));
  aff *mp = (aff *)malloc(nt * sizeof(aff));
  fe_mul(SC, &ms[0], &p->a, &p->b);
  mp[0] = *Q;
  for (size_t i = 0; i < n; i++) {
    sc tt;
    fe_mul(SC, &tt, &p->a, &s[i]); fe_mul(SC, &ms[1 + i], &tt, &G_factors[i]);
    fe_mul(SC, &tt, &p->b, &s[n - 1 - i]); fe_mul(SC, &ms[1 + n + i], &tt, &H_factors[i]);
    mp[1 + i] = G[i];
    mp[1 + n + i] = H[i];
  }
  for (size_t i = 0; i < k; i++) {
    fe_neg(SC, &ms[1 + 2 * n + i], &u_sq[i]);
    fe_neg(SC, &ms[1 + 2 * n + k + i], &u_inv_sq[i]);
    mp[1 + 2 * n + i] = p->L[i];
    mp[1 + 2 * n + k + i] = p->R[i];
  }
  jac e, pj;
  msm(&e, ms, mp, nt);
  jac_from_aff(&pj, P);
  rc = jac_eq(&e, &pj) ? BPO_OK : BPO_ERR_VERIFICATION;
  free(ch); free(u_sq); free(u_inv_sq); free(s); free(ms); free(mp);
  return rc;
}
