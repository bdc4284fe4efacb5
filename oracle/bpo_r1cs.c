/* bpo_r1cs.c -- CPU oracle: R1CS prover / verifier, restating src/r1cs/prover.rs and
 * src/r1cs/verifier.rs, plus the gadgets of tests/r1cs.rs and benches/r1cs.rs.
 * TEST INFRASTRUCTURE ONLY (see bpo.h). */
#include "bpo.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- linear combinations */
void lc_init(lincomb *l) { l->t = NULL; l->n = l->cap = 0; }
void lc_free(lincomb *l) { free(l->t); lc_init(l); }
void lc_copy(lincomb *d, const lincomb *s) {
  d->n = d->cap = s->n;
  d->t = (term_t *)malloc((s->n ? s->n : 1) * sizeof(term_t));
  memcpy(d->t, s->t, s->n * sizeof(term_t));
}
/* linear_combination.rs:129-135: coefficients of an existing key are summed */
void lc_add_term(lincomb *l, var_t v, const sc *coeff) {
  for (size_t i = 0; i < l->n; i++)
    if (l->t[i].var.kind == v.kind && (v.kind == VAR_ONE || l->t[i].var.idx == v.idx)) {
      fe_add(SC, &l->t[i].coeff, &l->t[i].coeff, coeff);
      return;
    }
  if (l->n == l->cap) {
    l->cap = l->cap ? 2 * l->cap : 4;
    l->t = (term_t *)realloc(l->t, l->cap * sizeof(term_t));
  }
  l->t[l->n].var = v;
  l->t[l->n].coeff = *coeff;
  l->n++;
}
void lc_add_term_i64(lincomb *l, var_t v, int64_t c) {
  sc s;
  fe_from_u64(SC, &s, (uint64_t)(c < 0 ? -c : c));
  if (c < 0) fe_neg(SC, &s, &s);
  lc_add_term(l, v, &s);
}

/* ---------------------------------------------------------------- constraint system */
void cs_init(cs_t *cs, int is_prover, transcript *tr) {
  memset(cs, 0, sizeof *cs);
  cs->is_prover = is_prover;
  cs->tr = tr;
  cs->B = BPO_G;          /* generators.rs:61-70: B = B_blinding = generator */
  cs->B_blinding = BPO_G;
  cs->pending_multiplier = -1;
  tr_r1cs_domain_sep(tr); /* prover.rs:286 / verifier.rs:271 */
}
void cs_free(cs_t *cs) {
  for (size_t i = 0; i < cs->nc; i++) lc_free(&cs->constraints[i]);
  free(cs->constraints);
  free(cs->a_L); free(cs->a_R); free(cs->a_O);
  free(cs->v); free(cs->v_blinding); free(cs->V);
  memset(cs, 0, sizeof *cs);
}
static void commit(aff *out, const cs_t *cs, const sc *v, const sc *r) { /* generators.rs:41-43 */
  sc s[2] = {*v, *r};
  aff p[2] = {cs->B, cs->B_blinding};
  jac j;
  msm_naive(&j, s, p, 2);
  jac_to_aff(out, &j);
}
static void grow_v(cs_t *cs) {
  if (cs->nv == cs->cap_v) {
    cs->cap_v = cs->cap_v ? 2 * cs->cap_v : 8;
    cs->v = (sc *)realloc(cs->v, cs->cap_v * sizeof(sc));
    cs->v_blinding = (sc *)realloc(cs->v_blinding, cs->cap_v * sizeof(sc));
    cs->V = (aff *)realloc(cs->V, cs->cap_v * sizeof(aff));
  }
}
/* prover.rs:319-329 */
var_t cs_commit_prover(cs_t *cs, const sc *v, const sc *v_blinding, aff *V_out) {
  grow_v(cs);
  size_t i = cs->nv++;
  cs->v[i] = *v;
  cs->v_blinding[i] = *v_blinding;
  commit(&cs->V[i], cs, v, v_blinding);
  tr_append_point(cs->tr, "V", &cs->V[i]);
  if (V_out) *V_out = cs->V[i];
  return (var_t){VAR_V, (uint32_t)i};
}
/* verifier.rs:298-306 */
var_t cs_commit_verifier(cs_t *cs, const aff *V) {
  grow_v(cs);
  size_t i = cs->nv++;
  cs->V[i] = *V;
  tr_append_point(cs->tr, "V", V);
  return (var_t){VAR_V, (uint32_t)i};
}
/* prover.rs:171-173 / verifier.rs:153-160: blinding factor one */
var_t cs_commit_public(cs_t *cs, const sc *v) {
  if (cs->is_prover) return cs_commit_prover(cs, v, &SC->one, NULL);
  aff V;
  commit(&V, cs, v, &SC->one);
  return cs_commit_verifier(cs, &V);
}
void cs_constrain(cs_t *cs, lincomb *l) {
  if (cs->nc == cs->cap_c) {
    cs->cap_c = cs->cap_c ? 2 * cs->cap_c : 16;
    cs->constraints = (lincomb *)realloc(cs->constraints, cs->cap_c * sizeof(lincomb));
  }
  cs->constraints[cs->nc++] = *l;
  lc_init(l);
}
size_t cs_num_multipliers(const cs_t *cs) { return cs->is_prover ? cs->nmul : cs->num_vars; }
/* prover.rs:179-194 (verifier.rs:168-174 returns zero) */
void cs_eval(const cs_t *cs, const lincomb *l, sc *out) {
  sc acc, t;
  memset(&acc, 0, sizeof acc);
  if (cs->is_prover)
    for (size_t i = 0; i < l->n; i++) {
      const sc *val;
      switch (l->t[i].var.kind) {
        case VAR_L: val = &cs->a_L[l->t[i].var.idx]; break;
        case VAR_R: val = &cs->a_R[l->t[i].var.idx]; break;
        case VAR_O: val = &cs->a_O[l->t[i].var.idx]; break;
        case VAR_V: val = &cs->v[l->t[i].var.idx]; break;
        default: val = &SC->one; break;
      }
      fe_mul(SC, &t, &l->t[i].coeff, val);
      fe_add(SC, &acc, &acc, &t);
    }
  *out = acc;
}
static size_t push_mul(cs_t *cs, const sc *l, const sc *r, const sc *o) {
  if (cs->nmul == cs->cap_mul) {
    cs->cap_mul = cs->cap_mul ? 2 * cs->cap_mul : 16;
    cs->a_L = (sc *)realloc(cs->a_L, cs->cap_mul * sizeof(sc));
    cs->a_R = (sc *)realloc(cs->a_R, cs->cap_mul * sizeof(sc));
    cs->a_O = (sc *)realloc(cs->a_O, cs->cap_mul * sizeof(sc));
  }
  size_t i = cs->nmul++;
  cs->a_L[i] = *l; cs->a_R[i] = *r; cs->a_O[i] = *o;
  return i;
}
/* prover.rs:99-125 / verifier.rs:99-120 */
void cs_multiply(cs_t *cs, lincomb *left, lincomb *right, var_t out[3]) {
  size_t i;
  if (cs->is_prover) {
    sc l, r, o;
    cs_eval(cs, left, &l);
    cs_eval(cs, right, &r);
    fe_mul(SC, &o, &l, &r);
    i = push_mul(cs, &l, &r, &o);
  } else {
    i = cs->num_vars++;
  }
  out[0] = (var_t){VAR_L, (uint32_t)i};
  out[1] = (var_t){VAR_R, (uint32_t)i};
  out[2] = (var_t){VAR_O, (uint32_t)i};
  lc_add_term_i64(left, out[0], -1);
  lc_add_term_i64(right, out[1], -1);
  cs_constrain(cs, left);
  cs_constrain(cs, right);
}
/* prover.rs:148-165 / verifier.rs:137-151 */
void cs_allocate_multiplier(cs_t *cs, const sc *l, const sc *r, var_t out[3]) {
  size_t i;
  if (cs->is_prover) {
    sc o;
    fe_mul(SC, &o, l, r);
    i = push_mul(cs, l, r, &o);
  } else {
    i = cs->num_vars++;
  }
  out[0] = (var_t){VAR_L, (uint32_t)i};
  out[1] = (var_t){VAR_R, (uint32_t)i};
  out[2] = (var_t){VAR_O, (uint32_t)i};
}
void cs_specify_randomized_constraints(cs_t *cs, cs_callback cb, void *ctx) {
  cs->cb[cs->ncb] = cb;
  cs->cb_ctx[cs->ncb] = ctx;
  cs->ncb++;
}
void cs_challenge_scalar(cs_t *cs, const char *label, sc *out) { tr_challenge_scalar(cs->tr, label, out); }

/* prover.rs:383-402 / verifier.rs:366-385 */
static int create_randomized_constraints(cs_t *cs) {
  cs->pending_multiplier = -1;
  if (cs->ncb == 0) {
    tr_r1cs_1phase_domain_sep(cs->tr);
    return 0;
  }
  tr_r1cs_2phase_domain_sep(cs->tr);
  size_t n = cs->ncb;
  cs->ncb = 0;
  for (size_t i = 0; i < n; i++) {
    int rc = cs->cb[i](cs, cs->cb_ctx[i]);
    if (rc) return rc;
  }
  return 0;
}

/* prover.rs:342-379 / verifier.rs:323-362 */
void cs_flattened_constraints(const cs_t *cs, const sc *z, sc *wL, sc *wR, sc *wO, sc *wV, sc *wc) {
  size_t n = cs_num_multipliers(cs), m = cs->nv;
  memset(wL, 0, n * sizeof(sc)); memset(wR, 0, n * sizeof(sc)); memset(wO, 0, n * sizeof(sc));
  memset(wV, 0, m * sizeof(sc));
  if (wc) memset(wc, 0, sizeof(sc));
  sc exp_z = *z, t;
  for (size_t r = 0; r < cs->nc; r++) {
    const lincomb *l = &cs->constraints[r];
    for (size_t i = 0; i < l->n; i++) {
      fe_mul(SC, &t, &exp_z, &l->t[i].coeff);
      uint32_t ix = l->t[i].var.idx;
      switch (l->t[i].var.kind) {
        case VAR_L: fe_add(SC, &wL[ix], &wL[ix], &t); break;
        case VAR_R: fe_add(SC, &wR[ix], &wR[ix], &t); break;
        case VAR_O: fe_add(SC, &wO[ix], &wO[ix], &t); break;
        case VAR_V: fe_sub(SC, &wV[ix], &wV[ix], &t); break;
        case VAR_ONE: if (wc) fe_sub(SC, wc, wc, &t); break;
      }
    }
    fe_mul(SC, &exp_z, &exp_z, z);
  }
}

void r1cs_proof_free(r1cs_proof *p) { ipp_free(&p->ipp); }
void verify_trace_free(verify_trace *t) { free(t->ipp_u); free(t->scalars); free(t->points); memset(t, 0, sizeof *t); }

static size_t next_pow2(size_t n) {
  size_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
static void set_inf(aff *a) { memset(a, 0, sizeof *a); a->inf = 1; }

/* s_L then s_R, `count` scalars each (prover.rs:461-462 / :526-527): scalar by scalar from the stream, or -- rng->vec_keys --
 * one 32-byte key from the stream expanded by blind_vector (nothing is drawn when count == 0, as a caller of the device path
 * draws no key for an empty phase) */
static void draw_blinding_vectors(splitmix *rng, sc *s_L, sc *s_R, size_t count) {
  if (!rng->vec_keys) {
    for (size_t i = 0; i < count; i++) sm_scalar(rng, &s_L[i]);
    for (size_t i = 0; i < count; i++) sm_scalar(rng, &s_R[i]);
    return;
  }
  if (!count) return;
  uint8_t key[32];
  for (int i = 0; i < 4; i++) {
    uint64_t w = sm_next(rng);
    for (int b = 0; b < 8; b++) key[8 * i + b] = (uint8_t)(w >> (8 * b));
  }
  blind_vector(s_L, key, 0, count);
  blind_vector(s_R, key, 1, count);
}
/* prover.rs:412-727 with the RNG injected */
int cs_prove(cs_t *cs, const aff *G, const aff *H, size_t gens_capacity, splitmix *rng, r1cs_proof *out) {
  transcript *tr = cs->tr;
  memset(out, 0, sizeof *out);
  tr_append_u64(tr, "m", cs->nv);
  size_t n1 = cs->nmul;
  if (gens_capacity < n1) return BPO_ERR_GENS;
  sc i_b1, o_b1, s_b1;
  sm_scalar(rng, &i_b1); sm_scalar(rng, &o_b1); sm_scalar(rng, &s_b1);
  sc *s_L = (sc *)malloc((n1 ? n1 : 1) * sizeof(sc)), *s_R = (sc *)malloc((n1 ? n1 : 1) * sizeof(sc));
  draw_blinding_vectors(rng, s_L, s_R, n1);
  /* prover.rs:465-494 */
  {
    size_t nt = 2 * n1 + 1;
    sc *ms = (sc *)malloc(nt * sizeof(sc));
    aff *mp = (aff *)malloc(nt * sizeof(aff));
    jac r;
    ms[0] = i_b1; mp[0] = cs->B_blinding;
    for (size_t i = 0; i < n1; i++) { ms[1 + i] = cs->a_L[i]; mp[1 + i] = G[i]; ms[1 + n1 + i] = cs->a_R[i]; mp[1 + n1 + i] = H[i]; }
    msm(&r, ms, mp, nt); jac_to_aff(&out->A_I1, &r);
    ms[0] = o_b1;
    for (size_t i = 0; i < n1; i++) ms[1 + i] = cs->a_O[i];
    msm(&r, ms, mp, n1 + 1); jac_to_aff(&out->A_O1, &r);
    ms[0] = s_b1;
    for (size_t i = 0; i < n1; i++) { ms[1 + i] = s_L[i]; ms[1 + n1 + i] = s_R[i]; }
    msm(&r, ms, mp, nt); jac_to_aff(&out->S1, &r);
    free(ms); free(mp);
  }
  tr_append_point(tr, "A_I1", &out->A_I1);
  tr_append_point(tr, "A_O1", &out->A_O1);
  tr_append_point(tr, "S1", &out->S1);
  int rc = create_randomized_constraints(cs);
  if (rc) { free(s_L); free(s_R); return rc; }
  size_t n = cs->nmul, n2 = n - n1, padded_n = next_pow2(n), pad = padded_n - n;
  if (gens_capacity < padded_n) { free(s_L); free(s_R); return BPO_ERR_GENS; }
  sc i_b2, o_b2, s_b2;
  memset(&i_b2, 0, sizeof(sc)); memset(&o_b2, 0, sizeof(sc)); memset(&s_b2, 0, sizeof(sc));
  if (n2 > 0) { sm_scalar(rng, &i_b2); sm_scalar(rng, &o_b2); sm_scalar(rng, &s_b2); }
  s_L = (sc *)realloc(s_L, (n ? n : 1) * sizeof(sc));
  s_R = (sc *)realloc(s_R, (n ? n : 1) * sizeof(sc));
  draw_blinding_vectors(rng, s_L + n1, s_R + n1, n - n1);
  if (n2 > 0) { /* prover.rs:532-565 */
    size_t nt = 2 * n2 + 1;
    sc *ms = (sc *)malloc(nt * sizeof(sc));
    aff *mp = (aff *)malloc(nt * sizeof(aff));
    jac r;
    ms[0] = i_b2; mp[0] = cs->B_blinding;
    for (size_t i = 0; i < n2; i++) { ms[1 + i] = cs->a_L[n1 + i]; mp[1 + i] = G[n1 + i]; ms[1 + n2 + i] = cs->a_R[n1 + i]; mp[1 + n2 + i] = H[n1 + i]; }
    msm(&r, ms, mp, nt); jac_to_aff(&out->A_I2, &r);
    ms[0] = o_b2;
    for (size_t i = 0; i < n2; i++) ms[1 + i] = cs->a_O[n1 + i];
    msm(&r, ms, mp, n2 + 1); jac_to_aff(&out->A_O2, &r);
    ms[0] = s_b2;
    for (size_t i = 0; i < n2; i++) { ms[1 + i] = s_L[n1 + i]; ms[1 + n2 + i] = s_R[n1 + i]; }
    msm(&r, ms, mp, nt); jac_to_aff(&out->S2, &r);
    free(ms); free(mp);
  } else {
    set_inf(&out->A_I2); set_inf(&out->A_O2); set_inf(&out->S2); /* prover.rs:566-576 */
  }
  tr_append_point(tr, "A_I2", &out->A_I2);
  tr_append_point(tr, "A_O2", &out->A_O2);
  tr_append_point(tr, "S2", &out->S2);
  sc y, z;
  tr_challenge_scalar(tr, "y", &y);
  tr_challenge_scalar(tr, "z", &z);
  size_t m = cs->nv, na = n ? n : 1;
  sc *wL = (sc *)malloc(na * sizeof(sc)), *wR = (sc *)malloc(na * sizeof(sc)), *wO = (sc *)malloc(na * sizeof(sc));
  sc *wV = (sc *)malloc((m ? m : 1) * sizeof(sc));
  cs_flattened_constraints(cs, &z, wL, wR, wO, wV, NULL);
  sc y_inv;
  fe_inv(SC, &y_inv, &y);
  sc *exp_y_inv = (sc *)malloc(padded_n * sizeof(sc));
  exp_iter(exp_y_inv, &y_inv, padded_n);
  /* prover.rs:596-617 */
  sc *l1 = (sc *)malloc(na * sizeof(sc)), *l2 = (sc *)malloc(na * sizeof(sc)), *l3 = (sc *)malloc(na * sizeof(sc));
  sc *r0 = (sc *)malloc(na * sizeof(sc)), *r1 = (sc *)malloc(na * sizeof(sc)), *r3 = (sc *)malloc(na * sizeof(sc));
  sc exp_y = SC->one, t;
  for (size_t i = 0; i < n; i++) {
    fe_mul(SC, &t, &exp_y_inv[i], &wR[i]); fe_add(SC, &l1[i], &cs->a_L[i], &t);
    l2[i] = cs->a_O[i];
    l3[i] = s_L[i];
    fe_sub(SC, &r0[i], &wO[i], &exp_y);
    fe_mul(SC, &t, &exp_y, &cs->a_R[i]); fe_add(SC, &r1[i], &t, &wL[i]);
    fe_mul(SC, &r3[i], &exp_y, &s_R[i]);
    fe_mul(SC, &exp_y, &exp_y, &y);
  }
  /* util.rs:152-170 */
  sc t1, t2, t3, t4, t5, t6, tt;
  sc_inner_product(&t1, l1, r0, n);
  sc_inner_product(&t2, l1, r1, n); sc_inner_product(&tt, l2, r0, n); fe_add(SC, &t2, &t2, &tt);
  sc_inner_product(&t3, l2, r1, n); sc_inner_product(&tt, l3, r0, n); fe_add(SC, &t3, &t3, &tt);
  sc_inner_product(&t4, l1, r3, n); sc_inner_product(&tt, l3, r1, n); fe_add(SC, &t4, &t4, &tt);
  sc_inner_product(&t5, l2, r3, n);
  sc_inner_product(&t6, l3, r3, n);
  sc tb1, tb3, tb4, tb5, tb6;
  sm_scalar(rng, &tb1); sm_scalar(rng, &tb3); sm_scalar(rng, &tb4); sm_scalar(rng, &tb5); sm_scalar(rng, &tb6);
  commit(&out->T_1, cs, &t1, &tb1);
  commit(&out->T_3, cs, &t3, &tb3);
  commit(&out->T_4, cs, &t4, &tb4);
  commit(&out->T_5, cs, &t5, &tb5);
  commit(&out->T_6, cs, &t6, &tb6);
  tr_append_point(tr, "T_1", &out->T_1);
  tr_append_point(tr, "T_3", &out->T_3);
  tr_append_point(tr, "T_4", &out->T_4);
  tr_append_point(tr, "T_5", &out->T_5);
  tr_append_point(tr, "T_6", &out->T_6);
  sc u, x;
  tr_challenge_scalar(tr, "u", &u);
  tr_challenge_scalar(tr, "x", &x);
  sc tb2;
  sc_inner_product(&tb2, wV, cs->v_blinding, m); /* prover.rs:644-648 */
  /* util.rs:192-194 Poly6::eval */
#define POLY6(out_, c1, c2, c3, c4, c5, c6)                                   \
  do {                                                                        \
    sc acc_ = (c6);                                                           \
    fe_mul(SC, &acc_, &acc_, &x); fe_add(SC, &acc_, &acc_, &(c5));            \
    fe_mul(SC, &acc_, &acc_, &x); fe_add(SC, &acc_, &acc_, &(c4));            \
    fe_mul(SC, &acc_, &acc_, &x); fe_add(SC, &acc_, &acc_, &(c3));            \
    fe_mul(SC, &acc_, &acc_, &x); fe_add(SC, &acc_, &acc_, &(c2));            \
    fe_mul(SC, &acc_, &acc_, &x); fe_add(SC, &acc_, &acc_, &(c1));            \
    fe_mul(SC, &(out_), &acc_, &x);                                           \
  } while (0)
  POLY6(out->t_x, t1, t2, t3, t4, t5, t6);
  POLY6(out->t_x_blinding, tb1, tb2, tb3, tb4, tb5, tb6);
  /* util.rs:172-181 VecPoly3::eval; prover.rs:661-672 padding */
  sc *l_vec = (sc *)calloc(padded_n, sizeof(sc)), *r_vec = (sc *)calloc(padded_n, sizeof(sc));
  for (size_t i = 0; i < n; i++) {
    sc acc = l3[i];
    fe_mul(SC, &acc, &acc, &x); fe_add(SC, &acc, &acc, &l2[i]);
    fe_mul(SC, &acc, &acc, &x); fe_add(SC, &acc, &acc, &l1[i]);
    fe_mul(SC, &l_vec[i], &acc, &x); /* l0 = 0 */
    acc = r3[i];
    fe_mul(SC, &acc, &acc, &x);      /* r2 = 0 */
    fe_mul(SC, &acc, &acc, &x); fe_add(SC, &acc, &acc, &r1[i]);
    fe_mul(SC, &acc, &acc, &x); fe_add(SC, &r_vec[i], &acc, &r0[i]);
  }
  for (size_t i = n; i < padded_n; i++) {
    fe_neg(SC, &r_vec[i], &exp_y);
    fe_mul(SC, &exp_y, &exp_y, &y);
  }
  sc i_b, o_b, s_b;
  fe_mul(SC, &t, &u, &i_b2); fe_add(SC, &i_b, &i_b1, &t);
  fe_mul(SC, &t, &u, &o_b2); fe_add(SC, &o_b, &o_b1, &t);
  fe_mul(SC, &t, &u, &s_b2); fe_add(SC, &s_b, &s_b1, &t);
  /* e_blinding = x (i_b + x (o_b + x s_b)), prover.rs:678 */
  fe_mul(SC, &t, &x, &s_b); fe_add(SC, &t, &t, &o_b);
  fe_mul(SC, &t, &t, &x); fe_add(SC, &t, &t, &i_b);
  fe_mul(SC, &out->e_blinding, &t, &x);
  tr_append_scalar(tr, "t_x", &out->t_x);
  tr_append_scalar(tr, "t_x_blinding", &out->t_x_blinding);
  tr_append_scalar(tr, "e_blinding", &out->e_blinding);
  sc w;
  tr_challenge_scalar(tr, "w", &w);
  aff Q;
  {
    jac bj, qj;
    jac_from_aff(&bj, &cs->B);
    jac_mul(&qj, &bj, &w);
    jac_to_aff(&Q, &qj);
  }
  /* prover.rs:689-697 */
  sc *Gf = (sc *)malloc(padded_n * sizeof(sc)), *Hf = (sc *)malloc(padded_n * sizeof(sc));
  for (size_t i = 0; i < padded_n; i++) {
    Gf[i] = i < n1 ? SC->one : u;
    fe_mul(SC, &Hf[i], &exp_y_inv[i], &Gf[i]);
  }
  aff *Gc = (aff *)malloc(padded_n * sizeof(aff)), *Hc = (aff *)malloc(padded_n * sizeof(aff));
  memcpy(Gc, G, padded_n * sizeof(aff));
  memcpy(Hc, H, padded_n * sizeof(aff));
  ipp_create(&out->ipp, tr, &Q, Gf, Hf, Gc, Hc, l_vec, r_vec, padded_n, NULL);
  free(Gc); free(Hc); free(Gf); free(Hf); free(l_vec); free(r_vec);
  free(l1); free(l2); free(l3); free(r0); free(r1); free(r3);
  free(exp_y_inv); free(wL); free(wR); free(wO); free(wV); free(s_L); free(s_R);
  (void)pad;
  return BPO_OK;
}

/* verifier.rs:393-554 */
int cs_verify(cs_t *cs, const r1cs_proof *proof, const aff *G, const aff *H, size_t gens_capacity,
              verify_trace *trace) {
  transcript *tr = cs->tr;
  if (trace) memset(trace, 0, sizeof *trace);
  tr_append_u64(tr, "m", cs->nv);
  size_t n1 = cs->num_vars;
  if (tr_validate_and_append_point(tr, "A_I1", &proof->A_I1)) return BPO_ERR_VERIFICATION;
  if (tr_validate_and_append_point(tr, "A_O1", &proof->A_O1)) return BPO_ERR_VERIFICATION;
  if (tr_validate_and_append_point(tr, "S1", &proof->S1)) return BPO_ERR_VERIFICATION;
  int rc = create_randomized_constraints(cs);
  if (rc) return rc;
  size_t n = cs->num_vars, n2 = n - n1, padded_n = next_pow2(n), pad = padded_n - n, m = cs->nv;
  if (gens_capacity < padded_n) return BPO_ERR_GENS;
  tr_append_point(tr, "A_I2", &proof->A_I2);
  tr_append_point(tr, "A_O2", &proof->A_O2);
  tr_append_point(tr, "S2", &proof->S2);
  sc y, z, u, x, w, r;
  tr_challenge_scalar(tr, "y", &y);
  tr_challenge_scalar(tr, "z", &z);
  if (tr_validate_and_append_point(tr, "T_1", &proof->T_1)) return BPO_ERR_VERIFICATION;
  if (tr_validate_and_append_point(tr, "T_3", &proof->T_3)) return BPO_ERR_VERIFICATION;
  if (tr_validate_and_append_point(tr, "T_4", &proof->T_4)) return BPO_ERR_VERIFICATION;
  if (tr_validate_and_append_point(tr, "T_5", &proof->T_5)) return BPO_ERR_VERIFICATION;
  if (tr_validate_and_append_point(tr, "T_6", &proof->T_6)) return BPO_ERR_VERIFICATION;
  tr_challenge_scalar(tr, "u", &u);
  tr_challenge_scalar(tr, "x", &x);
  tr_append_scalar(tr, "t_x", &proof->t_x);
  tr_append_scalar(tr, "t_x_blinding", &proof->t_x_blinding);
  tr_append_scalar(tr, "e_blinding", &proof->e_blinding);
  tr_challenge_scalar(tr, "w", &w);
  size_t na = n ? n : 1;
  sc *wL = (sc *)malloc(na * sizeof(sc)), *wR = (sc *)malloc(na * sizeof(sc)), *wO = (sc *)malloc(na * sizeof(sc));
  sc *wV = (sc *)malloc((m ? m : 1) * sizeof(sc)), wc;
  cs_flattened_constraints(cs, &z, wL, wR, wO, wV, &wc);
  size_t k = proof->ipp.k;
  sc *ch = (sc *)malloc((k ? k : 1) * sizeof(sc));
  rc = ipp_challenges(&proof->ipp, padded_n, tr, ch);
  if (rc) { free(wL); free(wR); free(wO); free(wV); free(ch); return BPO_ERR_VERIFICATION; }
  sc *u_sq = (sc *)malloc((k ? k : 1) * sizeof(sc)), *u_inv_sq = (sc *)malloc((k ? k : 1) * sizeof(sc));
  sc *s = (sc *)malloc(padded_n * sizeof(sc));
  verification_scalars(ch, k, padded_n, u_sq, u_inv_sq, s);
  const sc *a = &proof->ipp.a, *b = &proof->ipp.b;
  sc y_inv, t, t2;
  fe_inv(SC, &y_inv, &y);
  sc *y_inv_vec = (sc *)malloc(padded_n * sizeof(sc));
  exp_iter(y_inv_vec, &y_inv, padded_n);
  sc *yneg_wR = (sc *)calloc(padded_n, sizeof(sc));
  for (size_t i = 0; i < n; i++) fe_mul(SC, &yneg_wR[i], &wR[i], &y_inv_vec[i]);
  sc delta;
  sc_inner_product(&delta, yneg_wR, wL, n);
  tr_challenge_scalar(tr, "r", &r); /* verifier.rs:506 */
  sc xx, rxx, xxx;
  fe_mul(SC, &xx, &x, &x);
  fe_mul(SC, &rxx, &r, &xx);
  fe_mul(SC, &xxx, &x, &xx);
  size_t nt = 13 + m + 2 * padded_n + 2 * k;
  sc *ms = (sc *)malloc(nt * sizeof(sc));
  aff *mp = (aff *)malloc(nt * sizeof(aff));
  size_t o = 0;
  ms[o] = x; mp[o++] = proof->A_I1;
  ms[o] = xx; mp[o++] = proof->A_O1;
  ms[o] = xxx; mp[o++] = proof->S1;
  fe_mul(SC, &ms[o], &u, &x); mp[o++] = proof->A_I2;
  fe_mul(SC, &ms[o], &u, &xx); mp[o++] = proof->A_O2;
  fe_mul(SC, &ms[o], &u, &xxx); mp[o++] = proof->S2;
  for (size_t i = 0; i < m; i++) { fe_mul(SC, &ms[o], &wV[i], &rxx); mp[o++] = cs->V[i]; }
  fe_mul(SC, &ms[o], &r, &x); mp[o++] = proof->T_1;          /* r x      */
  fe_mul(SC, &ms[o], &rxx, &x); mp[o++] = proof->T_3;        /* r x^3    */
  fe_mul(SC, &ms[o], &rxx, &xx); mp[o++] = proof->T_4;       /* r x^4    */
  fe_mul(SC, &ms[o], &rxx, &xxx); mp[o++] = proof->T_5;      /* r x^5    */
  fe_mul(SC, &t, &rxx, &xx); fe_mul(SC, &ms[o], &t, &xx); mp[o++] = proof->T_6; /* r x^6 */
  /* B: w (t_x - a b) + r (xx (wc + delta) - t_x), verifier.rs:525-527 */
  fe_mul(SC, &t, a, b); fe_sub(SC, &t, &proof->t_x, &t); fe_mul(SC, &t, &w, &t);
  fe_add(SC, &t2, &wc, &delta); fe_mul(SC, &t2, &xx, &t2); fe_sub(SC, &t2, &t2, &proof->t_x); fe_mul(SC, &t2, &r, &t2);
  fe_add(SC, &ms[o], &t, &t2); mp[o++] = cs->B;
  /* B_blinding: -e_blinding - r t_x_blinding */
  fe_mul(SC, &t, &r, &proof->t_x_blinding); fe_add(SC, &t, &t, &proof->e_blinding); fe_neg(SC, &ms[o], &t); mp[o++] = cs->B_blinding;
  /* g_scalars verifier.rs:487-491 */
  for (size_t i = 0; i < padded_n; i++) {
    fe_mul(SC, &t, &x, &yneg_wR[i]); fe_mul(SC, &t2, a, &s[i]); fe_sub(SC, &t, &t, &t2);
    if (i >= n1) fe_mul(SC, &t, &u, &t);
    ms[o] = t; mp[o++] = G[i];
  }
  /* h_scalars verifier.rs:493-501 */
  for (size_t i = 0; i < padded_n; i++) {
    sc acc;
    memset(&acc, 0, sizeof acc);
    if (i < n) { fe_mul(SC, &acc, &x, &wL[i]); fe_add(SC, &acc, &acc, &wO[i]); }
    fe_mul(SC, &t2, b, &s[padded_n - 1 - i]); fe_sub(SC, &acc, &acc, &t2);
    fe_mul(SC, &acc, &y_inv_vec[i], &acc); fe_sub(SC, &acc, &acc, &SC->one);
    if (i >= n1) fe_mul(SC, &acc, &u, &acc);
    ms[o] = acc; mp[o++] = H[i];
  }
  for (size_t i = 0; i < k; i++) { ms[o] = u_sq[i]; mp[o++] = proof->ipp.L[i]; }
  for (size_t i = 0; i < k; i++) { ms[o] = u_inv_sq[i]; mp[o++] = proof->ipp.R[i]; }
  jac mega;
  msm(&mega, ms, mp, nt);
  rc = jac_is_inf(&mega) ? BPO_OK : BPO_ERR_VERIFICATION;
  if (trace) {
    trace->y = y; trace->z = z; trace->u = u; trace->x = x; trace->w = w; trace->r = r;
    trace->ipp_u = ch; ch = NULL;
    trace->n1 = n1; trace->n2 = n2; trace->padded_n = padded_n; trace->k = k; trace->m = m; trace->nterms = nt;
    trace->scalars = ms; ms = NULL;
    trace->points = mp; mp = NULL;
    jac_to_aff(&trace->mega_check, &mega);
  }
  free(ms); free(mp); free(ch); free(u_sq); free(u_inv_sq); free(s); free(y_inv_vec); free(yneg_wR);
  free(wL); free(wR); free(wO); free(wV);
  (void)pad;
  return rc;
}

/* ---------------------------------------------------------------- gadgets */
/* tests/r1cs.rs:620-652 */
void gadget_range_proof(cs_t *cs, var_t v, int have, uint64_t q, size_t n_bits) {
  lincomb vl;
  lc_init(&vl);
  lc_add_term_i64(&vl, v, 1);
  sc exp_2 = SC->one;
  for (size_t i = 0; i < n_bits; i++) {
    var_t abo[3];
    sc l, r;
    uint64_t bit = have ? (q >> i) & 1 : 0;
    fe_from_u64(SC, &l, 1 - bit);
    fe_from_u64(SC, &r, bit);
    cs_allocate_multiplier(cs, &l, &r, abo);
    lincomb c;
    lc_init(&c); lc_add_term_i64(&c, abo[2], 1); cs_constrain(cs, &c);                 /* o = 0 */
    lc_init(&c); lc_add_term_i64(&c, abo[0], 1); lc_add_term_i64(&c, abo[1], 1);
    lc_add_term_i64(&c, (var_t){VAR_ONE, 0}, -1); cs_constrain(cs, &c);                 /* a + (b - 1) */
    sc neg;
    fe_neg(SC, &neg, &exp_2);
    lc_add_term(&vl, abo[1], &neg);                                                     /* v -= b * 2^i */
    fe_add(SC, &exp_2, &exp_2, &exp_2);
  }
  cs_constrain(cs, &vl);
}

typedef struct { var_t *x, *y; size_t k; } shuffle_ctx;
static int shuffle_cb(cs_t *cs, void *vctx) {
  shuffle_ctx *c = (shuffle_ctx *)vctx;
  size_t k = c->k;
  sc z, mz;
  cs_challenge_scalar(cs, "shuffle challenge", &z);
  fe_neg(SC, &mz, &z);
  var_t one = {VAR_ONE, 0}, out[3], firstx;
  for (int side = 0; side < 2; side++) {
    var_t *v = side ? c->y : c->x;
    lincomb l, r;
    lc_init(&l); lc_add_term_i64(&l, v[k - 1], 1); lc_add_term(&l, one, &mz);
    lc_init(&r); lc_add_term_i64(&r, v[k - 2], 1); lc_add_term(&r, one, &mz);
    cs_multiply(cs, &l, &r, out);
    for (size_t i = k - 2; i-- > 0;) {
      lc_init(&l); lc_add_term_i64(&l, out[2], 1);
      lc_init(&r); lc_add_term_i64(&r, v[i], 1); lc_add_term(&r, one, &mz);
      cs_multiply(cs, &l, &r, out);
    }
    if (!side) firstx = out[2];
  }
  lincomb e;
  lc_init(&e); lc_add_term_i64(&e, firstx, 1); lc_add_term_i64(&e, out[2], -1);
  cs_constrain(cs, &e);
  free(c->x); free(c->y); free(c);
  return 0;
}
/* tests/r1cs.rs:23-62 */
void gadget_shuffle(cs_t *cs, const var_t *x, const var_t *y, size_t k) {
  if (k == 1) {
    lincomb l;
    lc_init(&l); lc_add_term_i64(&l, y[0], 1); lc_add_term_i64(&l, x[0], -1);
    cs_constrain(cs, &l);
    return;
  }
  shuffle_ctx *c = (shuffle_ctx *)malloc(sizeof *c);
  c->k = k;
  c->x = (var_t *)malloc(k * sizeof(var_t));
  c->y = (var_t *)malloc(k * sizeof(var_t));
  memcpy(c->x, x, k * sizeof(var_t));
  memcpy(c->y, y, k * sizeof(var_t));
  cs_specify_randomized_constraints(cs, shuffle_cb, c);
}
/* tests/r1cs.rs:217-228 with (a1,a2,b1,b2,c1) = committed v[0..5), c2 constant */
void gadget_example(cs_t *cs, const var_t v[5], uint64_t c2) {
  lincomb l, r, c;
  var_t out[3];
  lc_init(&l); lc_add_term_i64(&l, v[0], 1); lc_add_term_i64(&l, v[1], 1);
  lc_init(&r); lc_add_term_i64(&r, v[2], 1); lc_add_term_i64(&r, v[3], 1);
  cs_multiply(cs, &l, &r, out);
  sc c2s;
  fe_from_u64(SC, &c2s, c2);
  lc_init(&c); lc_add_term_i64(&c, v[4], 1); lc_add_term(&c, (var_t){VAR_ONE, 0}, &c2s);
  lc_add_term_i64(&c, out[2], -1);
  cs_constrain(cs, &c);
}
/* benches/r1cs.rs:24-33: a chain of squarings of one public value */
void gadget_dummy(cs_t *cs, const sc *val, size_t n_constraints) {
  var_t var = cs_commit_public(cs, val), out[3];
  for (size_t i = 0; i < n_constraints; i++) {
    lincomb l, r;
    lc_init(&l); lc_add_term_i64(&l, var, 1);
    lc_init(&r); lc_add_term_i64(&r, var, 1);
    cs_multiply(cs, &l, &r, out);
    var = out[2];
  }
}
