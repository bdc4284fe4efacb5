/*
 * bpo.h -- CPU oracle for the Bulletproofs hot path over the Stark curve.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load liboracle.so.  The product
 * (mpc_bulletproof_amd/) never links, includes or calls anything in oracle/.
 *
 * What it restates: the arithmetic that renegade-fi/mpc-bulletproof delegates to
 * the crate `mpc-stark 0.2` (Cargo.toml:21; crates.io; NOT under /root/reference,
 * no Cargo.lock) -- Scalar, StarkPoint, msm, batch_inverse -- plus the crate's own
 * inner-product argument (src/inner_product_proof.rs), R1CS prover/verifier
 * (src/r1cs/prover.rs, src/r1cs/verifier.rs), util (src/util.rs) and generator
 * chain (src/generators.rs).  Every function cites the file:line it follows.
 *
 * Pinning: checked against the reference's own KATs (inner_product == 40
 * inner_product_proof.rs:620-635; powers of 2 util.rs:295-303; sums of powers of
 * 10 util.rs:335-345; sum_of_powers == slow util.rs:322-333; EXAMPLE_GADGET_WEIGHTS
 * tests/r1cs.rs:516-538) and against golden vectors emitted by the independent
 * pure-Python model oracle/pymodel.py (tests/golden/, script oracle/gen_golden.py).
 * Transcript bytes: PARITY UNPINNED (merlin fork source absent) -- see pymodel.py.
 * The reference itself cannot be built here (Rust; no toolchain) -- DESIGN.md.
 */
#ifndef BPO_H
#define BPO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ fields */
typedef struct { uint64_t v[4]; } fe;           /* Montgomery form, R = 2^256, canonical [0,m) */
typedef struct {
  uint64_t m[4];   /* modulus */
  uint64_t n0;     /* -m^-1 mod 2^64 */
  fe r2;           /* R^2 mod m */
  fe one;          /* R mod m */
} fctx;
extern const fctx BPO_FP; /* base field   p = 2^251 + 17*2^192 + 1 */
extern const fctx BPO_FN; /* scalar field n (group order)          */

void fe_add(const fctx *c, fe *r, const fe *a, const fe *b);
void fe_sub(const fctx *c, fe *r, const fe *a, const fe *b);
void fe_neg(const fctx *c, fe *r, const fe *a);
void fe_mul(const fctx *c, fe *r, const fe *a, const fe *b);
void fe_inv(const fctx *c, fe *r, const fe *a);          /* Fermat; 0 -> 0 */
void fe_from_u64(const fctx *c, fe *r, uint64_t x);
int  fe_from_le(const fctx *c, fe *r, const uint8_t b[32]);   /* -1 if >= m */
void fe_from_le_wide(const fctx *c, fe *r, const uint8_t b[64]); /* 512-bit LE mod m */
void fe_to_le(const fctx *c, uint8_t b[32], const fe *a);     /* canonical */
void fe_to_int(const fctx *c, uint64_t out[4], const fe *a);  /* canonical integer */
int  fe_is_zero(const fe *a);
int  fe_eq(const fe *a, const fe *b);

/* scalar-field shorthands */
#define SC (&BPO_FN)
typedef fe sc;
void sc_batch_inverse(sc *v, size_t n);   /* Scalar::batch_inverse, inner_product_proof.rs:283 */
void sc_inner_product(sc *out, const sc *a, const sc *b, size_t n); /* inner_product_proof.rs:463 */

/* ------------------------------------------------------------------ curve */
typedef struct { fe X, Y, Z; } jac;     /* Jacobian; Z == 0 <=> identity */
typedef struct { fe x, y; int inf; } aff;

void jac_set_inf(jac *r);
int  jac_is_inf(const jac *a);
void jac_from_aff(jac *r, const aff *a);
void jac_to_aff(aff *r, const jac *a);
void jac_neg(jac *r, const jac *a);
void jac_dbl(jac *r, const jac *a);
void jac_add(jac *r, const jac *a, const jac *b);     /* complete */
void jac_madd(jac *r, const jac *a, const aff *b);    /* complete */
int  jac_eq(const jac *a, const jac *b);
void jac_mul(jac *r, const jac *p, const sc *k);      /* double-and-add */
int  aff_from_bytes(aff *r, const uint8_t b[64]);     /* x||y LE; zeros = identity; -1 off-curve */
void aff_to_bytes(uint8_t b[64], const aff *a);       /* util.rs:274-289 */
void jac_to_bytes(uint8_t b[64], const jac *a);
extern const aff BPO_G;                                /* curve generator */

/* MSM: StarkPoint::msm_iter (mpc-stark) -- call sites SURVEY K1 */
void msm_naive(jac *r, const sc *s, const aff *p, size_t n);
void msm_pippenger(jac *r, const sc *s, const aff *p, size_t n);
void msm(jac *r, const sc *s, const aff *p, size_t n);     /* picks by n */
void batch_to_aff(aff *out, const jac *in, size_t n);       /* one inversion */

/* ------------------------------------------------------------------ hash / transcript */
void bpo_keccak256(const uint8_t *in, size_t len, uint8_t out[32]);
void hash_to_scalar(sc *r, const uint8_t low[32]);          /* util.rs:252-267 */

typedef struct { uint8_t state[32]; } transcript;
void tr_init(transcript *t, const uint8_t *label, size_t len);
void tr_append_message(transcript *t, const char *label, const uint8_t *msg, size_t len);
void tr_append_u64(transcript *t, const char *label, uint64_t x);
void tr_challenge_bytes(transcript *t, const char *label, uint8_t out[32]);
/* TranscriptProtocol, src/transcript.rs:63-121 */
void tr_innerproduct_domain_sep(transcript *t, uint64_t n);
void tr_r1cs_domain_sep(transcript *t);
void tr_r1cs_1phase_domain_sep(transcript *t);
void tr_r1cs_2phase_domain_sep(transcript *t);
void tr_append_scalar(transcript *t, const char *label, const sc *s);
void tr_append_point(transcript *t, const char *label, const aff *p);
int  tr_validate_and_append_point(transcript *t, const char *label, const aff *p); /* -1 identity */
void tr_challenge_scalar(transcript *t, const char *label, sc *out);

/* ------------------------------------------------------------------ generators */
/* src/generators.rs:76-129, 182-235: party share `party`, label 'G' or 'H' */
void gens_chain(aff *out, sc *dlog_or_null, char which, uint32_t party, size_t count);

/* injectable RNG replacing thread_rng() (prover.rs:435-445) */
/* vec_keys != 0: the prover's blinding VECTORS s_L, s_R (prover.rs:461-462, 526-527) are not drawn scalar by scalar from the
 * stream; a 32-byte key (four words of the stream, little endian) is drawn in their place, per phase, and expanded by
 * blind_vector ("BlindVec v1": what include/bpgpu.h's bpgpu_r1cs_prover_commit does on the device). */
typedef struct { uint64_t s; int vec_keys; } splitmix;
uint64_t sm_next(splitmix *r);
void sm_scalar(splitmix *r, sc *out);
/* s_v[i] for i < count (v = 0: s_L, 1: s_R):  block(key, v, j) = first 128 bytes of Keccak-f[1600] over the keccak256-padded
 * 48-byte message key || u64le(v) || u64le(j);  s_v[i] = int_LE(block(key, v, i / 2)[64 (i mod 2) .. +64]) mod n */
void blind_vector(sc *out, const uint8_t key[32], int v, size_t count);

/* ------------------------------------------------------------------ util.rs */
void exp_iter(sc *out, const sc *x, size_t n);              /* util.rs:73-76 */
void sum_of_powers(sc *out, const sc *x, size_t n);         /* util.rs:218-234 */
void sum_of_powers_slow(sc *out, const sc *x, size_t n);    /* util.rs:237-239 */

/* ------------------------------------------------------------------ inner product proof */
typedef struct {
  size_t k;            /* lg n */
  aff *L, *R;          /* k each */
  sc a, b;
} ipp_proof;
void ipp_free(ipp_proof *p);
/* inner_product_proof.rs:202-248; outputs length n (inputs are the L/R halves) */
void fold_witness(const sc *u, const sc *u_inv, size_t n,
                  const sc *a_L, const sc *a_R, const sc *b_L, const sc *b_R,
                  const aff *G_L, const aff *G_R, const aff *H_L, const aff *H_R,
                  sc *a_out, sc *b_out, aff *G_out, aff *H_out);
/* inner_product_proof.rs:49-193. G,H,a,b are consumed (overwritten). challenges_out (k) optional */
void ipp_create(ipp_proof *out, transcript *t, const aff *Q, const sc *G_factors,
                const sc *H_factors, aff *G, aff *H, sc *a, sc *b, size_t n, sc *challenges_out);
/* inner_product_proof.rs:259-278 (transcript replay). -1 on VerificationError */
int ipp_challenges(const ipp_proof *p, size_t n, transcript *t, sc *challenges);
/* inner_product_proof.rs:280-309 */
void verification_scalars(const sc *challenges, size_t k, size_t n, sc *u_sq, sc *u_inv_sq, sc *s);
/* inner_product_proof.rs:317-372: 0 ok, -1 VerificationError */
int ipp_verify(const ipp_proof *p, size_t n, transcript *t, const sc *G_factors,
               const sc *H_factors, const aff *P, const aff *Q, const aff *G, const aff *H);

/* ------------------------------------------------------------------ R1CS */
enum { VAR_L = 0, VAR_R = 1, VAR_O = 2, VAR_V = 3, VAR_ONE = 4 }; /* linear_combination.rs:15-28 */
typedef struct { uint32_t kind, idx; } var_t;
typedef struct { var_t var; sc coeff; } term_t;
typedef struct { term_t *t; size_t n, cap; } lincomb;

void lc_init(lincomb *l);
void lc_free(lincomb *l);
void lc_copy(lincomb *d, const lincomb *s);
void lc_add_term(lincomb *l, var_t v, const sc *coeff);   /* linear_combination.rs:129-135 */
void lc_add_term_i64(lincomb *l, var_t v, int64_t c);

typedef struct cs cs_t;
typedef int (*cs_callback)(cs_t *cs, void *ctx);

typedef struct {
  aff A_I1, A_O1, S1, A_I2, A_O2, S2, T_1, T_3, T_4, T_5, T_6;
  sc t_x, t_x_blinding, e_blinding;
  ipp_proof ipp;
} r1cs_proof;   /* src/r1cs/proof.rs:35-67 */
void r1cs_proof_free(r1cs_proof *p);

struct cs {
  int is_prover;
  transcript *tr;
  aff B, B_blinding;                 /* PedersenGens, generators.rs:61-70 */
  lincomb *constraints; size_t nc, cap_c;
  /* prover */
  sc *a_L, *a_R, *a_O; size_t nmul, cap_mul;
  sc *v, *v_blinding; size_t nv, cap_v;
  /* verifier */
  aff *V;                             /* shares nv/cap_v with the prover arrays */
  size_t num_vars;
  /* deferred */
  cs_callback cb[8]; void *cb_ctx[8]; size_t ncb;
  long pending_multiplier;
};
void cs_init(cs_t *cs, int is_prover, transcript *tr);
void cs_free(cs_t *cs);
var_t cs_commit_prover(cs_t *cs, const sc *v, const sc *v_blinding, aff *V_out); /* prover.rs:319 */
var_t cs_commit_verifier(cs_t *cs, const aff *V);                                 /* verifier.rs:298 */
var_t cs_commit_public(cs_t *cs, const sc *v);                 /* prover.rs:171 / verifier.rs:153 */
void cs_multiply(cs_t *cs, lincomb *left, lincomb *right, var_t out[3]);  /* consumes left,right */
void cs_allocate_multiplier(cs_t *cs, const sc *l, const sc *r, var_t out[3]);
void cs_constrain(cs_t *cs, lincomb *l);                        /* consumes l */
void cs_specify_randomized_constraints(cs_t *cs, cs_callback cb, void *ctx);
void cs_challenge_scalar(cs_t *cs, const char *label, sc *out);
size_t cs_num_multipliers(const cs_t *cs);
void cs_eval(const cs_t *cs, const lincomb *l, sc *out);        /* prover.rs:179-194 */

/* prover.rs:342-379 / verifier.rs:323-362 (wc only written for the verifier; may be NULL) */
void cs_flattened_constraints(const cs_t *cs, const sc *z, sc *wL, sc *wR, sc *wO, sc *wV, sc *wc);

typedef struct {          /* optional trace of one verification (fixtures / GPU parity) */
  sc y, z, u, x, w, r;
  sc *ipp_u;              /* k */
  size_t n1, n2, padded_n, k, m, nterms;
  sc *scalars;            /* nterms, order verifier.rs:517-532 */
  aff *points;            /* nterms, order verifier.rs:533-546 */
  aff mega_check;
} verify_trace;
void verify_trace_free(verify_trace *t);

#define BPO_OK 0
#define BPO_ERR_VERIFICATION (-1)   /* R1CSError::VerificationError */
#define BPO_ERR_GENS (-2)           /* R1CSError::InvalidGeneratorsLength */
/* prover.rs:412-727; gens: share-0 G,H arrays of capacity gens_capacity */
int cs_prove(cs_t *cs, const aff *G, const aff *H, size_t gens_capacity, splitmix *rng,
             r1cs_proof *out);
/* verifier.rs:393-554 */
int cs_verify(cs_t *cs, const r1cs_proof *proof, const aff *G, const aff *H,
              size_t gens_capacity, verify_trace *trace_or_null);

/* gadgets (tests/r1cs.rs, benches/r1cs.rs) */
void gadget_range_proof(cs_t *cs, var_t v, int have_assignment, uint64_t v_assignment, size_t n_bits); /* tests/r1cs.rs:620-652 */
void gadget_shuffle(cs_t *cs, const var_t *x, const var_t *y, size_t k);  /* tests/r1cs.rs:23-62 */
void gadget_example(cs_t *cs, const var_t v[5], uint64_t c2);             /* tests/r1cs.rs:217-228 */
void gadget_dummy(cs_t *cs, const sc *val, size_t n_constraints);          /* benches/r1cs.rs:24-33 */

#ifdef __cplusplus
}
#endif
#endif
