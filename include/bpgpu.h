/*
 * bpgpu.h -- C ABI of the MI355X-native Bulletproofs hot path (Stark curve).
 *
 * Drop-in boundary for renegade-fi/mpc-bulletproof: the reference has no FFI; its hot path calls
 * free functions of the external crate mpc-stark (Scalar, StarkPoint) at the sites cited below
 * (paths relative to the reference root).  A Rust shim binds these symbols in place of those calls
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - scalar : 32-byte little-endian canonical integer < n (group order).  Non-canonical -> BPGPU_E_ARG.
 *   - point  : affine x || y, 32-byte little-endian canonical each (< p); 64 zero bytes = identity
 *              (the encoding the reference absorbs into the transcript, src/util.rs:274-289).
 *              Off-curve / non-canonical -> BPGPU_E_ARG.
 *   - all pointers are caller-owned; nothing allocated here crosses the boundary except opaque
 *     handles released by the matching *_destroy; functions never throw; return 0 or BPGPU_E_*.
 *   - thread-safe: a bpgpu_ctx may be shared by threads (calls on one ctx are serialised by an
 *     internal mutex; use one ctx per thread for concurrency) -- the reference calls msm from rayon
 *     workers and MPC-fabric executor threads (src/inner_product_proof.rs:233-247, src/transcript.rs:155-161).
 *   - `_dev` variants take device pointers obtained from bpgpu_malloc (or any HIP allocation of the
 *     same device, e.g. a torch tensor's data_ptr) and enqueue on the ctx stream without a host sync.
 *   - no CPU fallback exists: without a usable HIP device every call returns BPGPU_E_DEVICE.
 */
#ifndef BPGPU_H
#define BPGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPGPU_OK 0
#define BPGPU_E_ARG (-1)     /* malformed input: non-canonical scalar, off-curve point, null pointer */
#define BPGPU_E_LEN (-2)     /* length mismatch / not a power of two where required (reference panics: inner_product_proof.rs:62-70) */
#define BPGPU_E_DEVICE (-3)  /* HIP error / no device */
#define BPGPU_E_OOM (-4)
#define BPGPU_E_GENS (-5)    /* generator capacity too small: R1CSError::InvalidGeneratorsLength (verifier.rs:421-423) */

typedef struct bpgpu_ctx bpgpu_ctx;
typedef struct bpgpu_gens bpgpu_gens;       /* resident generators + fixed-base tables */
typedef struct bpgpu_circuit bpgpu_circuit; /* resident constraint weights (column-major) */

/* ---- context ------------------------------------------------------------------------------- */
int bpgpu_device_count(void);
int bpgpu_create(int device, bpgpu_ctx **out);
void bpgpu_destroy(bpgpu_ctx *ctx);
const char *bpgpu_strerror(int code);
const char *bpgpu_last_error(bpgpu_ctx *ctx);   /* text of the last HIP failure on this ctx */
int bpgpu_sync(bpgpu_ctx *ctx);
void *bpgpu_stream(bpgpu_ctx *ctx);             /* hipStream_t of the ctx (for event timing) */
/* Tuning hint for the batch-verification entry points of this context.  0 (default): fewest instructions -- right for a
 * caller that keeps several batches in flight (several contexts / streams), where throughput is bound by instruction issue.
 * 1: more, shorter lanes in the two longest launches of a batch's kernel chain -- one batch alone completes ~25 % sooner
 * (0.8 instead of 1.05 ms for 1024 x 64-bit range proofs), a saturated pipeline runs ~5 % slower.  Results are identical. */
int bpgpu_set_latency_mode(bpgpu_ctx *ctx, int on);
/* Launch-route options of ONE context (results never depend on them: every route computes the same bytes; the parity tests
 * walk all of them through this setter).  The environment variables BPGPU_<NAME> of the same names only seed a new context's
 * defaults, once, inside bpgpu_create -- nothing reads the environment per call.  BPGPU_E_ARG for an unknown option or a
 * value outside its range. */
#define BPGPU_OPT_MSM_WP_MAX 1             /* MSMs of up to this many terms take the window-parallel launches (default 2^15; 0 = never) */
#define BPGPU_OPT_MSM_PIP2_SINGLE 2        /* 1: one mid-size MSM through k_pip2.hip's one-instance bucket pipeline (default 0) */
#define BPGPU_OPT_VERIFY_NO_FUSE 3         /* 1: verification as separate point-import / Straus / fixed-base / finalize launches (default 0) */
#define BPGPU_OPT_VERIFY_WINDOW_PARALLEL 4 /* 0: the fused Straus + fixed-base launch instead of the window-parallel chain (default 1) */
#define BPGPU_OPT_VERIFY_STRAUS_NP 5       /* points per Straus lane on those paths, 1..4 (default 4) */
#define BPGPU_OPT_IPP_LITERAL 6            /* 1: bpgpu_ipp_begin always runs the reference's literal schedule (generators folded every round) */
#define BPGPU_OPT_VS_LARGE_MIN 7           /* padded n / m from which the verifier's scalar assembly is split over the grid (default 4096) */
#define BPGPU_OPT_TABLE_NP 8               /* proof points per table lane of the window-parallel chain: 1, 2, 4, 8; 0 = by latency mode */
#define BPGPU_OPT_IPP_TABLE_MAX_N 9        /* bpgpu_ipp_begin builds per-session generator tables up to this n (default 2^16), literal schedule above */
#define BPGPU_OPT_STREAM_LANES 10          /* lanes (streams + workspaces) a bpgpu_r1cs_verify_stream call spreads its batches over, 1..64 (default 20) */
#define BPGPU_OPT_STREAM_BATCH 11          /* proofs per batch of a bpgpu_r1cs_verify_stream call (default 1024) */
#define BPGPU_OPT_SCREEN_BATCH 12          /* proofs per combined check of a bpgpu_r1cs_verify_screened call (default 2560; capped so that one check's proof points stay within the one-instance bucket pipeline) */
#define BPGPU_OPT_HORNER_FORM 13           /* Horner pass of the window-parallel chain / MSMs: 0 = by mode, 1 = a lane, 2 = a DPP quad, 3 = a whole wave (row-distributed field arithmetic) per proof / group */
#define BPGPU_OPT_HORNER_ROW_MAX 14        /* latency mode and the MSMs take the wave-per-chain form up to this many chains (default 1536: one wave per SIMD and a half) */
#define BPGPU_OPT_PIPPENGER_MIN 15         /* terms from which an MSM that is not served by the window-parallel launches takes the bucket method (default 512) */
#define BPGPU_OPT_IPP_PIPPENGER_MIN 16     /* the same for the L / R MSMs of bpgpu_ipp_round's literal schedule (default 257) */
#define BPGPU_OPT_FIXED_LPM 17             /* lanes per fixed-base MSM in the verification's back launch: 16, 32, 64; 0 = by mode */
#define BPGPU_OPT_GROUPS_FORM 18           /* first Horner stage: 0 = by mode, 1 = a lane, 2 = a DPP quad, 3 = a whole wave per group of 8 windows */
#define BPGPU_OPT_FIXED_CHUNK_GENS 19      /* generator half of the verification's back launch: 0 = by mode, -1 = FIXED_LPM lanes per proof and a butterfly; g in 1..64 = a proof
                                            * per lane, g generators per wave, the partial sums added in the verdict launch */
#define BPGPU_OPT_COUNT 20
int bpgpu_set_option(bpgpu_ctx *ctx, int option, int64_t value);
int bpgpu_get_option(bpgpu_ctx *ctx, int option, int64_t *value);
/* synchronise and report whether an operand of the `_dev` (device-resident, asynchronous) calls issued since the last read -- or
 * since the last synchronous entry point on this context, each of which uses and clears the flag for its own BPGPU_E_ARG -- was
 * malformed (1); reading clears the flag.  A diagnostic: the verification entry points reject a malformed proof on its own
 * (ok[p] = 0; that includes a non-canonical gadget challenge of bpgpu_r1cs_verify_batch_param) and never fail the call for it. */
int bpgpu_input_flag(bpgpu_ctx *ctx, int *bad);

/* Per-kernel timing with HIP events recorded on the stream each kernel is launched on (the numbers
 * bench.py's `roofline` uses).  BPGPU_PROF_KINDS kinds.  Window-parallel verification chain (the default):
 * 8 front (proof-point tables | inversion pass), 0 scalar assembly (k_verify_scalars), 7 window sums, 9 Horner
 * groups, 10 back (Horner pass | fixed-base MSMs), 11 verdict; 5 device transcript; 12..15 the stages of the
 * combined batch check (scalars + weights, proof-point MSM, generator MSM, tail).  Other launch paths: 6 fused Straus +
 * fixed-base launch, 1 fixed-base MSM, 2 point import, 3 Straus, 4 verify tail.  read() synchronises, returns sums
 * since the last read (at most 64 launches per kind are kept between two reads).  select() restricts the timing to the
 * kinds of a bit mask (bit k = kind k; default all): an event pair is a barrier in the queue on either side of its kernel,
 * so a caller that wants one kernel's duration out of a pipelined run pays for that kernel only. */
#define BPGPU_PROF_KINDS 24
/* Kinds 16..21 cover the prover (one pair per device phase of a call, i.e. from its first to its last launch): 16 phase commitments
 * (bpgpu_r1cs_prover_commit), 17 polynomial build, 18 bpgpu_msm_gens (T commitments), 19 IPP session set-up, 20 the IPP round
 * loop, 21 the L / R table-lookup MSM of one round (the prover's dominant kernel, nested inside 20). */
int bpgpu_profile_enable(bpgpu_ctx *ctx, int on);
int bpgpu_profile_select(bpgpu_ctx *ctx, uint32_t kind_mask);
int bpgpu_profile_read(bpgpu_ctx *ctx, double ms_sum[BPGPU_PROF_KINDS], uint64_t launches[BPGPU_PROF_KINDS]);
/* The same measurements as intervals: bpgpu_profile_epoch records (and waits for) a reference event on ctx's stream and returns
 * its handle (owned by ctx, valid until bpgpu_destroy; NULL on failure); bpgpu_profile_intervals returns every timed launch since
 * the last read -- kind, start and end in milliseconds relative to the epoch of ANY context of the same device -- up to `cap`
 * entries, instead of bpgpu_profile_read's sums.  The union of the intervals of all contexts is the time the GPU was busy:
 * sums of overlapping launches exceed the wall clock, their union cannot. */
void *bpgpu_profile_epoch(bpgpu_ctx *ctx);
int bpgpu_profile_intervals(bpgpu_ctx *ctx, void *epoch, size_t cap, int32_t *kind, double *start_ms, double *end_ms, size_t *count);

/* device memory plumbing */
int bpgpu_malloc(bpgpu_ctx *ctx, size_t bytes, void **dptr);
int bpgpu_free(bpgpu_ctx *ctx, void *dptr);
int bpgpu_upload(bpgpu_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int bpgpu_download(bpgpu_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* the same, enqueued on the context's stream without waiting: the host buffer must stay alive (and should be page-locked:
 * bpgpu_host_alloc) until bpgpu_sync.  A caller streaming batches through `_dev` entry points overlaps the upload of
 * batch i+1 with the kernels of batch i this way (bench.py `h2d_inclusive`). */
int bpgpu_upload_async(bpgpu_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int bpgpu_download_async(bpgpu_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* Page-locked host memory for staging.  Operands handed over from such a buffer reach the device by DMA at PCIe
 * speed; pageable buffers go through the runtime's bounce buffers (3-10 GB/s measured).  Optional -- every entry point
 * accepts ordinary host pointers (the Rust side would keep its packed scalar / point staging Vec in one of these).
 * The memory belongs to the process, not to a context.  BPGPU_E_DEVICE without a HIP device, BPGPU_E_OOM. */
int bpgpu_host_alloc(size_t bytes, void **out);
void bpgpu_host_free(void *p);

/* ---- scalar field --------------------------------------------------------------------------
 * Scalar::batch_inverse(&mut [Scalar])   -- src/inner_product_proof.rs:283
 * Scalar::inverse()                      -- src/inner_product_proof.rs:123,181; r1cs/prover.rs:593; r1cs/verifier.rs:468
 * In place.  A zero element -> BPGPU_E_ARG (mpc-stark/ark-ff leave zeros untouched; the reference
 * only passes non-zero challenges). */
int bpgpu_batch_inverse(bpgpu_ctx *ctx, uint8_t *scalars, size_t n);
/* inner_product(a, b)                    -- src/inner_product_proof.rs:463-472 (panics on length
 * mismatch: here the single n makes that unrepresentable) */
int bpgpu_inner_product(bpgpu_ctx *ctx, const uint8_t *a, const uint8_t *b, size_t n, uint8_t out[32]);

/* ---- multi-scalar multiplication ------------------------------------------------------------
 * StarkPoint::msm_iter(scalars, points) / StarkPoint::msm(&[..], &[..])
 *   -- src/r1cs/prover.rs:465,477,485,535,546,555; src/inner_product_proof.rs:90,103,159,166,226-227,353;
 *      src/r1cs/verifier.rs:516.  Accepts identity points, zero scalars, duplicate points, n == 0. */
int bpgpu_msm(bpgpu_ctx *ctx, const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[64]);
/* nb independent MSMs of n terms each (term-major within an MSM); out = nb points */
int bpgpu_msm_batch(bpgpu_ctx *ctx, size_t nb, size_t n, const uint8_t *scalars, const uint8_t *points,
                    uint8_t *out);
/* bpgpu_msm_batch with device-resident operands and result (boundary encodings in HBM, e.g. from bpgpu_upload):
 * asynchronous on the context's stream; malformed operands raise the context's input flag (bpgpu_input_flag)
 * instead of an error code.  For callers that keep points resident between MSMs. */
int bpgpu_msm_batch_dev(bpgpu_ctx *ctx, size_t nb, size_t n, const void *scalars_dev, const void *points_dev,
                        void *out_dev);
/* sum of n points, no scalars (StarkPoint + StarkPoint): the local reduction of the <= 8 partial results that the GPUs of
 * a node all-gather after a term-range-sharded MSM or a combined batch check -- one launch instead of an
 * MSM with unit scalars (whose 252-doubling chain costs ~1 ms however few the terms). */
int bpgpu_points_sum(bpgpu_ctx *ctx, const uint8_t *points, size_t n, uint8_t out[64]);
/* ---- arkworks in-memory forms: zero-copy hand-over from the Rust host --------------------------------
 * mpc-stark's Scalar is ark_ff::Fp256<MontBackend<_, 4>>: four little-endian u64 limbs holding x * 2^256 mod n (32 bytes as
 * they lie in memory); its StarkPoint is ark_ec::short_weierstrass::Projective: Jacobian (X : Y : Z), the point is
 * (X / Z^2, Y / Z^3), each coordinate four u64 limbs holding c * 2^256 mod p (96 bytes), identity Z = 0.  These entry points take
 * and return exactly those bytes: no de-Montgomery / serialisation / inversion on the host (one Montgomery multiplication per
 * element on the device instead).  Non-canonical limbs (>= modulus) and points off the curve -> BPGPU_E_ARG.
 *   bpgpu_msm_ark            StarkPoint::msm / msm_iter with operands and result in arkworks form
 *                            -- verifier.rs:516-547, prover.rs:465-564, inner_product_proof.rs:90-172
 *   bpgpu_scalars_from_ark / _to_ark, bpgpu_points_from_ark / _to_ark: convert whole vectors to / from the boundary encodings
 *                            (32-byte LE canonical scalars, 64-byte affine x || y) that every other entry point takes. */
int bpgpu_msm_ark(bpgpu_ctx *ctx, const uint8_t *scalars_mont, const uint8_t *points_jac_mont, size_t n,
                  uint8_t out_jac_mont[96]);
int bpgpu_scalars_from_ark(bpgpu_ctx *ctx, const uint8_t *scalars_mont, size_t n, uint8_t *scalars_le);
int bpgpu_scalars_to_ark(bpgpu_ctx *ctx, const uint8_t *scalars_le, size_t n, uint8_t *scalars_mont);
int bpgpu_points_from_ark(bpgpu_ctx *ctx, const uint8_t *points_jac_mont, size_t n, uint8_t *points_xy);
int bpgpu_points_to_ark(bpgpu_ctx *ctx, const uint8_t *points_xy, size_t n, uint8_t *points_jac_mont);
/* nsets MSMs over ONE point vector: out[s] = sum_i scalars[s*n + i] * points[i].  This is the local work of
 * StarkPoint::msm_authenticated_iter in the two-party prover -- one MSM each over the secret shares, the MACs and
 * the public modifiers of the same authenticated scalars against the same points (r1cs_mpc/mpc_prover.rs:621-657,
 * 717-750; r1cs_mpc/mpc_inner_product.rs:104-126,172-186): the points are validated and converted
 * once.  scalars: nsets x n x 32 B; points: n x 64 B; out: nsets x 64 B. */
int bpgpu_msm_shared(bpgpu_ctx *ctx, size_t nsets, size_t n, const uint8_t *scalars, const uint8_t *points,
                     uint8_t *out);

/* ---- resident generators ---------------------------------------------------------------------
 * BulletproofGens::share(0).G(n) / .H(n) and PedersenGens{B, B_blinding}
 *   -- src/generators.rs:32-37,158-167,310-320.  Uploads the points once and precomputes signed
 * fixed-window tables (window_bits in {4, 8, 10, 12, 14, 16, 20}; table bytes = (2*cap+2) * (252/c+1) * 2^(c-1) * 64). */
int bpgpu_gens_create(bpgpu_ctx *ctx, const uint8_t *G, const uint8_t *H, size_t gens_capacity,
                      const uint8_t B[64], const uint8_t B_blinding[64], int window_bits,
                      bpgpu_gens **out);
void bpgpu_gens_destroy(bpgpu_ctx *ctx, bpgpu_gens *g);
size_t bpgpu_gens_capacity(const bpgpu_gens *g);
/* sum_i s_i * X_i over the resident set, nb independent scalar vectors.  Layout of one vector:
 * [B, B_blinding, G_0..G_{n-1}, H_0..H_{n-1}] (2 + 2n scalars), n <= capacity.
 *   -- the generator part of prover.rs:465-494 and verifier.rs:525-530,541-544 */
int bpgpu_msm_gens(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *scalars,
                   uint8_t *out);
/* the same with the scalars as they lie in the reference's memory (ark-ff Montgomery limbs, x * 2^256 mod n, 4 x u64 little
 * endian; see the arkworks section below): converted on the device, no de-Montgomery per scalar on the host.  The 256
 * lock-step provers of BASELINE configs[2] hand over 1.3 M witness and blinding scalars per batch this way. */
int bpgpu_msm_gens_ark(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *scalars_ark,
                       uint8_t *out);

/* ---- inner-product argument ------------------------------------------------------------------
 * InnerProductProof::fold_witness  -- src/inner_product_proof.rs:202-248
 * Inputs hold [L half || R half] (2n entries each); outputs n entries:
 *   a' = a_L u + u^-1 a_R ; b' = b_L u^-1 + u b_R ; G' = u^-1 G_L + u G_R ; H' = u H_L + u^-1 H_R */
int bpgpu_fold_witness(bpgpu_ctx *ctx, size_t n, const uint8_t u[32], const uint8_t u_inv[32],
                       const uint8_t *a, const uint8_t *b, const uint8_t *G, const uint8_t *H,
                       uint8_t *a_out, uint8_t *b_out, uint8_t *G_out, uint8_t *H_out);
/* arithmetic half of InnerProductProof::verification_scalars -- src/inner_product_proof.rs:280-309
 * challenges u_1..u_k (creation order) -> u_sq[k], u_inv_sq[k], s[n], n == 2^k else BPGPU_E_LEN
 * (the transcript replay of :271-278 stays on the host). */
int bpgpu_verification_scalars(bpgpu_ctx *ctx, const uint8_t *challenges, size_t k, size_t n,
                               uint8_t *u_sq, uint8_t *u_inv_sq, uint8_t *s);

/* InnerProductProof::create -- src/inner_product_proof.rs:49-193 -- split at the Fiat-Shamir transcript
 * (which stays on the host) for nb independent proofs advancing in lock-step; all state stays in HBM
 * between rounds.  n must be a power of two (reference asserts, :70).  G/H: n points shared by all
 * proofs (shared_gens = 1) or nb x n.  Protocol per round while bpgpu_ipp_len() > 1:
 *   bpgpu_ipp_round -> L[nb], R[nb]   (c_L, c_R :87-88,156-157 and the two MSMs :90-114,159-172;
 *                                       the first round folds G_factors/H_factors into the scalars)
 *   host: append L, R; u = challenge; u_inv = u.inverse()              (:119-123,177-181)
 *   bpgpu_ipp_fold(u[nb], u_inv[nb])  (first round also scales the generators, :125-134; then
 *                                       fold_witness :135-146,183-184)
 * then bpgpu_ipp_finish -> a[nb], b[nb] (:187-192). */
typedef struct bpgpu_ipp bpgpu_ipp;
int bpgpu_ipp_begin(bpgpu_ctx *ctx, size_t nb, size_t n, const uint8_t *Q, const uint8_t *G_factors,
                    const uint8_t *H_factors, const uint8_t *G, const uint8_t *H, int shared_gens,
                    const uint8_t *a, const uint8_t *b, bpgpu_ipp **out);
/* The same session over RESIDENT generators (G, H = the first n of `g`, Q = w * B with B = g's Pedersen base:
 * exactly what r1cs/prover.rs:687-708 passes).  No generator is ever folded: round j's L, R are fixed-base
 * MSMs over the original generators with the accumulated challenge products folded into the scalars, so a
 * round costs table lookups instead of two 252-doubling chains.  Same outputs as bpgpu_ipp_begin.
 * w: nb x 32 B. */
int bpgpu_ipp_begin_gens(bpgpu_ctx *ctx, const bpgpu_gens *g, size_t nb, size_t n, const uint8_t *w,
                         const uint8_t *G_factors, const uint8_t *H_factors, const uint8_t *a, const uint8_t *b,
                         bpgpu_ipp **out);
/* All remaining rounds of a session with the transcript on the DEVICE (SURVEY 8f N1 applied to the prover): per round
 * L, R, transcript.append_point("L" / "R"), challenge_scalar("u") (inner_product_proof.rs:119-123, 177-181), u^-1 and
 * the fold, back to back on the stream -- no host round trip per round.  states_in: the provers' 32-byte hash-chain
 * states after innerproduct_domain_sep (:72); L_out / R_out: nb x k x 64 B (proof-major); a_out, b_out: nb x 32 B;
 * states_out (optional): the chain states afterwards.  The hash chain is the build's stand-in for merlin's. */
int bpgpu_ipp_run_fs(bpgpu_ctx *ctx, bpgpu_ipp *s, const uint8_t *states_in, uint8_t *L_out, uint8_t *R_out,
                     uint8_t *a_out, uint8_t *b_out, uint8_t *states_out);
void bpgpu_ipp_destroy(bpgpu_ctx *ctx, bpgpu_ipp *s);
size_t bpgpu_ipp_len(const bpgpu_ipp *s);
int bpgpu_ipp_round(bpgpu_ctx *ctx, bpgpu_ipp *s, uint8_t *L, uint8_t *R);
int bpgpu_ipp_fold(bpgpu_ctx *ctx, bpgpu_ipp *s, const uint8_t *u, const uint8_t *u_inv);
int bpgpu_ipp_finish(bpgpu_ctx *ctx, bpgpu_ipp *s, uint8_t *a_out, uint8_t *b_out);
/* Resident-generator sessions only, after the last fold (bpgpu_ipp_len == 1): the folded generators as points,
 * G' = sum_i cG[i] G_i and H' = sum_i cH[i] H_i (nb x 64 B each) -- the pair the final (a, b) refers to.  A rank of a
 * vector-sharded IPP (one proof's a, b, G, H dealt cyclically over the GPUs of a node: SURVEY 8e.2,
 * mpc_bulletproof_amd/sharding.py sharded_ipp_create) hands (a, b, G', H') to the last log2(ranks) rounds. */
int bpgpu_ipp_folded_gens(bpgpu_ctx *ctx, bpgpu_ipp *s, uint8_t *G_out, uint8_t *H_out);

/* ---- R1CS ------------------------------------------------------------------------------------
 * Constraint rows as the reference holds them (Vec<LinearCombination>, r1cs/prover.rs:31,
 * r1cs/verifier.rs:38): CSR with row_ptr[q+1]; per term kind (0 MultiplierLeft, 1 MultiplierRight,
 * 2 MultiplierOutput, 3 Committed, 4 One -- linear_combination.rs:15-28), index, coefficient. */
int bpgpu_circuit_create(bpgpu_ctx *ctx, size_t q, const uint32_t *row_ptr, const uint32_t *kind,
                         const uint32_t *idx, const uint8_t *coeff, size_t n_multipliers,
                         size_t m_commitments, bpgpu_circuit **out);
void bpgpu_circuit_destroy(bpgpu_ctx *ctx, bpgpu_circuit *c);
/* Prover/Verifier::flattened_constraints(z) -- r1cs/prover.rs:342-379, r1cs/verifier.rs:323-362.
 * nb challenges z (one per proof) -> wL,wR,wO (nb x n), wV (nb x m), wc (nb) ; wc may be NULL (prover). */
int bpgpu_flatten_constraints(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *z,
                              uint8_t *wL, uint8_t *wR, uint8_t *wO, uint8_t *wV, uint8_t *wc);

/* bpgpu_circuit_create with the coefficients as they lie in the reference's memory (ark-ff Montgomery limbs, x * 2^256 mod n):
 * converted on the device.  The 2^14-shuffle's ~200 000 coefficients cost the host 8 ms of de-Montgomery per upload otherwise. */
int bpgpu_circuit_create_ark(bpgpu_ctx *ctx, size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx,
                             const uint8_t *coeff_ark, size_t n_multipliers, size_t m_commitments, bpgpu_circuit **out);
/* Circuits with RANDOMIZED (second-phase) constraints whose coefficients are affine in the gadget's challenge
 * (RandomizableConstraintSystem::specify_randomized_constraints + cs.challenge_scalar(label) -- r1cs/verifier.rs:366-385,
 * r1cs/prover.rs:383-402; the shuffle gadget (x_i - z), tests/r1cs.rs:23-62): coefficient of a term = c0 + sum_j chi_j * c_j.
 * The CSR has (1 + nchi) * q rows: rows [0, q) hold the constant parts c0, rows [j q, (j + 1) q) the chi_j parts (row r of block j
 * belongs to constraint r).  One such circuit serves every proof of the gadget, whatever challenge its transcript produced.
 * Use it with bpgpu_r1cs_verify_batch_param (gadget challenges from the host) or bpgpu_r1cs_verify_batch_fs2 (transcript on the
 * device); the other entry points reject it (BPGPU_E_ARG).  nchi <= 8 (the device transcript supports nchi <= 1). */
int bpgpu_circuit_create_param(bpgpu_ctx *ctx, size_t q, size_t nchi, const uint32_t *row_ptr, const uint32_t *kind,
                               const uint32_t *idx, const uint8_t *coeff, size_t n_multipliers, size_t m_commitments,
                               bpgpu_circuit **out);
/* Prover::prove arithmetic between the y,z and the u,x challenges -- r1cs/prover.rs:587-619:
 * flattened_constraints(z), exp_y / exp_y_inv, the l(x)/r(x) coefficient vectors and
 * t_1..t_6 = VecPoly3::special_inner_product (util.rs:152-170), for nb provers of ONE circuit.
 * Inputs proof-major: y, y_inv, z (nb); a_L, a_R, a_O, s_L, s_R (nb x n, n = multipliers).
 * Outputs: t_coeffs nb x 6 (t1 t2 t3 t4 t5 t6), wV nb x m (for t_2_blinding, prover.rs:644-648).
 * The coefficient vectors stay in HBM in *out for bpgpu_r1cs_prover_eval -- prover.rs:659-672:
 * l_vec, r_vec (nb x padded_n) = l(x), r(x) with the zero / -y^i padding. */
typedef struct bpgpu_prover bpgpu_prover;
int bpgpu_r1cs_prover_polys(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *y,
                            const uint8_t *y_inv, const uint8_t *z, const uint8_t *a_L, const uint8_t *a_R,
                            const uint8_t *a_O, const uint8_t *s_L, const uint8_t *s_R, uint8_t *t_coeffs,
                            uint8_t *wV, bpgpu_prover **out);
/* the same with every INPUT scalar (y, y_inv, z, a_L, a_R, a_O, s_L, s_R) in ark-ff Montgomery form; outputs as above */
int bpgpu_r1cs_prover_polys_ark(bpgpu_ctx *ctx, const bpgpu_circuit *c, size_t nb, const uint8_t *y,
                                const uint8_t *y_inv, const uint8_t *z, const uint8_t *a_L, const uint8_t *a_R,
                                const uint8_t *a_O, const uint8_t *s_L, const uint8_t *s_R, uint8_t *t_coeffs,
                                uint8_t *wV, bpgpu_prover **out);
int bpgpu_r1cs_prover_eval(bpgpu_ctx *ctx, bpgpu_prover *s, size_t padded_n, const uint8_t *x,
                           uint8_t *l_vec, uint8_t *r_vec);
/* prover.rs:659-708 in one call with everything staying in HBM: evaluates l(x), r(x) (with the padding), forms
 * G_factors = [1; n1] ++ [u; padded_n - n1] and H_factors = y^-i * G_factors (prover.rs:689-697) and opens the
 * resident-generator IPP session (bpgpu_ipp_begin_gens) on them with Q = w * B.  x, u, y_inv, w: nb x 32 B.
 * Continue with bpgpu_ipp_round / fold / finish. */
int bpgpu_r1cs_prover_ipp_begin(bpgpu_ctx *ctx, bpgpu_prover *s, const bpgpu_gens *g, size_t padded_n, size_t n1,
                                const uint8_t *x, const uint8_t *u, const uint8_t *y_inv, const uint8_t *w,
                                bpgpu_ipp **out);
void bpgpu_prover_destroy(bpgpu_ctx *ctx, bpgpu_prover *s);
/* Resident-witness prover sessions: the same Prover::prove arithmetic with the witness crossing the bus ONCE and, optionally, the
 * blinding vectors never crossing it at all (256 provers x 1024 multipliers: 25 MB up instead of 92 MB, no 525 000 host-side
 * scalar draws).  All scalars in ark-ff Montgomery form unless stated.
 *
 * bpgpu_r1cs_prover_commit -- r1cs/prover.rs:457-494 (first call: *session == NULL, phase-1 multipliers) and :519-565 (second call
 *   on the same session: the n_new multipliers the randomized constraints added; skip it when there are none -- the reference sets
 *   A_I2 = A_O2 = S2 = identity, :566-576).  a_L, a_R, a_O: nb x n_new.  Blinding vectors s_L, s_R (:461-462, :526-527): either
 *   explicit (nb x n_new each; vector_keys NULL) or drawn on the device from vector_keys (nb x 32 raw bytes the caller takes from
 *   its RNG where the reference draws the vectors; s_L = s_R = NULL): "BlindVec v1",
 *     block(key, v, j) = first 128 bytes of Keccak-f[1600] over the padded 48-byte message key || u64le(v) || u64le(j)   (v = 0: s_L, 1: s_R)
 *     s_v[i] = int_LE(block(key, v, i / 2)[64 (i mod 2) .. +64]) mod n
 *   (the CPU oracle restates the stream, so such proofs replay under a test RNG).  blindings: nb x 3 (i_blinding, o_blinding,
 *   s_blinding).  commitments out: nb x 3 x 64 B (A_I, A_O, S).  BPGPU_E_GENS when the multipliers exceed the generators.
 * bpgpu_r1cs_prover_session_polys -- :587-619 on the session's planes: y^-1 (:593, on the device), flattened constraints, l / r
 *   coefficient vectors, t_1..t_6.  y, z: nb x 32 B canonical LE (fresh transcript challenges); outputs as bpgpu_r1cs_prover_polys.
 *   Continue with bpgpu_r1cs_prover_ipp_begin, whose y_inv may then be NULL (the session's own). */
int bpgpu_r1cs_prover_commit(bpgpu_ctx *ctx, const bpgpu_gens *g, bpgpu_prover **session, size_t nb, size_t n_new,
                             const uint8_t *a_L, const uint8_t *a_R, const uint8_t *a_O, const uint8_t *s_L, const uint8_t *s_R,
                             const uint8_t *vector_keys, const uint8_t *blindings, uint8_t *commitments);
int bpgpu_r1cs_prover_session_polys(bpgpu_ctx *ctx, bpgpu_prover *s, const bpgpu_circuit *c, const uint8_t *y, const uint8_t *z,
                                    uint8_t *t_coeffs, uint8_t *wV);
/* The same for a circuit of bpgpu_circuit_create_param (second-phase constraints affine in the gadget challenges, prover.rs:383-402):
 * gadget_challenges = nb x nchi x 32 bytes, the values the provers' transcripts produced.  The rows of such a circuit cross the ABI
 * once per circuit shape; a prover of the 2^14-shuffle no longer builds and uploads 65 533 constraint rows per proof. */
int bpgpu_r1cs_prover_session_polys_param(bpgpu_ctx *ctx, bpgpu_prover *session, const bpgpu_circuit *c, const uint8_t *y, const uint8_t *z,
                                          const uint8_t *gadget_challenges, uint8_t *t_coeffs, uint8_t *wV);
/* out[i] = scalars[i] * (curve generator) -- GeneratorsChain::next (generators.rs:112-124),
 * Q = w * B (prover.rs:687), PedersenGens::commit with B = B_blinding (generators.rs:41-43,61-70) */
int bpgpu_generator_mul(bpgpu_ctx *ctx, const uint8_t *scalars, size_t n, uint8_t *out);

/* Point wire codec: StarkPoint::from_bytes / to_bytes at r1cs/proof.rs:90-108,150-155 and
 * inner_product_proof.rs:391-392,440-445.  The 32-byte form lives in the absent crate mpc-stark; restated as
 * arkworks' compressed short-Weierstrass encoding: x little-endian, bit 7 of byte 31 = "y is the larger of (y, -y)",
 * bit 6 = point at infinity (parity unpinned).  Decompression takes a square root in F_p
 * (2-adicity 192) per point on the device.  ok[i] = 1 iff encoding i is valid (canonical x on the curve, not both
 * flags); invalid ones decode to 64 zero bytes -- the caller maps them to FormatError.  xy: n x 64 B boundary form. */
int bpgpu_points_decompress(bpgpu_ctx *ctx, const uint8_t *compressed, size_t n, uint8_t *xy, int32_t *ok);
/* BPGPU_E_ARG when a point is not canonical / not on the curve */
int bpgpu_points_compress(bpgpu_ctx *ctx, const uint8_t *xy, size_t n, uint8_t *compressed);

/* Batched Verifier::verify arithmetic -- r1cs/verifier.rs:457-553 for nb proofs of ONE circuit.
 * Per proof p (all arrays proof-major):
 *   points     : (11 + m + 2k) x 64 B : A_I1 A_O1 S1 A_I2 A_O2 S2 | V_0..V_{m-1} | T_1 T_3 T_4 T_5 T_6 | L_0..L_{k-1} | R_0..R_{k-1}
 *   scalars    : 5 x 32 B             : t_x t_x_blinding e_blinding a b
 *   challenges : (6 + k) x 32 B       : y z u x w r u_1..u_k   (from the host transcript, verifier.rs:432-455,506)
 * n1 = phase-1 multipliers (verifier.rs:400); padded_n = 2^k.
 * Outputs: ok[p] = 1 iff mega_check is the identity (verifier.rs:549); mega (optional, nb x 64 B) the
 * mega_check point itself; msm_scalars (optional, nb x nterms x 32 B) the scalars of verifier.rs:517-532
 * in that order, nterms = 13 + m + 2*padded_n + 2k.
 * The identity checks of transcript.rs:101-113 belong to the host transcript replay. */
int bpgpu_r1cs_verify_batch(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb,
                            size_t n1, size_t k, const uint8_t *points, const uint8_t *scalars,
                            const uint8_t *challenges, int32_t *ok, uint8_t *mega, uint8_t *msm_scalars);
/* same with device-resident inputs/outputs (mega_dev / msm_scalars_dev may be NULL); asynchronous */
int bpgpu_r1cs_verify_batch_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb,
                                size_t n1, size_t k, const void *points_dev, const void *scalars_dev,
                                const void *challenges_dev, void *ok_dev, void *mega_dev,
                                void *msm_scalars_dev);

/* A STREAM of batches in ONE call on ONE context: the same per-proof verification for any number of proofs of one circuit.  The
 * library cuts them into batches of BPGPU_OPT_STREAM_BATCH proofs (1024) that take turns on a ring of BPGPU_OPT_STREAM_LANES (20)
 * internal streams + workspaces, so that the kernel chains of ~20 batches overlap on the GPU -- the pipelined rate (4 M
 * verifications/s for the 64-bit range gadget) without the caller creating contexts or exporting GPU_MAX_HW_QUEUES (libbpgpu.so
 * sets it to 24 when it is loaded, unless the process has set it; a process that has initialised HIP before loading the library
 * keeps the runtime's 4 queues and about a third of the rate).  This is what a Rust host's loop of Verifier::verify
 * (verifier.rs:393) becomes: collect the proofs of one circuit, one call.
 *   _dev : operands and ok[] resident in HBM (layouts of bpgpu_r1cs_verify_batch); asynchronous -- the lanes fork from the
 *          context's stream and are joined back into it: bpgpu_sync, or the next call on this context, waits for all of them.
 *   host : operands and ok[] in host memory (page-locked from bpgpu_host_alloc for DMA speed): each batch's upload, kernels and
 *          verdict download run on its lane, so the copies of one batch overlap the kernels of the others; returns when done. */
int bpgpu_r1cs_verify_stream_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                 const void *points_dev, const void *scalars_dev, const void *challenges_dev, void *ok_dev);
int bpgpu_r1cs_verify_stream(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                             const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges, int32_t *ok);

/* The same with the Fiat-Shamir transcript replayed ON THE DEVICE for circuits without
 * randomized constraints: instead of challenges the caller passes, per proof, the 32-byte hash-chain state it
 * holds when it would call Verifier::new (after Transcript::new(label) + any application preamble).  The
 * device performs verifier.rs:271,303,398-455,506 and inner_product_proof.rs:269-278 (keccak256, LE
 * absorption, hash_to_scalar util.rs:252-267), rejects identity points where the reference validates
 * (transcript.rs:101-113 -> ok = 0), then proceeds as bpgpu_r1cs_verify_batch.  challenges_out (optional):
 * nb x (6 + k) x 32 B, y z u x w r u_1..u_k.  The hash chain is the build's stand-in for merlin's
 * HashChainTranscript (absent from the reference tree; parity unpinned -- DESIGN.md). */
int bpgpu_r1cs_verify_batch_fs(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                               size_t k, const uint8_t *init_states, const uint8_t *points, const uint8_t *scalars,
                               int32_t *ok, uint8_t *mega, uint8_t *challenges_out);
int bpgpu_r1cs_verify_batch_fs_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb,
                                   size_t n1, size_t k, const void *init_states_dev, const void *points_dev,
                                   const void *scalars_dev, void *ok_dev, void *mega_dev, void *challenges_out_dev);

/* The whole of Verifier::verify from the reference's WIRE format, on the device: R1CSProof::from_bytes
 * (r1cs/proof.rs:128-207: version byte, 8 or 11 compressed points, 3 big-endian scalars, k (L, R) pairs, a, b),
 * point decompression, the transcript replay and the verification of bpgpu_r1cs_verify_batch_fs, for nb proofs of
 * one circuit without randomized constraints that all have the same length (the length fixes version and k).
 * proofs: nb x proof_len B; commitments: nb x m x 32 B compressed (as a node receives them); init_states: nb x 32 B.
 * ok[p] = 1 iff proof p decodes (otherwise the reference returns FormatError) and verifies.  BPGPU_E_LEN when
 * proof_len is not a valid proof size. */
int bpgpu_r1cs_verify_batch_wire(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                 size_t proof_len, const uint8_t *proofs, const uint8_t *commitments,
                                 const uint8_t *init_states, int32_t *ok);
int bpgpu_r1cs_verify_batch_wire_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                     size_t proof_len, const void *proofs_dev, const void *commitments_dev,
                                     const void *init_states_dev, void *ok_dev);

/* Verifier::verify for nb proofs of ONE two-phase circuit (bpgpu_circuit_create_param).  _param: as bpgpu_r1cs_verify_batch with the
 * gadget challenges (nb x nchi x 32 B, in the order the gadget drew them) beside the usual 6 + k challenges.  _fs2: as
 * bpgpu_r1cs_verify_batch_fs -- the whole transcript replay on the device, including the phase separator and the gadget's
 * challenge_scalar(gadget_label) (label zero-padded to 32 bytes; verifier.rs:373-383), whose value selects the constraint weights;
 * gadget_challenges_out (optional) returns it.  Same per-proof rejection rules as the one-phase entry points. */
int bpgpu_r1cs_verify_batch_param(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1,
                                  size_t k, const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges,
                                  const uint8_t *gadget_challenges, int32_t *ok, uint8_t *mega, uint8_t *msm_scalars);
int bpgpu_r1cs_verify_batch_fs2(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                const uint8_t *init_states, const uint8_t gadget_label[32], const uint8_t *points,
                                const uint8_t *scalars, int32_t *ok, uint8_t *mega, uint8_t *challenges_out,
                                uint8_t *gadget_challenges_out);
int bpgpu_r1cs_verify_batch_fs2_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                    const void *init_states_dev, const uint8_t gadget_label[32], const void *points_dev,
                                    const void *scalars_dev, void *ok_dev, void *mega_dev, void *challenges_out_dev,
                                    void *gadget_challenges_out_dev);
/* ONE large proof over the GPUs of a node (BASELINE configs[3], the 2^14-shuffle): every rank runs the protocol and
 * the O(n) scalar work in full, and the multi-scalar multiplications over ITS contiguous share of the terms; the ranks all-gather
 * their partial points (64 bytes each) and add them with bpgpu_points_sum -- the only exchange on the path.
 *   bpgpu_set_shard(ctx, rank, world): prover / IPP sessions opened on this context afterwards return PARTIAL sums --
 *     bpgpu_r1cs_prover_commit: A_I, A_O, S over the rank's generators (the B_blinding term on rank 0);
 *     bpgpu_ipp_round of a resident-generator session: L, R over the rank's generators (the c Q term on rank 0); the scalar
 *     vectors are folded on every rank alike; bpgpu_ipp_run_fs refuses such a session (BPGPU_E_ARG: the rounds need the exchange).
 *     world = 1 (the default) turns it off.
 *   bpgpu_r1cs_verify_shard: the rank's partial mega_check point of one proof (layouts of bpgpu_r1cs_verify_batch with nb = 1;
 *     gadget_challenges for a parametric circuit, else NULL).  The proof verifies iff the sum over the ranks is the identity.
 *     A malformed operand (off-curve or non-canonical point, non-canonical scalar / challenge) is seen only by the rank whose
 *     share holds it, and the verdict has to be collective: that rank returns BPGPU_OK with the POISON encoding in partial_xy
 *     (64 bytes 0xFF, which is not a point), so that bpgpu_points_sum over the gathered partials fails with BPGPU_E_ARG on
 *     every rank alike -- no rank is left waiting in the all-gather.  Shape errors (BPGPU_E_LEN / _GENS) are the same on all ranks. */
int bpgpu_set_shard(bpgpu_ctx *ctx, size_t rank, size_t world);
int bpgpu_r1cs_verify_shard(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t n1, size_t k, const uint8_t *points,
                            const uint8_t *scalars, const uint8_t *challenges, const uint8_t *gadget_challenges, size_t rank, size_t world,
                            uint8_t partial_xy[64]);

/* Combined batch check (NOT a reference API -- the reference verifies proof by proof; this is
 * the usual verifier-service batching and BASELINE.json's "single big MSM"): with caller-chosen random
 * weights rho (nb x 32 B, e.g. from a CSPRNG) computes  sum_p rho_p * mega_check_p  as ONE point:
 * one fixed-base MSM over the generators with scalars sum_p rho_p s_{p,g} plus one bucket-method MSM over
 * the nb*(11+m+2k) proof points (a ZERO weight drops its proof from the check: the weights must be random and non-zero; the
 * screened entry points below test for it).  All nb proofs are valid iff the point (summed over all GPUs: all-gather
 * of the 64-byte partials over RCCL, then a local add) is the identity.  Same input layout as
 * bpgpu_r1cs_verify_batch. */
int bpgpu_r1cs_verify_combined(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb,
                               size_t n1, size_t k, const uint8_t *points, const uint8_t *scalars,
                               const uint8_t *challenges, const uint8_t *rho, uint8_t partial_xy[64]);
int bpgpu_r1cs_verify_combined_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb,
                                   size_t n1, size_t k, const void *points_dev, const void *scalars_dev,
                                   const void *challenges_dev, const void *rho_dev, void *partial_xy_dev);
/* SCREENED stream: per-proof accept bits at (nearly) the combined check's price when the proofs are valid -- the usual shape of
 * a verifier service.  The call cuts the proofs into batches of BPGPU_OPT_SCREEN_BATCH proofs (2560: the combined check has a long
 * kernel chain and little work per proof, so its batches are larger than the per-proof path's), runs the combined check
 * sum_p rho_p * mega_check_p of every batch first (rho: nb x 32 B of caller-chosen random non-zero weights, unpredictable to
 * the provers, as for bpgpu_r1cs_verify_combined), sets ok[p] = 1 for every proof of a batch whose point is the identity and that
 * holds no malformed input and no zero weight (a zeroed rho buffer must not accept anything), and runs the per-proof verification
 * only for the other batches, so that ok[] is what
 * bpgpu_r1cs_verify_stream returns (up to the 2^-250 chance that random weights cancel an invalid proof).  One host-side wait
 * between the two phases (68 bytes per batch are read back).  fallback_batches (optional): how many batches took the per-proof
 * path.  No reference API (the reference verifies proof by proof); built from verifier.rs:457-553. */
int bpgpu_r1cs_verify_screened(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                               const uint8_t *points, const uint8_t *scalars, const uint8_t *challenges, const uint8_t *rho, int32_t *ok,
                               size_t *fallback_batches);
/* the same with operands, weights and ok[] resident in HBM; ok_dev is complete when bpgpu_sync (or the next call on the context)
 * returns -- the call itself waits once, between its two phases */
int bpgpu_r1cs_verify_screened_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                   const void *points_dev, const void *scalars_dev, const void *challenges_dev, const void *rho_dev,
                                   void *ok_dev, size_t *fallback_batches);
/* the whole of Verifier::verify that way: the transcript replayed on the device (bpgpu_r1cs_verify_batch_fs: init_states = the
 * 32-byte chain state per proof, 1-phase circuits) per batch, then the batch's combined check, the per-proof path (again with the
 * device transcript) only for a batch that fails either; ok[] is what bpgpu_r1cs_verify_batch_fs returns */
int bpgpu_r1cs_verify_screened_fs_dev(bpgpu_ctx *ctx, const bpgpu_gens *g, const bpgpu_circuit *c, size_t nb, size_t n1, size_t k,
                                      const void *init_states_dev, const void *points_dev, const void *scalars_dev, const void *rho_dev,
                                      void *ok_dev, size_t *fallback_batches);

#ifdef __cplusplus
}
#endif
#endif
