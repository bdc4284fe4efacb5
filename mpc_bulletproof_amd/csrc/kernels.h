// kernels.h -- host-callable launchers of the HIP kernels (k_ec.hip, k_scalar.hip).
// All pointers are device pointers; every launcher enqueues on `st` and returns immediately.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bpk {

struct JacRaw { int32_t v[27]; };   // unpacked Jacobian limbs (X[9] Y[9] Z[9]), device scratch format
struct AffDev { uint32_t w[16]; };  // packed Montgomery affine (x[8] y[8]); zeros = identity
struct Words8 { uint32_t w[8]; };   // one 256-bit field element (plain canonical at the boundary)

// ---- points ------------------------------------------------------------------------------------
// boundary bytes -> device affine; sets *bad (int) to 1 on a non-canonical / off-curve point
// bad_unit (optional, zeroed by the caller): [i / per_unit] = 1 for a malformed point i (per-proof attribution)
void points_from_boundary(hipStream_t st, const Words8 *xy /*2 per point*/, AffDev *out, size_t n, int *bad,
                          int32_t *bad_unit = nullptr, size_t per_unit = 1);
// out = sum of n boundary points (validated; *bad |= 1 on a malformed one, which is skipped)
void points_sum(hipStream_t st, const Words8 *xy, size_t n, Words8 *out_xy, int *bad);
// JacRaw -> boundary bytes (one inversion per point)
void jac_to_boundary(hipStream_t st, const JacRaw *in, Words8 *xy_out, size_t n);
// JacRaw[n] -> device affine, Montgomery's trick in runs of `run` points per lane
void batch_normalize(hipStream_t st, const JacRaw *in, AffDev *out, size_t n, int run);

// out[i] = sum_{j<np} scalar_j(i) * point_j(i), np in {1..4} (the np points of a lane share one
// doubling chain: ~2 270 + 1 080 np field multiplications per lane instead of 3 350 np).  Two-level indexing for batched
// (proof-major) arrays: i = p * inner + r;
//   point_j(i)  = pts[j] + p * pt_outer[j] + r * pt_stride[j]          (AffDev units)
//   scalar_j(i) = sc[j]  + p * sc_outer[j] + r * sc_stride[j]          (u32 words; stride 0 = broadcast)
// inner == 0 means a flat array (p = 0, r = i).
struct StrausArgs {
  const AffDev *pts[4];
  size_t pt_stride[4];
  const uint32_t *sc[4];
  size_t sc_stride[4];
  size_t inner;
  size_t pt_outer[4];
  size_t sc_outer[4];
  size_t out_outer;   // with inner != 0: out[p * out_outer + r * out_stride]; 0 = dense (out[i])
  size_t out_stride;  // 0 = 1
  int from_boundary;  // pts are ABI bytes (x || y canonical LE words): validate + convert in the kernel, *bad |= 1 on failure
  int *bad;
  int32_t *bad_inner; // optional (zeroed by the caller): [r] = 1 when a point of inner index r (the proof, role-major) is malformed
  int prio;           // fused launch: raise the Straus waves' issue priority
};
// scratch: straus_scratch_bytes(np, n) bytes of device memory private to this launch until it completes
size_t straus_scratch_bytes(int np, size_t n);
void straus(hipStream_t st, int np, const StrausArgs &a, JacRaw *out, size_t n, void *scratch);

// Straus over the proof points (n_lanes lanes) and the small fixed-base MSMs (nb of them) in one launch; false =
// this (np, c, size) combination has no fused kernel, launch the two parts separately
bool verify_msm_fused(hipStream_t st, int np, const StrausArgs &a, JacRaw *out_var, size_t n_lanes, void *scratch, int c,
                      const AffDev *table, size_t n, size_t cap, const uint32_t *scalars, size_t sc_stride_words,
                      JacRaw *out_fixed, size_t nb);

// Window-parallel variant of the same (k_ec.hip): front [tables | inversion pass] -> (k_verify_scalars) -> windows -> groups ->
// back [Horner | fixed-base MSMs] -> verdict, which writes ok / mega itself (no verify_finalize).
// bad_sc: the per-proof canonicity bits written by verify_scalars (nullable)
// latency_mode: more, shorter lanes in the two longest launches (1 point per table lane, 32 lanes per fixed-base MSM, a quad per group in the first Horner stage): one batch
// alone finishes ~25 % sooner, a pipelined stream of batches runs ~5 % slower (more instructions)
// points_converted: points_abi holds AffDev rows (already validated and in Montgomery form: points_from_boundary) instead of ABI bytes
// table_np: points per table lane (1, 2, 4, 8; 0 = by mode: 4, or 1 in latency mode) -- BPGPU_OPT_TABLE_NP
struct VerifyWp { const AffDev *points_abi; size_t nb, nvar; void *scratch /* verify_wp_scratch_bytes */; int *bad; const int32_t *bad_sc; bool latency_mode; bool points_converted = false; int table_np = 0;
                  int horner_form = 0;   // Horner pass: 0 = by mode, 1 = a lane, 2 = a DPP quad, 3 = a wave (row form, ec29_row.cuh) per proof -- BPGPU_OPT_HORNER_FORM
                  size_t row_max = 1536; // most proofs / groups for which latency mode takes the row form -- BPGPU_OPT_HORNER_ROW_MAX
                  int fixed_lpm = 0;     // lanes per fixed-base MSM in the back launch: 16 / 32 / 64, 0 = by mode -- BPGPU_OPT_FIXED_LPM
                  int groups_form = 0;   // first Horner stage: 0 = by mode, 1 = a lane, 2 = a quad, 3 = a wave per group of 8 windows -- BPGPU_OPT_GROUPS_FORM
                  int fixed_chunk_gens = 0;   // generator half of the back launch: 0 = by mode, -1 = fixed_lpm lanes per proof + butterfly; g > 0 = a proof per lane, g generators per wave, partial sums added in the verdict launch -- BPGPU_OPT_FIXED_CHUNK_GENS
};
struct VerifyDims { size_t nb, n1, n, padded_n, k, m; const Words8 *chi; /* nb x nchi gadget challenges (plain words) or nullptr */
                    size_t vs_large_min = 0; /* padded_n / m from which the scalar assembly is split over the grid (0 = 4096) -- BPGPU_OPT_VS_LARGE_MIN */ };
struct VsPrepArgs;
size_t verify_wp_scratch_bytes(size_t nb, size_t nvar);
bool verify_wp_supported(size_t nb, size_t nvar, int c, size_t n);
// the launches below carve v.scratch up by (nb, nvar, table_np): true when that layout stays inside verify_wp_scratch_bytes(nb, nvar).
// The caller checks it once after sizing the buffer and fails the call (BPGPU_E_DEVICE) otherwise -- an internal sizing bug must
// not become an out-of-bounds write on the device, nor end the host process.
bool verify_wp_layout_fits(const VerifyWp &v);
// prep_* describe the inversion pass to fuse (vs_prep.cuh; prep_nb == 0: tables only)
// fast_* (all set): the inversion pass also does the serial part of a wave-sized proof's scalar assembly (vs_prep.cuh)
void verify_wp_front_launch(hipStream_t st, const VerifyWp &v, const VerifyDims &d, const Words8 *challenges, int32_t *aux,
                            size_t aux_stride, bool with_prep, const Words8 *fast_proof_scalars = nullptr, Words8 *fast_fixed_sc = nullptr,
                            Words8 *fast_var_sc = nullptr, Words8 *fast_full_sc = nullptr);
void verify_wp_windows(hipStream_t st, const VerifyWp &v, const uint32_t *var_scalars);
void verify_wp_groups(hipStream_t st, const VerifyWp &v);
void verify_wp_back(hipStream_t st, const VerifyWp &v, int c, const AffDev *table, size_t n, size_t cap,
                    const uint32_t *fixed_scalars, size_t sc_stride_words, JacRaw *out_fixed);
// `parts` partial sums of the generator half per proof in `fixed` (verify_wp_fixed_parts: 1 unless v.fixed_chunk_gens > 0 AND the generator
// half rode in verify_wp_back; out_fixed of verify_wp_back must then hold nb x parts entries)
size_t verify_wp_fixed_parts(const VerifyWp &v, size_t n);
void verify_wp_verdict(hipStream_t st, const VerifyWp &v, const JacRaw *fixed, int32_t *ok, Words8 *mega, size_t parts = 1);
// window sums of v's instances (v.nb = v2.nb * per consecutive ones per MSM) added up into v2's window sums: verify_wp_groups / _back then run on v2
void verify_wp_reduce_instances(hipStream_t st, const VerifyWp &v, const VerifyWp &v2, size_t per);
const JacRaw *verify_wp_varsum(const VerifyWp &v);   // nb sums of the proof-point halves, valid after verify_wp_back

// bucket-method MSM of one large instance: out = sum_i scalars[i] * pts[i]   (k_pip.hip)
int pippenger_window(size_t n);
size_t pippenger_scratch_bytes(size_t n, int c);
void pippenger(hipStream_t st, const AffDev *pts, const uint32_t *scalars, size_t n, int c, JacRaw *out, void *scratch);
// ninst independent instances of n terms each (instance-major arrays); out[inst * out_stride]
size_t pippenger_scratch_bytes_batch(size_t ninst, size_t n, int c);
void pippenger_batch(hipStream_t st, const AffDev *pts, const uint32_t *scalars, size_t ninst, size_t n, int c, JacRaw *out,
                     size_t out_stride, void *scratch);
// ---- k_pip2.hip: ONE mid-size instance in six launches (+ a front launch from boundary bytes); 2^8 <= n <= 2^18
int pippenger2_window(size_t n);
bool pippenger2_supported(size_t n);
size_t pippenger2_scratch_bytes(size_t n, int c);
void pippenger2(hipStream_t st, const AffDev *pts, const uint32_t *scalars, size_t n, int c, JacRaw *out, void *scratch, int *bad);
void pippenger2_boundary(hipStream_t st, const Words8 *points_xy, const Words8 *scalars, size_t n, int c, Words8 *out_xy,
                         AffDev *pts_tmp, void *scratch, int *bad);
// gather helper: dst[i] = src[idx(i)] for two-level strided sources (IPP round operands -> contiguous MSM inputs)
void gather_points(hipStream_t st, const AffDev *src, size_t src_outer, size_t cnt, size_t nb, AffDev *dst, size_t dst_outer);
void gather_scalars(hipStream_t st, const Words8 *src, size_t src_outer, size_t cnt, size_t nb, Words8 *dst, size_t dst_outer);

// out[b] = sum_{i<n} in[b*n + i]
void segmented_sum(hipStream_t st, const JacRaw *in, JacRaw *out, size_t nb, size_t n);

// ---- arkworks in-memory forms (k_ark.hip): Scalar = 4 x u64 limbs of x 2^256 mod n; StarkPoint = Jacobian (X : Y : Z), each
// coordinate 4 x u64 limbs of c 2^256 mod p, identity Z = 0
void scalars_from_ark(hipStream_t st, const Words8 *in, Words8 *out_plain, size_t n, int *bad);
void scalars_to_ark(hipStream_t st, const Words8 *in_plain, Words8 *out, size_t n, int *bad);
void points_from_ark(hipStream_t st, const Words8 *in /* 3 per point */, JacRaw *out, size_t n, int *bad);
void points_to_ark(hipStream_t st, const JacRaw *in, Words8 *out /* 3 per point */, size_t n);
void aff_to_jacraw(hipStream_t st, const AffDev *in, JacRaw *out, size_t n);

// ---- fixed-base tables -------------------------------------------------------------------------
// table[(g*W + w) * 2^(c-1) + (d-1)] = d * 2^(c*w) * P_g,  W = 252/c + 1
size_t fixed_table_entries(int c, size_t ngens);
void fixed_table_build(hipStream_t st, int c, const AffDev *gens, size_t ngens, AffDev *table,
                       JacRaw *scratch /* >= ngens*W + entries */);
// out[b] = sum_g scalars[b*sc_stride + g*8 ..] * P_g over the generators [B, Bb, G_0..G_{n-1}, H_0..H_{n-1}] of a
// table built for capacity cap >= n, via table lookups (plain canonical scalars, 2 + 2n per MSM).
// partials: scratch of nb * fixed_msm_chunks(c, n, nb) points (NULL: one block per MSM).
// out[i] = scalars[i] * (generator 0 of a c = 16 table), one lane per scalar
void fixed_single16(hipStream_t st, const AffDev *table, const uint32_t *scalars, JacRaw *out, size_t n);
// L / R MSMs of an IPP round over resident generators, compact scalars (k_ipp_gens_scalars)
size_t fixed_msm_ipp_chunks(int c, size_t n0, size_t nmsm);
void fixed_msm_ipp(hipStream_t st, int c, const AffDev *table, size_t n0, size_t cap, size_t cur, const uint32_t *scalars,
                   JacRaw *out, size_t nmsm, JacRaw *partials, bool sum_partials = true /* false: the caller sums the nmsm x chunks partials itself (k_ipp_round_tail) */);
// kinds > 0: the nb MSMs are large, many (>= 64 per class) and come in `kinds` interleaved classes (MSM i is of class i % kinds) whose
// scalars look alike within a class -- a batch's A_I, A_O, S rows: an MSM per lane, a run of generators per wave (k_fixed_msm_m)
size_t fixed_msm_chunks(int c, size_t n, size_t nb, int kinds = 0);
void fixed_msm(hipStream_t st, int c, const AffDev *table, size_t n, size_t cap, const uint32_t *scalars,
               size_t sc_stride_words, JacRaw *out, size_t nb, JacRaw *partials, int lpm = 0 /* lanes per small MSM: 32 = shorter lanes for a launch that is a link of a lone chain; 0 = by batch size */,
               int kinds = 0);

// ---- verification tail -------------------------------------------------------------------------
// per proof: sum of nvar variable-base results + the fixed-base partial; ok = is_identity;
// mega (optional) = boundary affine of the sum
// bad_sc / bad_pt (optional): per-proof malformed-input bits; a proof with one set gets ok = 0
void verify_finalize(hipStream_t st, const JacRaw *var, size_t nvar, const JacRaw *fixed, size_t nb,
                     int32_t *ok, Words8 *mega_xy, const int32_t *bad_sc = nullptr, const int32_t *bad_pt = nullptr);

// ---- scalar field (k_scalar.hip) ---------------------------------------------------------------
void scalars_check(hipStream_t st, const Words8 *in, size_t n, int *bad);   // canonical (< n)?
// the same with per-unit attribution: bad_unit[i / per_unit] = 1 for a non-canonical scalar i (bad_unit zeroed by the caller)
void scalars_check_proof(hipStream_t st, const Words8 *in, size_t n, size_t per_unit, int *bad, int32_t *bad_unit);
void batch_inverse(hipStream_t st, Words8 *io, size_t n, int *bad_zero);
void inner_product(hipStream_t st, const Words8 *a, const Words8 *b, size_t n, Words8 *out, void *scratch);
size_t inner_product_scratch_bytes(size_t n);
void fold_scalars(hipStream_t st, size_t n, const Words8 *u, const Words8 *u_inv, const Words8 *a,
                  const Words8 *b, Words8 *a_out, Words8 *b_out);
void verification_scalars(hipStream_t st, const Words8 *challenges, size_t k, size_t n, Words8 *u_sq,
                          Words8 *u_inv_sq, Words8 *s);
// batched IPP scalar kernels (proof-major arrays of nb x n)
// out[p][i] = x[p*x_outer + i*x_stride] * y[p*y_outer + i*y_stride]  for i < cnt (strides in Words8 units)
void sc_mul_strided(hipStream_t st, size_t nb, size_t cnt, const Words8 *x, size_t x_outer, size_t x_stride,
                    const Words8 *y, size_t y_outer, size_t y_stride, Words8 *out);
// out[p] = <x[p*x_outer .. +cnt), y[p*y_outer .. +cnt)>
void sc_dot_batched(hipStream_t st, size_t nb, size_t cnt, const Words8 *x, size_t x_outer, const Words8 *y,
                    size_t y_outer, Words8 *out, size_t out_stride);
// a'[p][i] = a[p][i] u_p + u_p^-1 a[p][h+i] ; b'[p][i] = b[p][i] u_p^-1 + u_p b[p][h+i]   (in: nb x 2h, out: nb x h)
// IPP over resident generators: L/R MSM scalars over the original generators, and the generator fold as a
// coefficient update (see k_scalar.hip)
void ipp_gens_scalars(hipStream_t st, size_t nb, size_t n0, size_t cur, const Words8 *a, const Words8 *b,
                      const Words8 *cG, const Words8 *cH, const Words8 *cLR, const Words8 *w, Words8 *msc, size_t slo = 0,
                      size_t shi = (size_t)-1, bool with_q = true);
void ipp_r1cs_factors(hipStream_t st, size_t nb, size_t np, size_t n1, const Words8 *u, const Words8 *y_inv, Words8 *cG,
                      Words8 *cH);
void ipp_gens_fold(hipStream_t st, size_t nb, size_t n0, size_t cur, const Words8 *u, const Words8 *u_inv, Words8 *cG,
                   Words8 *cH);
void fold_scalars_batched(hipStream_t st, size_t nb, size_t h, const Words8 *u, const Words8 *u_inv, const Words8 *a,
                          const Words8 *b, Words8 *a_out, Words8 *b_out);

// x[p][i] *= w[p]  (in place, nb x cnt) ; out[i] = sum_p w[p] * x[p][i]
void sc_scale_rows(hipStream_t st, size_t nb, size_t cnt, Words8 *x, const Words8 *w);
void sc_weighted_colsum(hipStream_t st, size_t nb, size_t cnt, const Words8 *x, const Words8 *w, Words8 *out);

// column-major constraint weights: for output o in [0, 3n + m + 1): terms col_ptr[o]..col_ptr[o+1]
// outputs ordered wL[0..n) wR[0..n) wO[0..n) wV[0..m) wc
struct CircuitDev {
  const uint32_t *col_ptr;   // 3n + m + 2
  const uint32_t *row;       // nnz: constraint row of the term
  const Words8 *coeff;       // nnz: Montgomery form
  size_t q, n, m, nnz;
  // Randomized (second-phase) constraints whose coefficients are affine in gadget challenges chi_1..chi_nchi (verifier.rs:366-385):
  // a term's row index r' = j * q + r selects the multiplier chi_j (chi_0 = 1) of z^(r+1); the z-power table of a proof then has
  // qz = (1 + nchi) * q entries, zpow[j * q + r] = chi_j * z^(r+1), and the flattening itself is unchanged.
  size_t nchi, qz;
};
// CSR constraint rows (device copies of the ABI arrays; coefficients plain canonical or, ark, ark-ff Montgomery limbs) -> the
// column-major form above: col_ptr (3n + m + 2), rows / coeff_out (nnz, Montgomery form); fill: 3n + m + 1 scratch counters;
// *bad |= 1 on an unknown variable kind, an index out of range or a non-canonical coefficient
void circuit_transpose(hipStream_t st, size_t q, const uint32_t *row_ptr, const uint32_t *kind, const uint32_t *idx, const Words8 *coeff_in,
                       size_t n_mul, size_t m, bool ark, uint32_t *col_ptr, uint32_t *fill, uint32_t *rows, Words8 *coeff_out, int *bad);
// zpow scratch: nb * q field elements (9 int32 each)
void flatten(hipStream_t st, const CircuitDev &c, size_t nb, const Words8 *z, size_t z_stride_words,
             Words8 *wL, Words8 *wR, Words8 *wO, Words8 *wV, Words8 *wc, int32_t *zpow_scratch, const Words8 *chi = nullptr);

// ---- R1CS prover polynomials (r1cs/prover.rs:587-619, 659-672) -----------------------------------
// polys: raw Montgomery limbs, layout [6][nb][n][9]: l1 l2 l3 r0 r1 r3
void prover_polys(hipStream_t st, const CircuitDev &c, size_t nb, const Words8 *y, const Words8 *y_inv,
                  const Words8 *a_L, const Words8 *a_R, const Words8 *a_O, const Words8 *s_L, const Words8 *s_R,
                  const int32_t *zpow, int32_t *polys, Words8 *wV_out);
// t[nb][6] = t1..t6 (util.rs:152-170 special_inner_product)
void prover_tcoeffs(hipStream_t st, size_t nb, size_t n, const int32_t *polys, Words8 *t_out);
// l_vec, r_vec[nb][padded_n] = l(x), r(x) with the padding of prover.rs:661-672
void prover_eval(hipStream_t st, size_t nb, size_t n, size_t padded_n, const Words8 *x, const Words8 *y,
                 const int32_t *polys, Words8 *l_vec, Words8 *r_vec);
// zpow[b][j * q + r] = chi_{b,j} * z_b^(r+1), j <= nchi (chi: nb x nchi plain words, may be nullptr when nchi = 0)
void zpow_table(hipStream_t st, size_t nb, size_t q, const Words8 *z, size_t z_stride_words, int32_t *zpow, size_t nchi = 0,
                const Words8 *chi = nullptr);

// ---- device-side verifier transcript (k_transcript.hip) ------------------------------
struct TrStep { uint8_t kind, label, validate, pad; uint32_t src; uint64_t value; };
size_t transcript_schedule_max(size_t m, size_t k);
int transcript_schedule(TrStep *out, size_t m, size_t k, size_t padded_n, size_t nchi = 0);   // host: fills the step list, returns its length
                                                                                                // (nchi > 0: two-phase circuit with nchi gadget challenges)
// gadget_label (32 bytes, zero-padded) + chi_out (nb x nchi): the challenge of a second-phase gadget (schedule built with nchi = 1)
void verify_transcript(hipStream_t st, size_t nb, size_t m, size_t k, const TrStep *steps_dev, int nsteps, const Words8 *init_state,
                       const Words8 *points, const Words8 *scalars, Words8 *challenges, int32_t *tr_bad,
                       const uint8_t *gadget_label = nullptr, Words8 *chi_out = nullptr, size_t nchi = 0);
void and_not(hipStream_t st, int32_t *ok, const int32_t *bad, size_t n);
void or_flag(hipStream_t st, const int32_t *bad, size_t n, int *flag);
void zero_flag(hipStream_t st, const Words8 *s, size_t n, int *flag);

// Verifier scalar assembly (r1cs/verifier.rs:457-532).  Writes
//   fixed_sc[nb][2 + 2*padded_n] (B, B_blinding, g, h) and var_sc[nb][11 + m + 2k]
//   (A_I1 A_O1 S1 A_I2 A_O2 S2 V.. T_1 T_3 T_4 T_5 T_6 L.. R..), plain canonical words;
//   full_sc (optional): nb x (13 + m + 2 padded_n + 2k) in verifier.rs:517-532 order.
//   bad (optional): set to 1 when a challenge or proof scalar is not canonical (< n)
//   zpow_scratch: verify_scalars_scratch_ints(c, d) int32 (z powers + the large-proof path's partials)
size_t verify_scalars_scratch_ints(const CircuitDev &c, const VerifyDims &d);
//   bad_proof (optional): nb int32, [p] = 1 when one of proof p's challenges / scalars is not canonical, else 0
//   prep_done: the inversion pass (vs_prep.cuh) has already run into the aux area of zpow_scratch (verify_scalars_aux)
void verify_scalars(hipStream_t st, const CircuitDev &c, const VerifyDims &d, const Words8 *challenges,
                    const Words8 *proof_scalars, Words8 *fixed_sc, Words8 *var_sc, Words8 *full_sc,
                    int32_t *zpow_scratch, int *bad, int32_t *bad_proof = nullptr, bool prep_done = false, bool prep_fast = false);
// wave-sized proofs (padded n = 64): the inversion pass can also do the serial part of the assembly when it is given the proof
// scalars and the output arrays (vs_prep.cuh); prep_fast tells verify_scalars that the caller's fused launch has done so
bool verify_scalars_fast_shape(const CircuitDev &c, const VerifyDims &d);
// where the inversion pass writes (inside zpow_scratch) and its per-proof stride in field elements; false = the large-proof
// path, which runs its own inversion pass
bool verify_scalars_aux(const CircuitDev &c, const VerifyDims &d, int32_t *zpow_scratch, int32_t **aux, size_t *aux_stride);

// ---- combined batch check in eight launches (k_pip2.hip): sum_p rho_p * mega_check_p as ONE boundary point
// optional stage markers for per-kernel event timing: fn(ctx, kind, stream) is called before and after a stage
typedef void (*ProfMarkFn)(void *ctx, int kind, hipStream_t st);
struct ProfMark {
  ProfMarkFn fn; void *ctx; int kind; hipStream_t st;
  ProfMark(ProfMarkFn f, void *c, int k, hipStream_t s) : fn(f), ctx(c), kind(k), st(s) { if (fn) fn(ctx, kind, st); }
  ~ProfMark() { if (fn) fn(ctx, kind, st); }
};
struct CombinedArgs {
  CircuitDev circ; VerifyDims d; size_t nvar;
  const Words8 *points, *proof_scalars, *challenges, *rho;     // ABI bytes in HBM
  Words8 *fixed_sc, *var_sc;                                   // nb x (2 + 2 padded_n), nb x nvar (written here)
  int32_t *zpow_scratch;                                       // verify_scalars_scratch_ints
  const AffDev *table; size_t cap; int c;                      // resident generator tables
  void *scratch;                                               // verify_combined2_scratch_bytes
  int *bad; Words8 *partial_xy;                                // out: 64 boundary bytes
  ProfMarkFn prof; void *prof_ctx;
};
size_t verify_combined2_scratch_bytes(size_t nb, size_t nvar, size_t nfix);
bool verify_combined2_supported(size_t nb, size_t nvar, int c, size_t padded_n);
void verify_combined2(hipStream_t st, const CombinedArgs &a);

// the prover's blinding vectors drawn on the device from per-prover keys (k_transcript.hip, "BlindVec v1"):
// sL / sR [p * stride + off + i] for i < cnt, plain canonical words
void blind_vectors(hipStream_t st, const Words8 *keys, size_t nb, size_t cnt, Words8 *sL, Words8 *sR, size_t stride, size_t off);
// MSM scalar rows of the three commitments A_I, A_O, S of nb provers over [B, B_blinding, G_0.., H_0..] (prover.rs:465-494 /
// :532-565): rows[(3 p + w) * (2 + 2 n)] for w = 0, 1, 2 from the witness planes (nb x stride, plain canonical; multipliers
// [lo, n) are live, the rest of a row is zero) and blinds (nb x 3: i, o, s blinding)
// [slo, shi): the generator indices of THIS rank's share when one large proof is split over the GPUs of a node (the other
// entries of a row are zero; with_blind: the B_blinding term belongs to one rank only); default = everything
void commit_rows(hipStream_t st, size_t nb, size_t n, size_t lo, size_t stride, const Words8 *aL, const Words8 *aR, const Words8 *aO,
                 const Words8 *sL, const Words8 *sR, const Words8 *blinds, Words8 *rows, size_t slo = 0, size_t shi = (size_t)-1,
                 bool with_blind = true);
void shard_mask(hipStream_t st, Words8 *fixed, size_t np, size_t slo, size_t shi, bool keep_pedersen, Words8 *var, size_t nvar, size_t vlo,
                size_t vhi);
// one IPP prover round of the device transcript: append L, R; u = challenge (k_transcript.hip)
void ipp_round_challenge(hipStream_t st, size_t nb, uint64_t *states, const Words8 *lr_xy, Words8 *u_out);
// the whole tail of a round in one launch: L, R (Jacobian sums, nb x 2) -> boundary bytes, the three transcript steps on a
// wave-cooperative Keccak, u and u^-1
void ipp_round_tail(hipStream_t st, size_t nb, const JacRaw *sums, uint64_t *states, Words8 *lr_xy, Words8 *u_out, Words8 *uinv_out, const JacRaw *partials = nullptr, size_t chunks = 0 /* > 1: sums[2p + s] = sum of partials[(2p + s) * chunks + i], added up here */);

// ---- wire codec of points (k_codec.hip): 32-byte compressed <-> 64-byte affine boundary form ------
size_t sqrt_table_bytes();
void sqrt_tables_build(hipStream_t st, void *tab /* sqrt_table_bytes() */);
void points_decompress(hipStream_t st, const Words8 *in, Words8 *out_xy, int32_t *ok, size_t n, const void *tab);
void points_compress(hipStream_t st, const Words8 *xy, Words8 *out, size_t n);
// wire-format proofs -> operands of the verification pipeline (k_codec.hip)
void wire_unpack(hipStream_t st, const uint8_t *proofs, size_t proof_len, const uint8_t *commitments, size_t nb, size_t m,
                 size_t k, int two_phase, Words8 *comp, Words8 *scalars, int32_t *fmt_ok);
void wire_and_ok(hipStream_t st, int32_t *ok, const int32_t *fmt_ok, const int32_t *dec_ok, size_t nb, size_t nvar);

}  // namespace bpk
