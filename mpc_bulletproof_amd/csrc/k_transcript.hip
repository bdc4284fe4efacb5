// k_transcript.hip -- SURVEY.md 8f row N1: the verifier's Fiat-Shamir transcript on the device
// (Keccak-f[1600] / keccak256 with the original 0x01 padding, little-endian scalar and affine-xy point
// absorption, hash_to_scalar), so that a whole batch of Verifier::verify calls -- reference
// src/r1cs/verifier.rs:393-554 including the transcript replay :398-455,506 and
// src/inner_product_proof.rs:269-278 -- runs without per-proof host hashing (at 10^6 verifications/s the
// ~80 keccak permutations per proof would need tens of host cores).
//
// Scope: circuits without randomized (second-phase) constraints, and two-phase circuits whose gadgets draw their challenges
// right after the phase separator (RandomizableConstraintSystem::specify_randomized_constraints with
// cs.challenge_scalar(label): the shuffle gadget, tests/r1cs.rs:23-62) -- the schedule is still fixed; the gadget challenges are
// exported (chi) and the constraint weights depend on them through the parametric circuit form (kernels.h CircuitDev).
// The hash chain is the build's stand-in for merlin's HashChainTranscript (source absent from the
// reference tree: transcript bytes are parity-unpinned); it is bit-identical to the host
// transcripts of mpc_bulletproof_amd/host and of the oracle:
//   append_message : state = keccak256(state || 0x00 || pad_label(l) || u32le(len) || msg)
//   challenge_bytes: state = keccak256(state || 0x01 || pad_label(l)),  output = state
// One lane per proof; state and message blocks live in registers (all layouts are compile-time).
#include "fe29.cuh"
#include "ec_dev.cuh"
#include "fn_dev.cuh"
#include "kernels.h"

using namespace bp;

namespace bpk {

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

__device__ __forceinline__ void keccak_f(uint64_t s[25]) {
  constexpr uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 1
  for (int r = 0; r < 24; r++) {
    uint64_t c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20], c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];
    uint64_t c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22], c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];
    uint64_t c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
    uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
#pragma unroll
    for (int y = 0; y < 25; y += 5) { s[y] ^= d0; s[y + 1] ^= d1; s[y + 2] ^= d2; s[y + 3] ^= d3; s[y + 4] ^= d4; }
    // rho + pi
    uint64_t b[25];
    b[0] = s[0];            b[10] = rotl64(s[1], 1);   b[20] = rotl64(s[2], 62);  b[5] = rotl64(s[3], 28);   b[15] = rotl64(s[4], 27);
    b[16] = rotl64(s[5], 36); b[1] = rotl64(s[6], 44);  b[11] = rotl64(s[7], 6);   b[21] = rotl64(s[8], 55);  b[6] = rotl64(s[9], 20);
    b[7] = rotl64(s[10], 3);  b[17] = rotl64(s[11], 10); b[2] = rotl64(s[12], 43);  b[12] = rotl64(s[13], 25); b[22] = rotl64(s[14], 39);
    b[23] = rotl64(s[15], 41); b[8] = rotl64(s[16], 45); b[18] = rotl64(s[17], 15); b[3] = rotl64(s[18], 21);  b[13] = rotl64(s[19], 8);
    b[14] = rotl64(s[20], 18); b[24] = rotl64(s[21], 2); b[9] = rotl64(s[22], 61);  b[19] = rotl64(s[23], 56); b[4] = rotl64(s[24], 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
      s[y] = b[y] ^ (~b[y + 1] & b[y + 2]);
      s[y + 1] = b[y + 1] ^ (~b[y + 2] & b[y + 3]);
      s[y + 2] = b[y + 2] ^ (~b[y + 3] & b[y + 4]);
      s[y + 3] = b[y + 3] ^ (~b[y + 4] & b[y]);
      s[y + 4] = b[y + 4] ^ (~b[y] & b[y + 1]);
    }
    s[0] ^= RC[r];
  }
}

// pad_label: label right-padded with zeros to 32 bytes, as 4 little-endian words
struct Label { uint64_t w[4]; };
constexpr Label mk_label(const char *s) {
  Label l{{0, 0, 0, 0}};
  for (int i = 0; s[i] && i < 32; i++) l.w[i >> 3] |= (uint64_t)(uint8_t)s[i] << (8 * (i & 7));
  return l;
}

// state = keccak256(state[32] || flag || label[32] [|| u32le(tail[0]) || tail[1 .. NT)]): NT = 0 is challenge_bytes (65 bytes),
// NT > 0 append_message with tail[0] = the message length and NT - 1 message words, which start at byte 69.  The byte shifts
// (one after the flag, five after the length) are resolved at compile time.  A 64-byte point makes 133 bytes: ONE rate block.
template <int NT>
__device__ __forceinline__ void chain_hash(uint64_t state[4], uint8_t flag, const Label &lab, const uint64_t *tail) {
  constexpr int NM = NT > 0 ? NT - 1 : 0;                 // message words
  constexpr int TOTAL = NT > 0 ? 69 + 8 * NM : 65;        // message bytes
  constexpr int NW = 9 + NM;                               // words that hold them (the last one partially)
  uint64_t w[NW];
#pragma unroll
  for (int i = 0; i < 4; i++) w[i] = state[i];
  w[4] = (uint64_t)flag | (lab.w[0] << 8);
#pragma unroll
  for (int i = 1; i < 4; i++) w[4 + i] = (lab.w[i - 1] >> 56) | (lab.w[i] << 8);
  w[8] = lab.w[3] >> 56;
  if (NT > 0) {
    w[8] |= (tail[0] & 0xFFFFFFFFull) << 8;
    if (NM > 0) w[8] |= tail[1] << 40;
#pragma unroll
    for (int i = 0; i < NM; i++) w[9 + i] = (tail[1 + i] >> 24) | (i + 1 < NM ? tail[2 + i] << 40 : 0);
  }
  // pad: 0x01 at byte TOTAL, 0x80 at the last byte of the final block
  uint64_t st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = 0;
  constexpr int RATE_W = 17;
  constexpr int NBLK = TOTAL / 136 + 1;
#pragma unroll
  for (int blk = 0; blk < NBLK; blk++) {
#pragma unroll
    for (int i = 0; i < RATE_W; i++) {
      const int wi = blk * RATE_W + i;
      uint64_t v = wi < NW ? w[wi] : 0;
      if (wi == TOTAL / 8) v ^= (uint64_t)0x01 << (8 * (TOTAL % 8));
      if (blk == NBLK - 1 && i == RATE_W - 1) v ^= 0x8000000000000000ULL;
      st[i] ^= v;
    }
    keccak_f(st);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) state[i] = st[i];
}
// keccak256 of exactly 32 bytes (hash_to_scalar's second half, util.rs:254-255)
__device__ __forceinline__ void keccak256_32(const uint64_t in[4], uint64_t out[4]) {
  uint64_t st[25];
#pragma unroll
  for (int i = 0; i < 25; i++) st[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) st[i] = in[i];
  st[4] ^= 0x01;
  st[16] ^= 0x8000000000000000ULL;
  keccak_f(st);
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = st[i];
}

// int_LE(lo || hi) mod n as plain canonical words (Scalar::from_le_bytes_mod_order of 64 bytes)
__device__ __forceinline__ void wide_to_scalar(const uint64_t st[4], const uint64_t hi[4], Words8 *out) {
  uint32_t lw[8], hw[8];
#pragma unroll
  for (int i = 0; i < 4; i++) { lw[2 * i] = (uint32_t)st[i]; lw[2 * i + 1] = (uint32_t)(st[i] >> 32); hw[2 * i] = (uint32_t)hi[i]; hw[2 * i + 1] = (uint32_t)(hi[i] >> 32); }
  // lo + hi * 2^256 (mod n) through the lazy Montgomery arithmetic (256-bit inputs are within its value bound)
  Fn w256 = fe_zero<FN>();
  w256.v[8] = 1 << (256 - 232);              // plain 2^256 (limb 8 has weight 2^232)
  Fn lo = mul(unpack<FN>(lw), fe_r2<FN>());
  Fn hv = mul(mul(unpack<FN>(hw), fe_r2<FN>()), mul(w256, fe_r2<FN>()));   // (hi R)(2^256 R)/R = hi 2^256 R
  Fn one = fe_zero<FN>();
  one.v[0] = 1;
  uint32_t ow[8];
  pack(ow, canon(mul(add(lo, hv), one)));
#pragma unroll
  for (int i = 0; i < 8; i++) out->w[i] = ow[i];
}
// challenge_scalar: challenge_bytes then hash_to_scalar = int_LE(low || keccak256(low)) mod n  (util.rs:252-267)
__device__ __forceinline__ void tr_challenge_scalar(uint64_t st[4], const Label &l, Words8 *out) {
  chain_hash<0>(st, 0x01, l, nullptr);
  uint64_t hi[4];
  keccak256_32(st, hi);
  wide_to_scalar(st, hi, out);
}

__device__ __forceinline__ bool load64(const Words8 *p, uint64_t m[8]) {   // a point (two Words8); returns "is identity"
  uint64_t o = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    m[i] = (uint64_t)p[0].w[2 * i] | ((uint64_t)p[0].w[2 * i + 1] << 32);
    m[4 + i] = (uint64_t)p[1].w[2 * i] | ((uint64_t)p[1].w[2 * i + 1] << 32);
    o |= m[i] | m[4 + i];
  }
  return o == 0;
}

// The verifier's transcript schedule as data (built on the host by transcript_schedule): one switch with a
// single inlined copy of each message shape keeps the kernel at ~6 keccak-f bodies instead of ~40.
enum : uint8_t { TS_DOMSEP = 0, TS_U64 = 1, TS_POINT = 2, TS_SCALAR = 3, TS_CHALLENGE = 4, TS_GADGET_CHALLENGE = 5 };
enum : uint8_t { LB_V, LB_m, LB_AI1, LB_AO1, LB_S1, LB_AI2, LB_AO2, LB_S2, LB_y, LB_z, LB_T1, LB_T3, LB_T4, LB_T5, LB_T6, LB_u, LB_x,
                 LB_tx, LB_txb, LB_eb, LB_w, LB_r, LB_n, LB_L, LB_R, LB_r1cs, LB_1phase, LB_ipp, LB_2phase, LB_COUNT };
__constant__ Label TR_LABELS[LB_COUNT] = {
    mk_label("V"), mk_label("m"), mk_label("A_I1"), mk_label("A_O1"), mk_label("S1"), mk_label("A_I2"), mk_label("A_O2"), mk_label("S2"),
    mk_label("y"), mk_label("z"), mk_label("T_1"), mk_label("T_3"), mk_label("T_4"), mk_label("T_5"), mk_label("T_6"), mk_label("u"),
    mk_label("x"), mk_label("t_x"), mk_label("t_x_blinding"), mk_label("e_blinding"), mk_label("w"), mk_label("r"), mk_label("n"),
    mk_label("L"), mk_label("R"), mk_label("r1cs v1"), mk_label("r1cs-1phase"), mk_label("ipp v1"), mk_label("r1cs-2phase")};

// One lane per proof.  points layout as bpgpu_r1cs_verify_batch: A_I1 A_O1 S1 A_I2 A_O2 S2 | V[m] | T_1 T_3 T_4 T_5 T_6 | L[k] | R[k]
// scalars: t_x t_x_blinding e_blinding a b.  init_state: the 32-byte chain state the host holds when it would
// call Verifier::new (after Transcript::new(label) and any application preamble).
// challenges out: y z u x w r u_1..u_k ; tr_bad[p] = 1 if a validated point is the identity
// (TranscriptProtocol::validate_and_append_point -> VerificationError, transcript.rs:101-113).
__global__ void __launch_bounds__(64) k_verify_transcript(size_t nb, size_t nvar, size_t nch, const TrStep *steps, int nsteps,
                                                          const Words8 *init_state, const Words8 *points, const Words8 *scalars,
                                                          Words8 *challenges, int32_t *tr_bad, Label gadget_label, Words8 *chi,
                                                          size_t nchi) {
  // (no raised wave priority here, unlike the other short links of a batch's chain: a lone wave of dependent 64-bit operations
  // issues back to back, so at priority 3 the 16 transcript waves of each of ~20 batches in flight kept their SIMDs to
  // themselves for 0.7 ms at a time -- measured 3.54 -> 3.95 M verifications/s for the device-transcript leg without it)
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nb) return;
  const Words8 *pt = points + p * nvar * 2;
  const Words8 *sc = scalars + p * 5;
  Words8 *ch = challenges + p * nch;
  uint64_t st[4];
#pragma unroll
  for (int i = 0; i < 4; i++) st[i] = (uint64_t)init_state[p].w[2 * i] | ((uint64_t)init_state[p].w[2 * i + 1] << 32);
  bool bad = false;
#pragma unroll 1
  for (int t = 0; t < nsteps; t++) {
    const TrStep s = steps[t];
    const Label lab = TR_LABELS[s.label];
    switch (s.kind) {
      case TS_DOMSEP: {   // append_message("dom-sep", pad_label(what)): s.label is `what`
        constexpr Label DS = mk_label("dom-sep");
        uint64_t tail[5] = {32, lab.w[0], lab.w[1], lab.w[2], lab.w[3]};
        chain_hash<5>(st, 0x00, DS, tail);
        break;
      }
      case TS_SCALAR: {
        uint64_t tail[5];
        tail[0] = 32;
#pragma unroll
        for (int i = 0; i < 4; i++) tail[1 + i] = (uint64_t)sc[s.src].w[2 * i] | ((uint64_t)sc[s.src].w[2 * i + 1] << 32);
        chain_hash<5>(st, 0x00, lab, tail);
        break;
      }
      case TS_U64: {
        uint64_t tail[2] = {8, s.value};
        chain_hash<2>(st, 0x00, lab, tail);
        break;
      }
      case TS_POINT: {
        uint64_t tail[9];
        tail[0] = 64;
        bool inf = load64(pt + 2 * (size_t)s.src, tail + 1);
        if (s.validate && inf) bad = true;
        chain_hash<9>(st, 0x00, lab, tail);
        break;
      }
      case TS_GADGET_CHALLENGE:   // cs.challenge_scalar(gadget label) inside the randomized-constraints callback (verifier.rs:366-385)
        tr_challenge_scalar(st, gadget_label, &chi[p * nchi + s.src]);
        break;
      default:
        tr_challenge_scalar(st, lab, &ch[s.src]);
        break;
    }
  }
  tr_bad[p] = bad ? 1 : 0;
}
// Verifier::verify's transcript order for a circuit without randomized constraints
int transcript_schedule(TrStep *out, size_t m, size_t k, size_t padded_n, size_t nchi) {
  int n = 0;
  auto add = [&](uint8_t kind, uint8_t label, uint32_t src, uint8_t validate, uint64_t value) {
    out[n].kind = kind; out[n].label = label; out[n].validate = validate; out[n].src = src; out[n].value = value; n++;
  };
  add(TS_DOMSEP, LB_r1cs, 0, 0, 0);                                              // Verifier::new, verifier.rs:271
  for (size_t j = 0; j < m; j++) add(TS_POINT, LB_V, (uint32_t)(6 + j), 0, 0);    // Verifier::commit, :303
  add(TS_U64, LB_m, 0, 0, m);                                                     // :398
  add(TS_POINT, LB_AI1, 0, 1, 0); add(TS_POINT, LB_AO1, 1, 1, 0); add(TS_POINT, LB_S1, 2, 1, 0);   // :401-406
  if (nchi == 0) add(TS_DOMSEP, LB_1phase, 0, 0, 0);                              // :371
  else {                                                                          // :373-383: phase separator, then the gadgets' challenges
    add(TS_DOMSEP, LB_2phase, 0, 0, 0);
    for (size_t j = 0; j < nchi; j++) add(TS_GADGET_CHALLENGE, LB_z, (uint32_t)j, 0, 0);
  }
  add(TS_POINT, LB_AI2, 3, 0, 0); add(TS_POINT, LB_AO2, 4, 0, 0); add(TS_POINT, LB_S2, 5, 0, 0);   // :428-430
  add(TS_CHALLENGE, LB_y, 0, 0, 0); add(TS_CHALLENGE, LB_z, 1, 0, 0);             // :432-433
  const uint8_t tl[5] = {LB_T1, LB_T3, LB_T4, LB_T5, LB_T6};
  for (int j = 0; j < 5; j++) add(TS_POINT, tl[j], (uint32_t)(6 + m + j), 1, 0);  // :435-444
  add(TS_CHALLENGE, LB_u, 2, 0, 0); add(TS_CHALLENGE, LB_x, 3, 0, 0);             // :446-447
  add(TS_SCALAR, LB_tx, 0, 0, 0); add(TS_SCALAR, LB_txb, 1, 0, 0); add(TS_SCALAR, LB_eb, 2, 0, 0);   // :449-453
  add(TS_CHALLENGE, LB_w, 4, 0, 0);                                               // :455
  add(TS_DOMSEP, LB_ipp, 0, 0, 0); add(TS_U64, LB_n, 0, 0, padded_n);             // inner_product_proof.rs:269
  for (size_t j = 0; j < k; j++) {                                                // inner_product_proof.rs:274-278
    add(TS_POINT, LB_L, (uint32_t)(11 + m + j), 1, 0);
    add(TS_POINT, LB_R, (uint32_t)(11 + m + k + j), 1, 0);
    add(TS_CHALLENGE, LB_u, (uint32_t)(6 + j), 0, 0);
  }
  add(TS_CHALLENGE, LB_r, 5, 0, 0);                                               // verifier.rs:506
  return n;
}
size_t transcript_schedule_max(size_t m, size_t k) { return 48 + m + 3 * k; }
void verify_transcript(hipStream_t st, size_t nb, size_t m, size_t k, const TrStep *steps_dev, int nsteps, const Words8 *init_state,
                       const Words8 *points, const Words8 *scalars, Words8 *challenges, int32_t *tr_bad,
                       const uint8_t *gadget_label, Words8 *chi_out, size_t nchi) {
  if (!nb) return;
  Label gl{{0, 0, 0, 0}};
  if (gadget_label) for (int i = 0; i < 32; i++) gl.w[i >> 3] |= (uint64_t)gadget_label[i] << (8 * (i & 7));
  hipLaunchKernelGGL(k_verify_transcript, dim3((nb + 63) / 64), dim3(64), 0, st, nb, 11 + m + 2 * k, 6 + k, steps_dev, nsteps,
                     init_state, points, scalars, challenges, tr_bad, gl, chi_out, nchi);
}
// ok[p] &= !tr_bad[p]
__global__ void k_and_not(int32_t *ok, const int32_t *bad, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && bad[i]) ok[i] = 0;
}
// flag |= 2 if any bad[i] (the screened verification: a batch with a failed transcript replay takes the per-proof path)
__global__ void k_or_flag(const int32_t *bad, size_t n, int *flag) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && bad[i]) atomicOr(flag, 2);
}
void or_flag(hipStream_t st, const int32_t *bad, size_t n, int *flag) {
  if (n) hipLaunchKernelGGL(k_or_flag, dim3((n + 255) / 256), dim3(256), 0, st, bad, n, flag);
}
// flag |= 4 if any of the n scalars is zero (a zero weight would drop its proof from a combined check: the screened verification
// then does not trust the check and verifies the batch proof by proof)
__global__ void k_zero_flag(const Words8 *s, size_t n, int *flag) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t o = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) o |= s[i].w[j];
  if (o == 0) atomicOr(flag, 4);
}
void zero_flag(hipStream_t st, const Words8 *s, size_t n, int *flag) {
  if (n) hipLaunchKernelGGL(k_zero_flag, dim3((n + 255) / 256), dim3(256), 0, st, s, n, flag);
}
void and_not(hipStream_t st, int32_t *ok, const int32_t *bad, size_t n) {
  if (n) hipLaunchKernelGGL(k_and_not, dim3((n + 255) / 256), dim3(256), 0, st, ok, bad, n);
}

// One IPP prover round of the transcript (inner_product_proof.rs:119-123 / :177-181) for nb provers, one lane each:
// append_point("L", L_p), append_point("R", R_p), u_p = challenge_scalar("u").  states: 4 x u64 per prover, updated.
__global__ void __launch_bounds__(64) k_ipp_round_challenge(size_t nb, uint64_t *states, const Words8 *lr, Words8 *u_out) {
  __builtin_amdgcn_s_setprio(3);
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nb) return;
  uint64_t st[4];
#pragma unroll
  for (int i = 0; i < 4; i++) st[i] = states[p * 4 + i];
  uint64_t tail[9];
  tail[0] = 64;
  load64(lr + p * 4, tail + 1);
  chain_hash<9>(st, 0x00, TR_LABELS[LB_L], tail);
  load64(lr + p * 4 + 2, tail + 1);
  chain_hash<9>(st, 0x00, TR_LABELS[LB_R], tail);
  tr_challenge_scalar(st, TR_LABELS[LB_u], &u_out[p]);
#pragma unroll
  for (int i = 0; i < 4; i++) states[p * 4 + i] = st[i];
}
// The prover's blinding vectors s_L, s_R (r1cs/prover.rs:457-462, 519-527: n random scalars each) drawn ON THE DEVICE from a
// 32-byte key per prover and phase that the host takes from its RNG at the point where the reference draws the vectors: "BlindVec v1",
//   block(key, v, j) = the first 128 bytes of Keccak-f[1600] over the 48-byte message key || u64le(v) || u64le(j), padded as keccak256
//                      pads (0x01 after the message, 0x80 at byte 135): a keyed Keccak in counter mode, 512-bit capacity
//   s_v[i]           = int_LE(block(key, v, i / 2)[64 (i mod 2) .. +64]) mod n        (v = 0: s_L, v = 1: s_R)
// 512 bits per scalar: the reduction mod n is unbiased to 2^-260 (four words per scalar, as the host RNG draws them, leave a
// 3 % bump on the low part of the range).  One lane per (prover, vector, block): 2 x 256 x 512 independent permutations for 256
// provers of 1024 multipliers instead of 525 000 wide reductions and 34 MB of staging on the host.  The CPU oracle restates the
// same stream (blind_vector in its transcript file), so proofs made this way are replayable under the test RNG.
__global__ void __launch_bounds__(64) k_blind_vectors(const Words8 *keys, size_t nb, size_t cnt, Words8 *sL, Words8 *sR, size_t stride,
                                                      size_t off) {
  const size_t blocks = (cnt + 1) / 2;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * 2 * blocks) return;
  const size_t j = t % blocks, v = (t / blocks) & 1, p = t / (2 * blocks);
  uint64_t s[25];
#pragma unroll
  for (int i = 0; i < 25; i++) s[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) s[i] = (uint64_t)keys[p].w[2 * i] | ((uint64_t)keys[p].w[2 * i + 1] << 32);
  s[4] = v;
  s[5] = j;
  s[6] = 0x01;
  s[16] = 0x8000000000000000ULL;
  keccak_f(s);
  Words8 *dst = (v ? sR : sL) + p * stride + off + 2 * j;
  wide_to_scalar(s, s + 4, dst);
  if (2 * j + 1 < cnt) wide_to_scalar(s + 8, s + 12, dst + 1);
}
void blind_vectors(hipStream_t st, const Words8 *keys, size_t nb, size_t cnt, Words8 *sL, Words8 *sR, size_t stride, size_t off) {
  if (!nb || !cnt) return;
  const size_t lanes = nb * 2 * ((cnt + 1) / 2);
  hipLaunchKernelGGL(k_blind_vectors, dim3((lanes + 63) / 64), dim3(64), 0, st, keys, nb, cnt, sL, sR, stride, off);
}
// ---- Keccak-f[1600] spread over 25 lanes of a wave: lane i = x + 5 y holds state word i (two states per wave: lanes 0..24 and
// 32..56).  One lane runs a permutation as ~7 200 dependent instructions (13 us at the rate a lone wave issues them); here a
// round is 18 ds_bpermute moves (the five column words for theta, the two neighbour columns' parities, the rho-pi source word,
// the two chi neighbours; 64-bit values = two moves each) and ~50 instructions: ~2 000 per permutation, 3.6x shorter -- for 8.5x
// the issue slots per state (a wave carries 2 states instead of 64).  That trade is right where a hash chain IS the critical path
// of a small batch (the prover's per-round challenge: 256 states, an otherwise idle chip) and wrong for the verifier's 1024 x 80
// permutations per batch, which stay one lane per proof.  i < 25; `base` = first lane of this state's 32-lane half.
struct CoopK { int i, src_pi, c1, c2, c3, c4, dm, dp, x1, x2; int rho; };
__device__ __forceinline__ CoopK coop_keccak_setup(int i, int base) {
  constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};   // by index x + 5 y
  CoopK k;
  const int ii = i < 25 ? i : 0, x = ii % 5, y = ii / 5;
  k.i = ii;
  k.rho = RHO[ii];
  // pi: B[y'][(2 x' + 3 y') mod 5] = rot(A[x'][y']), i.e. the word at (X, Y) comes from x' = 3 (Y - 3 X) mod 5 ... in index form:
  // destination (X, Y) <- source (x', y') with X = y', Y = (2 x' + 3 y') mod 5  =>  y' = X, x' = 3 (Y - 3 X) mod 5
  const int xs = (3 * (((y - 3 * x) % 5 + 5) % 5)) % 5, ys = x;
  k.src_pi = base + xs + 5 * ys;
  k.c1 = base + (ii + 5) % 25; k.c2 = base + (ii + 10) % 25; k.c3 = base + (ii + 15) % 25; k.c4 = base + (ii + 20) % 25;
  k.dm = base + (x + 4) % 5; k.dp = base + (x + 1) % 5;                      // any lane of the neighbour columns holds that column's parity
  k.x1 = base + 5 * y + (x + 1) % 5; k.x2 = base + 5 * y + (x + 2) % 5;
  return k;
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64);
  return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t coop_keccak_f(uint64_t a, const CoopK &k) {
  constexpr uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 1
  for (int r = 0; r < 24; r++) {
    const uint64_t c = a ^ shfl64(a, k.c1) ^ shfl64(a, k.c2) ^ shfl64(a, k.c3) ^ shfl64(a, k.c4);      // theta: column parity (in every lane of the column)
    const uint64_t cp = shfl64(c, k.dp);
    a ^= shfl64(c, k.dm) ^ ((cp << 1) | (cp >> 63));
    const uint64_t t = k.rho ? ((a << k.rho) | (a >> (64 - k.rho))) : a;                                 // rho (this lane's own offset) ...
    const uint64_t b = shfl64(t, k.src_pi);                                                              // ... pi
    a = b ^ (~shfl64(b, k.x1) & shfl64(b, k.x2));                                                        // chi
    if (k.i == 0) a ^= RC[r];                                                                            // iota
  }
  return a;
}
// The tail of one IPP prover round for nb provers (inner_product_proof.rs:119-123 / :177-181), one 32-lane half-wave per prover, in ONE
// launch instead of three (point conversion | hash chain | batch inversion) with two waits between them:
//   lanes 0, 1: L, R (Jacobian partial sums) -> affine boundary bytes, an inversion each, side by side;
//   lanes 0..24: transcript.append_point("L"), ("R"), challenge_scalar("u") -- six Keccak permutations, cooperatively (above);
//   lane 0: u = hash_to_scalar, u^-1.
// sums: nb x 2 points; lr_xy out: nb x 2 x 64 B; states: 4 x u64 per prover, updated; u_out, uinv_out: nb x 32 B.
__global__ void __launch_bounds__(64) k_ipp_round_tail(size_t nb, const JacRaw *sums, uint64_t *states, Words8 *lr_xy, Words8 *u_out,
                                                       Words8 *uinv_out, const JacRaw *partials, size_t chunks) {
  __shared__ uint64_t sh[2][24];          // per half: [0..3] chain state, [4..11] L x || y, [12..19] R x || y, [20..23] scratch (challenge low)
  __builtin_amdgcn_s_setprio(3);
  const int half = threadIdx.x >> 5, l = threadIdx.x & 31, base = half * 32;
  const size_t p = (size_t)blockIdx.x * 2 + half;
  const bool live = p < nb;
  uint64_t *S = sh[half];
  // chunks > 1: the round's MSMs left `chunks` partial sums per point (k_fixed_msm_ipp_g): lanes 0..15 add up L's, lanes 16..31 R's
  // (a strided pass and a 4-level shuffle butterfly) -- the separate block-sum launch and its wait are gone
  Jac tot = jac_inf();
  if (chunks > 1) {
    const int side = l >> 4, ll = l & 15;
    const size_t pp = live ? p : 0;
    const JacRaw *src = partials + (2 * pp + side) * chunks;
#pragma unroll 1
    for (size_t i = ll; i < chunks; i += 16) tot = jac_add(tot, raw_load(&src[i]));
#pragma unroll 1
    for (int off = 8; off > 0; off >>= 1) {
      Jac o;
#pragma unroll
      for (int t = 0; t < NL; t++) {
        o.X.v[t] = __shfl_xor(tot.X.v[t], off, 64);
        o.Y.v[t] = __shfl_xor(tot.Y.v[t], off, 64);
        o.Z.v[t] = __shfl_xor(tot.Z.v[t], off, 64);
      }
      tot = jac_add(tot, o);
    }
    // lane 1 converts R: it takes lane 16's total
    Jac r16;
#pragma unroll
    for (int t = 0; t < NL; t++) {
      r16.X.v[t] = __shfl(tot.X.v[t], base + 16, 64);
      r16.Y.v[t] = __shfl(tot.Y.v[t], base + 16, 64);
      r16.Z.v[t] = __shfl(tot.Z.v[t], base + 16, 64);
    }
    if (l == 1) tot = r16;
  }
  if (live && l < 2) {                     // L (lane 0) and R (lane 1) to canonical affine bytes
    Jac q = chunks > 1 ? tot : raw_load(&sums[2 * p + l]);
    if (!jac_is_inf(q) && is_zero_exact(q.Z)) q = jac_inf();
    uint32_t w[16];
    aff_to_boundary(w, jac_to_aff(q));
#pragma unroll
    for (int j = 0; j < 8; j++) {
      lr_xy[(2 * p + l) * 2].w[j] = w[j];
      lr_xy[(2 * p + l) * 2 + 1].w[j] = w[8 + j];
      S[4 + 8 * l + j] = (uint64_t)w[2 * j] | ((uint64_t)w[2 * j + 1] << 32);
    }
  }
  if (live && l < 4) S[l] = states[p * 4 + l];
  __syncthreads();
  const CoopK ck = coop_keccak_setup(l, base);
  // message words of append_message(label, 64-byte point): state (4 words) | flag byte, label (32 bytes), u32le(64), x || y --
  // 133 bytes, ONE rate block of 17 words: the label is shifted by one byte, the point by five (chain_hash's layout)
  uint64_t a = 0;
  for (int pt = 0; pt < 2; pt++) {
    const Label lab = pt == 0 ? TR_LABELS[LB_L] : TR_LABELS[LB_R];
    const uint64_t *P = S + 4 + 8 * pt;
    uint64_t w = 0;
    if (l < 4) w = S[l];
    else if (l == 4) w = lab.w[0] << 8;                                                       // flag byte 0x00
    else if (l < 8) w = (lab.w[l - 5] >> 56) | (lab.w[l - 4] << 8);
    else if (l == 8) w = (lab.w[3] >> 56) | (64ull << 8) | (P[0] << 40);
    else if (l < 16) w = (P[l - 9] >> 24) | (P[l - 8] << 40);
    else if (l == 16) w = (P[7] >> 24) | (0x01ull << 40) | 0x8000000000000000ULL;             // bytes 128..132, pad 0x01 .. 0x80
    a = coop_keccak_f(l < 17 ? w : 0ull, ck);
    __syncthreads();
    if (l < 4) S[l] = a;                                                    // the chain state after this append
    __syncthreads();
  }
  {   // challenge_bytes("u"): state = keccak256(state || 0x01 || pad_label("u")) -- 65 bytes, one block
    const Label lab = TR_LABELS[LB_u];
    uint64_t w = 0;
    if (l < 4) w = S[l];
    else if (l < 8) w = (lab.w[l - 4] << 8) | (l == 4 ? 0x01ull : (lab.w[l - 5] >> 56));
    else if (l == 8) w = (lab.w[3] >> 56) | (0x01ull << 8);                // last label byte, pad start at byte 65
    if (l == 16) w ^= 0x8000000000000000ULL;
    a = coop_keccak_f(l < 17 ? w : 0ull, ck);
    __syncthreads();
    if (l < 4) { S[l] = a; S[20 + l] = a; }
    __syncthreads();
    // hash_to_scalar: high = keccak256(low)
    uint64_t h = 0;
    if (l < 4) h = S[20 + l];
    if (l == 4) h = 0x01;
    if (l == 16) h = 0x8000000000000000ULL;
    a = coop_keccak_f(l < 17 ? h : 0ull, ck);
    __syncthreads();
    if (l < 4) S[4 + l] = a;                                               // (the point words are no longer needed)
    __syncthreads();
  }
  if (live && l == 0) {
    Words8 uw;
    wide_to_scalar(S + 20, S + 4, &uw);
    u_out[p] = uw;
    Fn ui = inv(load_plain(&uw));                                          // challenges are non-zero up to 2^-252
    store_plain(&uinv_out[p], ui);
#pragma unroll
    for (int j = 0; j < 4; j++) states[p * 4 + j] = S[j];
  }
}
void ipp_round_tail(hipStream_t st, size_t nb, const JacRaw *sums, uint64_t *states, Words8 *lr_xy, Words8 *u_out, Words8 *uinv_out,
                    const JacRaw *partials, size_t chunks) {
  if (!nb) return;
  hipLaunchKernelGGL(k_ipp_round_tail, dim3((nb + 1) / 2), dim3(64), 0, st, nb, sums, states, lr_xy, u_out, uinv_out, partials, chunks);
}
void ipp_round_challenge(hipStream_t st, size_t nb, uint64_t *states, const Words8 *lr, Words8 *u_out) {
  if (!nb) return;
  hipLaunchKernelGGL(k_ipp_round_challenge, dim3((nb + 63) / 64), dim3(64), 0, st, nb, states, lr, u_out);
}

}  // namespace bpk
