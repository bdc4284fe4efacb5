// k_ec.hip -- elliptic-curve kernels for gfx950 (wave64): point import/export, per-lane Straus
// scalar multiplication with LDS-resident tables, segmented point sums, signed fixed-window
// fixed-base tables + lookup MSM, verification tail.
//
// Hot-path rows: a1 StarkPoint::msm_iter / msm, a2 fold_witness (point half),
// a3 first-round generator scaling, a9 mega_check.  All integer work (F_p, 9 x 29-bit limbs); the
// kernels are VALU-integer bound (v_mad_u64_u32), not HBM bound -- DESIGN.md has the numbers.
#include <cstdio>
#include <cstdlib>
#include "fixed_body.cuh"
#include "vs_prep.cuh"
#include "ec29_quad.cuh"
#include "ec29_row.cuh"

using namespace bp;

namespace bpk {

// ------------------------------------------------------------------------------------------------
// Straus with signed 4-bit windows; one output per lane; per-lane tables of {1..8} * P_j.
// The tables live in an L2-resident global scratch laid out [block][point][entry][limb][lane]: a wave's
// load of one limb touches at most 8 rows (one per digit magnitude) of 256 contiguous bytes, a few
// hundred bytes per point addition against ~3 000 integer instructions -- and, unlike the 108 KB LDS
// image this replaces (1 block per CU), it leaves occupancy to the register file, so the Straus
// kernels of several in-flight batches overlap (profiles/: 1.75 ms/launch at 0.5 wave/SIMD before).
constexpr int SW = 4;                 // window bits
constexpr int SE = 1 << (SW - 1);     // 8 table entries
constexpr int STE = 36;   // int32 per table entry slot: X Y Z + a prefix product while normalising; x y afterwards
template <int NP, int TPB>
__device__ __forceinline__ void straus_body(const StrausArgs &a, JacRaw *out, size_t n, int32_t *tab_all, size_t blk) {
  const int tid = threadIdx.x;
  int32_t *tab = tab_all + blk * (NP * SE * STE * TPB);
  size_t i = blk * TPB + tid;
  const bool live = i < n;
  if (!live) i = n - 1;               // keep the wave uniform; result discarded
  uint32_t sp[NP][9];
  unsigned skip = 0;   // bit j, wave-uniform: every lane's j-th point is the identity (A_I2, A_O2, S2 of 1-phase proofs
                       // when the lanes of a wave share a role) -> no table, no additions for j
  unsigned pinf = 0;   // bit j, per lane: this lane's j-th point is the identity
  auto slot = [&](int j, int e) { return tab + ((size_t)(j * SE + e) * STE) * TPB + tid; };
  auto ld = [&](const int32_t *p, int off) { Fp r; for (int t = 0; t < NL; t++) r.v[t] = p[(off + t) * TPB]; return r; };
  auto st = [&](int32_t *p, int off, const Fp &x) { for (int t = 0; t < NL; t++) p[(off + t) * TPB] = x.v[t]; };
#pragma unroll
  for (int j = 0; j < NP; j++) {
    uint32_t s[8];
    const size_t pp = a.inner ? i / a.inner : 0, rr = a.inner ? i - pp * a.inner : i;
    const uint32_t *src = a.sc[j] + pp * a.sc_outer[j] + rr * a.sc_stride[j];
#pragma unroll
    for (int t = 0; t < 8; t++) s[t] = src[t];
    recode_add_k<SW>(sp[j], s);
    const AffDev *psrc = a.pts[j] + pp * a.pt_outer[j] + rr * a.pt_stride[j];
    Aff P;
    if (a.from_boundary) {   // points straight from the ABI bytes: canonical + on-curve checks here (k_points_from_boundary)
      uint32_t w[16];
#pragma unroll
      for (int t = 0; t < 16; t++) w[t] = psrc->w[t];
      if (!aff_from_boundary(P, w)) {
        if (live) { atomicOr(a.bad, 1); if (a.bad_inner) a.bad_inner[rr] = 1; }
        P.x = fe_zero<FP>();
        P.y = fe_zero<FP>();
      }
    } else {
      P = aff_load(psrc);
    }
    const bool inf = aff_is_inf(P);
    if (__all(inf)) { skip |= 1u << j; continue; }
    if (inf) pinf |= 1u << j;
    Jac m = jac_from_aff(P);
#pragma unroll 1
    for (int e = 0; e < SE; e++) {
      int32_t *dst = slot(j, e);
      st(dst, 0, m.X); st(dst, NL, m.Y); st(dst, 2 * NL, m.Z);
      if (e + 1 < SE) m = jac_madd(m, P);
    }
  }
  // each lane only ever reads back its own stores (same thread, program order): no fence needed
  if constexpr (NP >= 2) {
    // Affine tables: one inversion per lane (Montgomery's trick over the Z of entries 2P..8P of every point) turns
    // the 63 NP general additions of the main loop into mixed ones (~2 300 -> ~1 650 instructions each) for
    // ~80 k instructions of normalisation.  Entry 0 is affine already.  Identity points contribute Z = 1.
    Fp prod = fe_one<FP>();
#pragma unroll 1
    for (int j = 0; j < NP; j++) {
      if ((skip >> j) & 1) continue;
#pragma unroll 1
      for (int e = 1; e < SE; e++) {
        int32_t *p = slot(j, e);
        st(p, 3 * NL, prod);
        if (!((pinf >> j) & 1)) prod = fpmul(prod, ld(p, 2 * NL));
      }
    }
    Fp pinv = inv(prod);
#pragma unroll 1
    for (int j = NP - 1; j >= 0; j--) {
      if ((skip >> j) & 1) continue;
#pragma unroll 1
      for (int e = SE - 1; e >= 1; e--) {
        int32_t *p = slot(j, e);
        if ((pinf >> j) & 1) continue;
        Fp zi = fpmul(pinv, ld(p, 3 * NL));
        pinv = fpmul(pinv, ld(p, 2 * NL));
        Fp zi2 = fpsqr(zi);
        st(p, 0, fpmul(ld(p, 0), zi2));
        st(p, NL, fpmul(ld(p, NL), fpmul(zi2, zi)));
      }
    }
  }
  Jac acc = jac_inf();
  constexpr int W = num_windows<SW>();
#pragma unroll 1
  for (int w = W - 1; w >= 0; w--) {
    if (w != W - 1) {
#pragma unroll 1
      for (int d = 0; d < SW; d++) acc = jac_dbl(acc);
    }
#pragma unroll 1
    for (int j = 0; j < NP; j++) {
      if ((skip >> j) & 1) continue;
      int dg = recode_digit<SW>(sp[j], w);
      if (dg != 0 && !((pinf >> j) & 1)) {
        int e = (dg < 0 ? -dg : dg) - 1;
        const int32_t *src = slot(j, e);
        if constexpr (NP >= 2) {
          Aff q;
          q.x = ld(src, 0);
          q.y = ld(src, NL);
          if (dg < 0) q.y = neg(q.y);
          acc = jac_madd_nzq(acc, q);   // entries of a point that is not the identity (pinf is checked above)
        } else {
          Jac q;
          q.X = ld(src, 0); q.Y = ld(src, NL); q.Z = ld(src, 2 * NL);
          if (dg < 0) q.Y = neg(q.Y);
          acc = jac_add(acc, q);
        }
      }
    }
  }
  if (live) {
    size_t o = i;
    if (a.inner && a.out_outer) { size_t pp = i / a.inner; o = pp * a.out_outer + (i - pp * a.inner) * (a.out_stride ? a.out_stride : 1); }
    raw_store(&out[o], acc);
  }
}
template <int NP, int TPB>
__global__ void __launch_bounds__(TPB) k_straus(StrausArgs a, JacRaw *out, size_t n, int32_t *tab_all) {
  straus_body<NP, TPB>(a, out, n, tab_all, blockIdx.x);
}
size_t straus_scratch_bytes(int np, size_t n) {
  const size_t tpb = 64;
  return ((n + tpb - 1) / tpb) * (size_t)np * SE * STE * tpb * 4;
}
void straus(hipStream_t st, int np, const StrausArgs &a, JacRaw *out, size_t n, void *scratch) {
  if (!n) return;
  constexpr int TPB = 64;
  dim3 grid((n + TPB - 1) / TPB), blk(TPB);
  switch (np) {
    case 1: hipLaunchKernelGGL((k_straus<1, TPB>), grid, blk, 0, st, a, out, n, (int32_t *)scratch); break;
    case 2: hipLaunchKernelGGL((k_straus<2, TPB>), grid, blk, 0, st, a, out, n, (int32_t *)scratch); break;
    case 3: hipLaunchKernelGGL((k_straus<3, TPB>), grid, blk, 0, st, a, out, n, (int32_t *)scratch); break;
    case 4: hipLaunchKernelGGL((k_straus<4, TPB>), grid, blk, 0, st, a, out, n, (int32_t *)scratch); break;
    default: break;
  }
}

template <int TPB>
__global__ void __launch_bounds__(TPB) k_segmented_sum(const JacRaw *in, JacRaw *out, size_t n) {
  __shared__ int32_t smem[27 * (TPB / 2)];
  const size_t b = blockIdx.x;
  Jac acc = jac_inf();
  for (size_t i = threadIdx.x; i < n; i += TPB) acc = jac_add(acc, raw_load(&in[b * n + i]));
  acc = block_sum<TPB>(acc, smem);
  if (threadIdx.x == 0) raw_store(&out[b], acc);
}
void segmented_sum(hipStream_t st, const JacRaw *in, JacRaw *out, size_t nb, size_t n) {
  if (!nb) return;
  if (n <= 64) hipLaunchKernelGGL((k_segmented_sum<16>), dim3(nb), dim3(16), 0, st, in, out, n);
  else hipLaunchKernelGGL((k_segmented_sum<128>), dim3(nb), dim3(128), 0, st, in, out, n);
}

// ---- window-parallel variable-base part of the mega_check (the 11 + m + 2k proof points of every proof) -----------
// Straus with a doubling chain per lane pays 252 doublings for every 4 points.  Here the phases are split, and every
// launch carries whatever else of the batch is independent of it, so that one batch is a chain of six launches
// none of which waits on a serial 16-wave kernel alone:
//   front    [tables | prep]   tables: lane per TNP points (role-major): {1..8} P_j, normalised to affine with one
//                              inversion per lane, written as AffRaw rows tab[proof][point][entry]; the points come
//                              straight from the ABI bytes and are validated here (per-lane verdict in bad_lane).
//                              prep: the inversion pass of the scalar assembly (vs_prep.cuh), lane per proof.
//   scalars  k_verify_scalars (k_scalar.hip), wave per proof
//   windows  one WAVE per proof, lane w = window w (64 signed 4-bit windows): S_w = sum_j d_{j,w} P_j by mixed
//            additions from the tables -- no doublings
//   groups   lane per 8 windows: T_g = sum_{i<8} 16^i S_{8g+i}
//   back     [horner | fixed]  horner: one LANE per proof: sum_g 2^(32 g) T_g (the proof's 224 remaining dependent
//                              doublings: the longest link of the chain); fixed: the table-lookup MSMs over the
//                              generators (fixed_body.cuh), which need nothing but the scalars and hide behind it
//   verdict  lane per proof: variable-base sum + fixed-base partial, identity test, malformed-input bits, mega_check
// ~80 k wave-instructions per proof against ~113 k for the per-lane Straus.
struct AffRaw { int32_t v[2 * NL]; };   // raw Montgomery limbs x[9] y[9]; all zero = identity
struct TablesArgs {
  const AffDev *points;     // ABI bytes, nb x nvar x 64 B
  size_t nb, nvar, lanes;   // lanes = ceil(nvar / TNP) per proof
  AffRaw *tab;              // nb x nvar x 8
  int32_t *scratch;         // lane-strided staging: blocks x TNP x 8 x STE x 64 int32
  int *bad;                 // context-wide diagnostic flag
  int32_t *bad_lane;        // nb x lanes: 1 = one of the lane's points is malformed (every lane writes its entry)
  int converted;            // points are AffDev rows (validated, Montgomery form) instead of ABI bytes
};
template <int TNP>
__device__ __forceinline__ void tables_body(const TablesArgs &a, size_t blk) {
  constexpr int TPB = 64;
  const int tid = threadIdx.x;
  int32_t *stg = a.scratch + blk * (TNP * SE * STE * TPB);
  size_t i = blk * TPB + tid;
  const size_t n = a.nb * a.lanes;
  const bool live = i < n;
  if (!live) i = n - 1;
  const size_t r = i / a.nb, p = i - r * a.nb;        // role-major: a wave holds one role of 64 proofs
  auto slot = [&](int j, int e) { return stg + ((size_t)(j * SE + e) * STE) * TPB + tid; };
  auto ld = [&](const int32_t *q, int off) { Fp x; for (int t = 0; t < NL; t++) x.v[t] = q[(off + t) * TPB]; return x; };
  auto st = [&](int32_t *q, int off, const Fp &x) { for (int t = 0; t < NL; t++) q[(off + t) * TPB] = x.v[t]; };
  unsigned skip = 0, pinf = 0;   // bit j: wave-uniform / per-lane "point j is the identity (or beyond nvar)"
  bool malformed = false;
#pragma unroll
  for (int j = 0; j < TNP; j++) {
    const size_t v = r + (size_t)j * a.lanes;
    Aff P;
    P.x = fe_zero<FP>();
    P.y = fe_zero<FP>();
    if (v < a.nvar) {
      const AffDev *psrc = a.points + p * a.nvar + v;
      if (a.converted) P = aff_load(psrc);
      else {
        uint32_t w[16];
#pragma unroll
        for (int t = 0; t < 16; t++) w[t] = psrc->w[t];
        if (!aff_from_boundary(P, w)) {
          malformed = true;
          P.x = fe_zero<FP>();
          P.y = fe_zero<FP>();
        }
      }
    }
    const bool inf = aff_is_inf(P);
    if (__all(inf)) { skip |= 1u << j; continue; }
    if (inf) pinf |= 1u << j;
    Jac m = jac_from_aff(P);
#pragma unroll 1
    for (int e = 0; e < SE; e++) {
      int32_t *dst = slot(j, e);
      st(dst, 0, m.X); st(dst, NL, m.Y); st(dst, 2 * NL, m.Z);
      if (e + 1 < SE) m = jac_madd(m, P);
    }
  }
  if (live) {
    a.bad_lane[p * a.lanes + r] = malformed ? 1 : 0;
    if (malformed) atomicOr(a.bad, 1);
  }
  // Montgomery's trick over the Z of entries 2P..8P of the lane's points (entry 0 is affine already)
  Fp prod = fe_one<FP>();
#pragma unroll 1
  for (int j = 0; j < TNP; j++) {
    if ((skip >> j) & 1) continue;
#pragma unroll 1
    for (int e = 1; e < SE; e++) {
      int32_t *q = slot(j, e);
      st(q, 3 * NL, prod);
      if (!((pinf >> j) & 1)) prod = fpmul(prod, ld(q, 2 * NL));
    }
  }
  Fp pinv = inv(prod);
#pragma unroll 1
  for (int j = TNP - 1; j >= 0; j--) {
    const size_t v = r + (size_t)j * a.lanes;
    AffRaw *row = a.tab + (p * a.nvar + v) * SE;
    const bool dead = ((skip >> j) & 1) || ((pinf >> j) & 1);
#pragma unroll 1
    for (int e = SE - 1; e >= 0; e--) {
      Fp x = fe_zero<FP>(), y = fe_zero<FP>();
      if (!dead) {
        int32_t *q = slot(j, e);
        x = ld(q, 0);
        y = ld(q, NL);
        if (e >= 1) {
          Fp zi = fpmul(pinv, ld(q, 3 * NL));
          pinv = fpmul(pinv, ld(q, 2 * NL));
          Fp zi2 = fpsqr(zi);
          x = fpmul(x, zi2);
          y = fpmul(y, fpmul(zi2, zi));
        }
      }
      if (live && v < a.nvar) {
#pragma unroll
        for (int t = 0; t < NL; t++) { row[e].v[t] = x.v[t]; row[e].v[NL + t] = y.v[t]; }
      }
    }
  }
}
// tables (blocks [0, table_blocks)) and the inversion pass of the scalar assembly in one launch
template <int TNP>
__global__ void __launch_bounds__(64) k_verify_front(TablesArgs t, unsigned table_blocks, VsPrepArgs prep) {
  if (blockIdx.x < table_blocks) tables_body<TNP>(t, blockIdx.x);
  else vs_prep_body(prep, blockIdx.x - table_blocks);
}
// one wave per proof, lane = window
__global__ void __launch_bounds__(64) k_verify_windows(const AffRaw *tab, const uint32_t *scalars /* nb x nvar x 8 words */,
                                                       size_t nvar, JacRaw *winsum) {
  const size_t p = blockIdx.x;
  const int w = threadIdx.x;
  Xyzz acc = xyzz_inf();       // additions only: extended-Jacobian accumulator (ec29.cuh)
#pragma unroll 1
  for (size_t v = 0; v < nvar; v++) {
    uint32_t s[8], sp[9];
    const uint32_t *src = scalars + (p * nvar + v) * 8;
#pragma unroll
    for (int t = 0; t < 8; t++) s[t] = src[t];
    recode_add_k<SW>(sp, s);
    const int dg = recode_digit<SW>(sp, w);
    if (dg != 0) {
      const AffRaw *e = tab + (p * nvar + v) * SE + ((dg < 0 ? -dg : dg) - 1);
      Aff q;
#pragma unroll
      for (int t = 0; t < NL; t++) { q.x.v[t] = e->v[t]; q.y.v[t] = e->v[NL + t]; }
      if (!aff_is_inf(q)) {
        if (dg < 0) q.y = neg(q.y);
        acc = xyzz_madd_nzq(acc, q);
      }
    }
  }
  raw_store(&winsum[p * 64 + w], xyzz_to_jac(acc));
}
// Horner over the 64 window sums in two stages.  Stage 1, one lane per (proof, group of 8 windows):
// T_g = sum_{i<8} 16^i S_{8g+i} (28 doublings, 7 additions), written over S_{8g}.  Stage 2, one lane per proof:
// sum_g 2^(32 g) T_g (224 doublings, 7 additions).  The dependency chain of a batch loses 49 of its 63 general
// additions for 8 x 28 extra doublings per proof (+3 % instructions).
constexpr int HG = 8;   // windows per group
__global__ void __launch_bounds__(64) k_verify_horner_groups(JacRaw *winsum, size_t nb) {
  __builtin_amdgcn_s_setprio(2);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb * (64 / HG)) return;
  JacRaw *s = winsum + t * HG;          // = winsum[p * 64 + g * HG]
  Jac acc = raw_load(&s[HG - 1]);
#pragma unroll 1
  for (int i = HG - 2; i >= 0; i--) {
#pragma unroll 1
    for (int d = 0; d < SW; d++) acc = jac_dbl(acc);
    acc = jac_add(acc, raw_load(&s[i]));
  }
  raw_store(&s[0], acc);
}
// stage 1 with a DPP quad per (proof, group): the 28 doublings and 7 additions at the quad's depth (latency mode: 4x the lanes
// and ~2.3x the instructions of this small stage for ~40 % of its time)
__global__ void __launch_bounds__(64) k_verify_horner_groups4(JacRaw *winsum, size_t nb) {
  __builtin_amdgcn_s_setprio(2);
  const int role = threadIdx.x & 3;
  size_t t = (size_t)blockIdx.x * 16 + (threadIdx.x >> 2);
  const size_t units = nb * (64 / HG);
  const bool live = t < units;
  if (!live) t = units - 1;            // whole quads stay active
  JacRaw *s = winsum + t * HG;
  JacT acc = jact_from_jac(raw_load(&s[HG - 1]));
#pragma unroll 1
  for (int i = HG - 2; i >= 0; i--) {
#pragma unroll 1
    for (int d = 0; d < SW; d++) acc = q4_dbl(acc, role);
    acc = q4_add(acc, jact_from_jac(raw_load(&s[i])), role);
  }
  // (a clamped quad recomputes the last unit; only live quads store, after every load of the unit's slot 0 is done)
  if (live && role == 0) raw_store(&s[0], jact_to_jac(acc));
}
// stage 1 with a whole WAVE per (proof, group) (ec29_row.cuh): 28 doublings + 7 additions in ~23 us instead of ~65 us on a quad --
// for the few groups of a small MSM or a handful of proofs, where this stage is a link of a lone chain
__global__ void __launch_bounds__(64) k_verify_horner_groups_row(JacRaw *winsum) {
  __builtin_amdgcn_s_setprio(2);
  const RowK K = rowk_init();
  JacRaw *s = winsum + (size_t)blockIdx.x * HG;
  JacR acc = jacr_from_limbs(K, s[HG - 1].v);
#pragma unroll 1
  for (int i = HG - 2; i >= 0; i--) {
#pragma unroll 1
    for (int d = 0; d < SW; d++) acc = rdbl(K, acc);
    acc = radd(K, acc, jacr_addend_from_limbs(K, s[i].v));
  }
  jacr_store(K, s[0].v, acc);      // (slot 0 was this wave's last load)
}
// one lane per proof over `count` partial sums `stride` slots apart, `dbl` doublings between them
struct HornerArgs { const JacRaw *winsum; JacRaw *varsum; size_t nb; int count, stride, dbl, quad; };
__device__ __forceinline__ void horner_body(const HornerArgs &h, size_t blk) {
  __builtin_amdgcn_s_setprio(2);   // 16 waves carrying the longest link of the chain
  size_t p = blk * 64 + threadIdx.x;
  if (p >= h.nb) return;
  Jac acc = raw_load(&h.winsum[p * 64 + (size_t)(h.count - 1) * h.stride]);
#pragma unroll 1
  for (int w = h.count - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < h.dbl; d++) acc = jac_dbl(acc);
    acc = jac_add(acc, raw_load(&h.winsum[p * 64 + (size_t)w * h.stride]));
  }
  raw_store(&h.varsum[p], acc);
}
// The same pass with the four lanes of a DPP quad per proof (ec29_quad.cuh): a doubling is 3 field multiplications
// deep instead of 9, a general addition 5 instead of 16 -- the longest link of a batch's chain shortened ~1.7x for
// 4x the lanes of a 16-wave kernel.  16 proofs per wave.
__device__ __forceinline__ void horner4_body(const HornerArgs &h, size_t blk) {
  __builtin_amdgcn_s_setprio(2);
  const int role = threadIdx.x & 3;
  size_t p = blk * 16 + (threadIdx.x >> 2);
  const bool live = p < h.nb;
  if (!live) p = h.nb - 1;             // whole quads stay active
  JacT acc = jact_from_jac(raw_load(&h.winsum[p * 64 + (size_t)(h.count - 1) * h.stride]));
#pragma unroll 1
  for (int w = h.count - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < h.dbl; d++) acc = q4_dbl(acc, role);
    acc = q4_add(acc, jact_from_jac(raw_load(&h.winsum[p * 64 + (size_t)w * h.stride])), role);
  }
  if (live && role == 0) raw_store(&h.varsum[p], jact_to_jac(acc));
}
// The same pass with a whole WAVE per proof (ec29_row.cuh: one field multiplication spread over the 16 lanes of a DPP row, the
// four rows = the four products of a level): a doubling is ~270 instructions deep instead of 675 -- 0.58 us against 1.35 us --
// for 16x the lanes of the quad form.  The form for a chain that has the chip to itself: a lone batch (latency mode), the MSMs.
__device__ __forceinline__ void horner_row_body(const HornerArgs &h, size_t blk) {
  __builtin_amdgcn_s_setprio(2);
  const RowK K = rowk_init();
  const size_t p = blk;
  JacR acc = jacr_from_limbs(K, h.winsum[p * 64 + (size_t)(h.count - 1) * h.stride].v);
#pragma unroll 1
  for (int w = h.count - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < h.dbl; d++) acc = rdbl(K, acc);
    acc = radd(K, acc, jacr_addend_from_limbs(K, h.winsum[p * 64 + (size_t)w * h.stride].v));
  }
  jacr_store(K, h.varsum[p].v, acc);
}
// the Horner lanes (blocks [0, horner_blocks)) and the small fixed-base MSMs in one launch
template <int C, int LPM>
__global__ void __launch_bounds__(64) k_verify_back(HornerArgs h, unsigned horner_blocks, FixedSmallArgs f) {
  if (blockIdx.x < horner_blocks) {
    if (h.quad == 2) horner_row_body(h, blockIdx.x);
    else if (h.quad) horner4_body(h, blockIdx.x);
    else horner_body(h, blockIdx.x);
  }
  else fixed_small_body<C, LPM>(f.table, f.n, f.cap, f.scalars, f.sc_stride, f.out, f.nb, blockIdx.x - horner_blocks);
}
// the same launch with the generator half walked a proof per lane, a run of generators per wave (fixed_chunk_body)
template <int C, int AHEAD>
__global__ void __launch_bounds__(64) k_verify_back_q(HornerArgs h, unsigned horner_blocks, FixedSmallArgs f) {
  if (blockIdx.x < horner_blocks) {
    if (h.quad == 2) horner_row_body(h, blockIdx.x);
    else if (h.quad) horner4_body(h, blockIdx.x);
    else horner_body(h, blockIdx.x);
  }
  else fixed_chunk_body<C, AHEAD>(f.table, f.n, f.cap, f.scalars, f.sc_stride, f.out, f.nb, f.chunks, f.gens_per_chunk, blockIdx.x - horner_blocks);
}
// lane per proof: variable-base sum + fixed-base partial; ok = identity and every input of the proof well-formed
__global__ void __launch_bounds__(64) k_verify_verdict(const JacRaw *varsum, const JacRaw *fixed, size_t nb, const int32_t *bad_lane,
                                                       size_t lanes, const int32_t *bad_sc, int32_t *ok, Words8 *mega) {
  __builtin_amdgcn_s_setprio(2);
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nb) return;
  Jac acc = jac_add(raw_load(&varsum[p]), raw_load(&fixed[p]));
  const bool inf = jac_is_inf(acc) || is_zero_exact(acc.Z);
  int malformed = bad_sc ? bad_sc[p] : 0;
  for (size_t l = 0; l < lanes; l++) malformed |= bad_lane[p * lanes + l];
  ok[p] = (inf && !malformed) ? 1 : 0;
  if (mega) {
    uint32_t w[16];
    if (inf) {
#pragma unroll
      for (int j = 0; j < 16; j++) w[j] = 0;
    } else {
      aff_to_boundary(w, jac_to_aff(acc));
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { mega[2 * p].w[j] = w[j]; mega[2 * p + 1].w[j] = w[8 + j]; }
  }
}
// the verdict over `chunks` partial sums of the generator half per proof (k_verify_back_q): R lanes per proof add chunks / R partials
// each, a butterfly over the R lanes, lane 0 of the proof decides
__global__ void __launch_bounds__(64) k_verify_verdict_q(const JacRaw *varsum, const JacRaw *part, unsigned chunks, unsigned R, size_t nb,
                                                         const int32_t *bad_lane, size_t lanes, const int32_t *bad_sc, int32_t *ok, Words8 *mega) {
  __builtin_amdgcn_s_setprio(2);
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t p = t / R;
  const unsigned r = (unsigned)(t - p * R);
  const bool live = p < nb;
  if (!live) p = nb - 1;
  Jac acc = r < chunks ? raw_load(&part[p * chunks + r]) : jac_inf();
#pragma unroll 1
  for (unsigned q = r + R; q < chunks; q += R) acc = jac_add(acc, raw_load(&part[p * chunks + q]));
#pragma unroll 1
  for (unsigned off = R / 2; off > 0; off >>= 1) {
    Jac o;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      o.X.v[j] = __shfl_xor(acc.X.v[j], off, 64);
      o.Y.v[j] = __shfl_xor(acc.Y.v[j], off, 64);
      o.Z.v[j] = __shfl_xor(acc.Z.v[j], off, 64);
    }
    acc = jac_add(acc, o);
  }
  if (r != 0 || !live) return;
  acc = jac_add(raw_load(&varsum[p]), acc);
  const bool inf = jac_is_inf(acc) || is_zero_exact(acc.Z);
  int malformed = bad_sc ? bad_sc[p] : 0;
  for (size_t l = 0; l < lanes; l++) malformed |= bad_lane[p * lanes + l];
  ok[p] = (inf && !malformed) ? 1 : 0;
  if (mega) {
    uint32_t w[16];
    if (inf) {
#pragma unroll
      for (int j = 0; j < 16; j++) w[j] = 0;
    } else {
      aff_to_boundary(w, jac_to_aff(acc));
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { mega[2 * p].w[j] = w[j]; mega[2 * p + 1].w[j] = w[8 + j]; }
  }
}
// points per table lane: 8 = fewest instructions (one inversion per 8 points), 4 (default) = half the dependency chain of
// the front launch for +1.3 % instructions per batch, 1 (latency mode) = 7 additions + one inversion per lane, the shortest
// chain (a lone batch: 0.665 ms against 0.715 ms with 2).  VerifyWp::table_np (BPGPU_OPT_TABLE_NP) overrides.
static int wp_tnp(const VerifyWp &v) {
  const int t = v.table_np ? v.table_np : (v.latency_mode ? 1 : 4);
  return t == 8 ? 8 : (t == 2 ? 2 : (t == 1 ? 1 : 4));
}
struct WpLayout { TablesArgs t; JacRaw *winsum, *varsum; unsigned blocks; size_t bytes; };
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
size_t verify_wp_scratch_bytes(size_t nb, size_t nvar) {
  // the layout depends on the points per table lane (wp_tnp): fewer points per lane = more lanes, but the staging is per
  // BLOCK of 64 lanes, so for a handful of proofs the block rounding makes the larger tnp the larger staging -- take the
  // maximum of each part over the three lane shapes
  size_t stage = 0, lanes_max = 0;
  for (size_t tnp : {(size_t)1, (size_t)2, (size_t)4, (size_t)8}) {
    const size_t lanes = (nvar + tnp - 1) / tnp, blocks = (nb * lanes + 63) / 64;
    const size_t b = al256(blocks * tnp * SE * STE * 64 * 4);
    if (b > stage) stage = b;
    if (lanes > lanes_max) lanes_max = lanes;
  }
  return stage + al256(nb * nvar * SE * sizeof(AffRaw)) + al256(nb * 64 * sizeof(JacRaw)) + al256(nb * sizeof(JacRaw)) +
         al256(nb * lanes_max * 4);
}
bool verify_wp_supported(size_t nb, size_t nvar, int c, size_t n) {
  const size_t total = (2 + 2 * n) * (252 / c + 1);
  return nb >= 1 && total <= 16384 && nvar && (c == 8 || c == 16 || c == 20);
}
static WpLayout wp_layout(const VerifyWp &v) {
  WpLayout L{};
  const size_t tnp = wp_tnp(v);
  L.t.points = v.points_abi; L.t.converted = v.points_converted ? 1 : 0; L.t.nb = v.nb; L.t.nvar = v.nvar; L.t.lanes = (v.nvar + tnp - 1) / tnp; L.t.bad = v.bad;
  const size_t nblk = (v.nb * L.t.lanes + 63) / 64;
  uint8_t *sp = (uint8_t *)v.scratch;
  L.t.scratch = (int32_t *)sp; sp += al256(nblk * tnp * SE * STE * 64 * 4);
  L.t.tab = (AffRaw *)sp; sp += al256(v.nb * v.nvar * SE * sizeof(AffRaw));
  L.winsum = (JacRaw *)sp; sp += al256(v.nb * 64 * sizeof(JacRaw));
  L.varsum = (JacRaw *)sp; sp += al256(v.nb * sizeof(JacRaw));
  L.t.bad_lane = (int32_t *)sp; sp += al256(v.nb * L.t.lanes * 4);
  L.blocks = (unsigned)nblk;
  L.bytes = (size_t)(sp - (uint8_t *)v.scratch);
  return L;
}
bool verify_wp_layout_fits(const VerifyWp &v) { return wp_layout(v).bytes <= verify_wp_scratch_bytes(v.nb, v.nvar); }
// tables of the proof points | inversion pass of the scalar assembly (with_prep = false: tables only)
void verify_wp_front_launch(hipStream_t st, const VerifyWp &v, const VerifyDims &d, const Words8 *challenges, int32_t *aux,
                            size_t aux_stride, bool with_prep, const Words8 *fast_proof_scalars, Words8 *fast_fixed_sc, Words8 *fast_var_sc,
                            Words8 *fast_full_sc) {
  VsPrepArgs prep{d, challenges, aux, aux_stride};
  prep.proof_scalars = fast_proof_scalars; prep.fixed_sc = fast_fixed_sc; prep.var_sc = fast_var_sc; prep.full_sc = fast_full_sc;
  if (!with_prep) prep.d.nb = 0;
  WpLayout L = wp_layout(v);
  const unsigned pb = (unsigned)((prep.d.nb + 63) / 64);
  switch (wp_tnp(v)) {
    case 8: hipLaunchKernelGGL((k_verify_front<8>), dim3(L.blocks + pb), dim3(64), 0, st, L.t, L.blocks, prep); break;
    case 2: hipLaunchKernelGGL((k_verify_front<2>), dim3(L.blocks + pb), dim3(64), 0, st, L.t, L.blocks, prep); break;
    case 1: hipLaunchKernelGGL((k_verify_front<1>), dim3(L.blocks + pb), dim3(64), 0, st, L.t, L.blocks, prep); break;
    default: hipLaunchKernelGGL((k_verify_front<4>), dim3(L.blocks + pb), dim3(64), 0, st, L.t, L.blocks, prep); break;
  }
}
// var_scalars: nb x nvar x 8 words in operand order
void verify_wp_windows(hipStream_t st, const VerifyWp &v, const uint32_t *var_scalars) {
  WpLayout L = wp_layout(v);
  hipLaunchKernelGGL(k_verify_windows, dim3(v.nb), dim3(64), 0, st, L.t.tab, var_scalars, v.nvar, L.winsum);
}
static bool wp_grouped() { return true; }   // (two-stage Horner: the one-stage pass stays in the code for reference)
void verify_wp_groups(hipStream_t st, const VerifyWp &v) {
  static_assert(num_windows<SW>() == 64, "64 window sums per proof");
  if (!wp_grouped()) return;
  WpLayout L = wp_layout(v);
  const bool quad = v.groups_form ? v.groups_form == 2 : v.latency_mode;   // BPGPU_OPT_GROUPS_FORM: 0 = by mode, 1 = lane, 2 = quad, 3 = wave per group of 8 windows
  const size_t units = v.nb * (64 / HG);
  if (v.groups_form == 3 || (v.groups_form == 0 && v.latency_mode && v.horner_form == 0 && units <= v.row_max)) {
    hipLaunchKernelGGL(k_verify_horner_groups_row, dim3((unsigned)units), dim3(64), 0, st, L.winsum);
    return;
  }
  if (quad) hipLaunchKernelGGL(k_verify_horner_groups4, dim3((v.nb * (64 / HG) + 15) / 16), dim3(64), 0, st, L.winsum, v.nb);
  else hipLaunchKernelGGL(k_verify_horner_groups, dim3((v.nb * (64 / HG) + 63) / 64), dim3(64), 0, st, L.winsum, v.nb);
}
static size_t wp_chunk_gens(const VerifyWp &v);
template <int C>
static void launch_back(hipStream_t st, const HornerArgs &h, unsigned hb, const FixedSmallArgs &f, bool latency_mode, int lpm_opt) {
  if (f.chunks) {
    // (rows two and three pairs ahead of the addition instead of one: the same rate, profiles/r04_burst20_sweep2.log)
    hipLaunchKernelGGL((k_verify_back_q<C, 1>), dim3(hb + (unsigned)((f.nb + 63) / 64) * f.chunks), dim3(64), 0, st, h, hb, f);
    return;
  }
  // lanes per fixed-base MSM: 16 = fewest instructions (4 butterfly levels), 32 = half the serial additions per lane
  // (the fixed-base lanes are the longest link of this launch once the Horner pass runs on quads); BPGPU_OPT_FIXED_LPM overrides
  const int lpm = lpm_opt == 16 || lpm_opt == 32 || lpm_opt == 64 ? lpm_opt : (f.nb >= 1024 && !latency_mode ? 16 : 32);
  if (lpm == 16) hipLaunchKernelGGL((k_verify_back<C, 16>), dim3(hb + (unsigned)((f.nb + 3) / 4)), dim3(64), 0, st, h, hb, f);
  else if (lpm == 32) hipLaunchKernelGGL((k_verify_back<C, 32>), dim3(hb + (unsigned)((f.nb + 1) / 2)), dim3(64), 0, st, h, hb, f);
  else hipLaunchKernelGGL((k_verify_back<C, 64>), dim3(hb + (unsigned)f.nb), dim3(64), 0, st, h, hb, f);
}
// Horner pass over the window sums | table-lookup MSMs over the generators (fixed scalars as for fixed_msm)
void verify_wp_back(hipStream_t st, const VerifyWp &v, int c, const AffDev *table, size_t n, size_t cap,
                    const uint32_t *fixed_scalars, size_t sc_stride, JacRaw *out_fixed) {
  WpLayout L = wp_layout(v);
  // v.horner_form: 0 = by mode (a wave per proof when the chain has the chip to itself: latency mode and at most v.row_max
  // proofs / groups; else a quad per proof), 1 = lane, 2 = quad, 3 = wave (row form)
  int quad = v.horner_form == 1 ? 0 : (v.horner_form == 3 ? 2 : 1);
  if (v.horner_form == 0 && v.latency_mode && v.nb <= v.row_max) quad = 2;
  HornerArgs h{L.winsum, L.varsum, v.nb, 64, 1, SW, quad};
  if (wp_grouped()) { h.count = 64 / HG; h.stride = HG; h.dbl = SW * HG; }
  const unsigned hb = (unsigned)(quad == 2 ? v.nb : (quad ? (v.nb + 15) / 16 : (v.nb + 63) / 64));
  FixedSmallArgs f{table, n, cap, fixed_scalars, sc_stride, out_fixed, v.nb};
  if (table && verify_wp_fixed_parts(v, n) > 1) { f.chunks = (unsigned)verify_wp_fixed_parts(v, n); f.gens_per_chunk = (unsigned)wp_chunk_gens(v); }
  if (!table) {   // the generator half ran as its own launch: Horner pass only
    f.nb = 0;
    launch_back<8>(st, h, hb, f, v.latency_mode, v.fixed_lpm);
    return;
  }
  if (c == 8) launch_back<8>(st, h, hb, f, v.latency_mode, v.fixed_lpm);
  else if (c == 16) launch_back<16>(st, h, hb, f, v.latency_mode, v.fixed_lpm);
  else launch_back<20>(st, h, hb, f, v.latency_mode, v.fixed_lpm);
}
// A large MSM cut into `per` instances of G points (msm_wp_batch): the 64 window sums of an MSM's instances are added up BEFORE the
// Horner stages, so that those run once per MSM instead of once per instance with a sum of `per` results behind them (2^14 terms:
// 1 024 Horner chains + a 1 024-point sum -> one reduction of 64 x 1 024 points + one chain).  v: the instances (v.nb = v2.nb * per),
// v2: the MSMs; block per (MSM, window).
template <int TPB>
__global__ void __launch_bounds__(TPB) k_winsum_reduce(const JacRaw *in, JacRaw *out, size_t per) {
  __shared__ int32_t smem[27 * (TPB / 2)];
  const size_t b = blockIdx.x >> 6, w = blockIdx.x & 63;
  Jac acc = jac_inf();
  for (size_t i = threadIdx.x; i < per; i += TPB) acc = jac_add(acc, raw_load(&in[(b * per + i) * 64 + w]));
  acc = block_sum<TPB>(acc, smem);
  if (threadIdx.x == 0) raw_store(&out[b * 64 + w], acc);
}
void verify_wp_reduce_instances(hipStream_t st, const VerifyWp &v, const VerifyWp &v2, size_t per) {
  const JacRaw *in = wp_layout(v).winsum;
  JacRaw *out = wp_layout(v2).winsum;
  const unsigned grid = (unsigned)(v2.nb * 64);
  if (per <= 32) hipLaunchKernelGGL((k_winsum_reduce<16>), dim3(grid), dim3(16), 0, st, in, out, per);
  else if (per <= 128) hipLaunchKernelGGL((k_winsum_reduce<64>), dim3(grid), dim3(64), 0, st, in, out, per);
  else hipLaunchKernelGGL((k_winsum_reduce<128>), dim3(grid), dim3(128), 0, st, in, out, per);
}
const JacRaw *verify_wp_varsum(const VerifyWp &v) { return wp_layout(v).varsum; }
// generators per wave of the proof-per-lane walk of the generator half (0 = the lanes-per-proof form): by default 5 (26 runs of a
// 64-bit range proof's 130 generators: 65 additions per lane) once the batch fills its waves -- against 16 lanes per proof and a butterfly
// +2.4 % sustained and +5 % in a 20-step burst (profiles/r04_burst20_sweep2.log); a lone batch in latency mode keeps the lanes-per-proof form
// (its generator half runs beside the chain on the second stream) and so does a batch of a few proofs, whose waves would be mostly dead lanes
static size_t wp_chunk_gens(const VerifyWp &v) {
  if (v.fixed_chunk_gens < 0) return 0;
  if (v.fixed_chunk_gens > 0) return (size_t)v.fixed_chunk_gens;
  return !v.latency_mode && v.nb >= 256 ? 5 : 0;
}
size_t verify_wp_fixed_parts(const VerifyWp &v, size_t n) {
  const size_t g = wp_chunk_gens(v);
  return g ? (2 + 2 * n + g - 1) / g : 1;
}
void verify_wp_verdict(hipStream_t st, const VerifyWp &v, const JacRaw *fixed, int32_t *ok, Words8 *mega, size_t parts) {
  WpLayout L = wp_layout(v);
  if (parts > 1) {
    const unsigned R = parts > 8 ? 8 : 4;   // lanes per proof: the verdict is the last link of a batch's chain -- 26 partials: 3 + 3 + 1 additions deep on 8 lanes, 6 + 2 + 1 on 4
    hipLaunchKernelGGL(k_verify_verdict_q, dim3((unsigned)((v.nb * R + 63) / 64)), dim3(64), 0, st, L.varsum, fixed, (unsigned)parts, R, v.nb,
                       L.t.bad_lane, L.t.lanes, v.bad_sc, ok, mega);
    return;
  }
  hipLaunchKernelGGL(k_verify_verdict, dim3((v.nb + 63) / 64), dim3(64), 0, st, L.varsum, fixed, v.nb, L.t.bad_lane, L.t.lanes,
                     v.bad_sc, ok, mega);
}

// Both halves of a batch's mega_check MSM in ONE launch: blocks [0, straus_blocks) run the per-lane Straus over the
// proof points, the rest the table-lookup MSMs over the generators.  A single 1024-proof batch gives either part
// only 100-250 waves for 1024 SIMDs and a stream runs one kernel at a time, so launching them together doubles the
// waves each in-flight batch keeps on the chip.
template <int NP, int C, int LPM>
__global__ void __launch_bounds__(64) k_verify_msm(StrausArgs a, JacRaw *out, size_t n, int32_t *tab_all,
                                                   unsigned straus_blocks, FixedSmallArgs f) {
  if (blockIdx.x < straus_blocks) {
    // the Straus lanes carry the launch's longest dependency chain (252 doublings + 63 NP additions against ~130
    // table additions per fixed-base lane): let them win the issue arbitration, the fixed-base waves fill the gaps
    if (a.prio) __builtin_amdgcn_s_setprio(1);
    straus_body<NP, 64>(a, out, n, tab_all, blockIdx.x);
  }
  else fixed_small_body<C, LPM>(f.table, f.n, f.cap, f.scalars, f.sc_stride, f.out, f.nb, blockIdx.x - straus_blocks);
}
template <int NP, int C>
static void launch_verify_msm(hipStream_t st, const StrausArgs &a, JacRaw *out, size_t n, void *scratch, const FixedSmallArgs &f) {
  unsigned sb = (unsigned)((n + 63) / 64);
  if (f.nb >= 1024) hipLaunchKernelGGL((k_verify_msm<NP, C, 16>), dim3(sb + (unsigned)((f.nb + 3) / 4)), dim3(64), 0, st, a, out, n, (int32_t *)scratch, sb, f);
  else hipLaunchKernelGGL((k_verify_msm<NP, C, 32>), dim3(sb + (unsigned)((f.nb + 1) / 2)), dim3(64), 0, st, a, out, n, (int32_t *)scratch, sb, f);
}
bool verify_msm_fused(hipStream_t st, int np, const StrausArgs &a, JacRaw *out_var, size_t n_lanes, void *scratch, int c,
                      const AffDev *table, size_t n, size_t cap, const uint32_t *scalars, size_t sc_stride,
                      JacRaw *out_fixed, size_t nb) {
  const size_t total = (2 + 2 * n) * (252 / c + 1);
  if (!n_lanes || nb < 64 || total > 16384) return false;
  FixedSmallArgs f{table, n, cap, scalars, sc_stride, out_fixed, nb};
  // (one instantiation per table window: 4 points per lane; other settings take the separate launches)
  if (c == 16 && np == 4) launch_verify_msm<4, 16>(st, a, out_var, n_lanes, scratch, f);
  else if (c == 20 && np == 4) launch_verify_msm<4, 20>(st, a, out_var, n_lanes, scratch, f);
  else if (c == 8 && np == 4) launch_verify_msm<4, 8>(st, a, out_var, n_lanes, scratch, f);
  else return false;
  return true;
}
}  // namespace bpk
