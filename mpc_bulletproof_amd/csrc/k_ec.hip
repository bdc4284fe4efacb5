// k_ec.hip -- elliptic-curve kernels for gfx950 (wave64): point import/export, per-lane Straus
// scalar multiplication with LDS-resident tables, segmented point sums, signed fixed-window
// fixed-base tables + lookup MSM, verification tail.
//
// Hot-path rows: a1 StarkPoint::msm_iter / msm, a2 fold_witness (point half),
// a3 first-round generator scaling, a9 mega_check.  All integer work (F_p, 9 x 29-bit limbs); the
// kernels are VALU-integer bound (v_mad_u64_u32), not HBM bound -- DESIGN.md has the numbers.
#include <cstdlib>
#include "fixed_body.cuh"

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
        if (live) atomicOr(a.bad, 1);
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
// Straus with a doubling chain per lane pays 252 doublings for every 4 points.  Here the three phases are split:
//   tables   lane per 4 points (role-major): {1..8} P_j, normalised to affine with one inversion per lane,
//            written as AffRaw rows tab[proof][point][entry]; the points come straight from the ABI bytes
//   windows  one WAVE per proof, lane w = window w (64 signed 4-bit windows): S_w = sum_j d_{j,w} P_j by mixed
//            additions from the tables -- no doublings
//   horner   one LANE per proof: sum_w 16^w S_w with the proof's only 252 doublings, + the fixed-base partial,
//            identity test (replaces k_verify_finalize)
// ~80 k wave-instructions per proof against ~113 k for the per-lane Straus, and a shorter kernel chain.
struct AffRaw { int32_t v[2 * NL]; };   // raw Montgomery limbs x[9] y[9]; all zero = identity
struct TablesArgs {
  const AffDev *points;     // ABI bytes, nb x nvar x 64 B
  size_t nb, nvar, lanes;   // lanes = ceil(nvar / 4) per proof
  AffRaw *tab;              // nb x nvar x 8
  int32_t *scratch;         // lane-strided staging: blocks x 4 x 8 x STE x 64 int32
  int *bad;
};
constexpr int TNP = 8;
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
#pragma unroll
  for (int j = 0; j < TNP; j++) {
    const size_t v = r + (size_t)j * a.lanes;
    Aff P;
    P.x = fe_zero<FP>();
    P.y = fe_zero<FP>();
    if (v < a.nvar) {
      const AffDev *psrc = a.points + p * a.nvar + v;
      uint32_t w[16];
#pragma unroll
      for (int t = 0; t < 16; t++) w[t] = psrc->w[t];
      if (!aff_from_boundary(P, w)) {
        if (live) atomicOr(a.bad, 1);
        P.x = fe_zero<FP>();
        P.y = fe_zero<FP>();
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
// one wave per proof, lane = window
__global__ void __launch_bounds__(64) k_verify_windows(const AffRaw *tab, const uint32_t *scalars /* nb x nvar x 8 words */,
                                                       size_t nvar, JacRaw *winsum) {
  const size_t p = blockIdx.x;
  const int w = threadIdx.x;
  Jac acc = jac_inf();
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
        acc = jac_madd_nzq(acc, q);
      }
    }
  }
  raw_store(&winsum[p * 64 + w], acc);
}
// Horner over the 64 window sums in two stages.  Stage 1, one lane per (proof, group of 8 windows):
// T_g = sum_{i<8} 16^i S_{8g+i} (28 doublings, 7 additions), written over S_{8g}.  Stage 2, one lane per proof:
// sum_g 2^(32 g) T_g (224 doublings, 7 additions) + the fixed-base partial, identity test.  The dependency chain of a
// batch loses 49 of its 63 general additions for 8 x 28 extra doublings per proof (+3 % instructions): a single batch
// takes 1.70 instead of 1.84 ms and bursts of 4 run at 1.58 instead of 1.27 M/s; the steady state is unchanged (it is
// bound by instruction issue, not by the length of the chain).  BPGPU_HORNER_GROUPS=0 selects the one-stage pass.
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
// one lane per proof over `count` partial sums `stride` slots apart, `dbl` doublings between them
__global__ void __launch_bounds__(64) k_verify_horner(const JacRaw *winsum, const JacRaw *fixed, size_t nb, int32_t *ok,
                                                      Words8 *mega, int count, int stride, int dbl) {
  __builtin_amdgcn_s_setprio(2);   // 16 waves carrying the longest link of the chain
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nb) return;
  Jac acc = raw_load(&winsum[p * 64 + (size_t)(count - 1) * stride]);
#pragma unroll 1
  for (int w = count - 2; w >= 0; w--) {
#pragma unroll 1
    for (int d = 0; d < dbl; d++) acc = jac_dbl(acc);
    acc = jac_add(acc, raw_load(&winsum[p * 64 + (size_t)w * stride]));
  }
  acc = jac_add(acc, raw_load(&fixed[p]));
  bool inf = jac_is_inf(acc) || is_zero_exact(acc.Z);
  ok[p] = inf ? 1 : 0;
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
// tables (blocks [0, table_blocks)) and the small fixed-base MSMs in one launch
template <int C, int LPM>
__global__ void __launch_bounds__(64) k_verify_tabfix(TablesArgs t, unsigned table_blocks, FixedSmallArgs f) {
  if (blockIdx.x < table_blocks) tables_body(t, blockIdx.x);
  else fixed_small_body<C, LPM>(f.table, f.n, f.cap, f.scalars, f.sc_stride, f.out, f.nb, blockIdx.x - table_blocks);
}
size_t verify_wp_scratch_bytes(size_t nb, size_t nvar) {
  size_t lanes = (nvar + TNP - 1) / TNP, blocks = (nb * lanes + 63) / 64;
  return blocks * TNP * SE * STE * 64 * 4 + nb * nvar * SE * sizeof(AffRaw) + nb * 64 * sizeof(JacRaw) + 256;
}
template <int C>
static void launch_tabfix(hipStream_t st, const TablesArgs &t, unsigned tb, const FixedSmallArgs &f) {
  if (f.nb >= 1024) hipLaunchKernelGGL((k_verify_tabfix<C, 16>), dim3(tb + (unsigned)((f.nb + 3) / 4)), dim3(64), 0, st, t, tb, f);
  else hipLaunchKernelGGL((k_verify_tabfix<C, 32>), dim3(tb + (unsigned)((f.nb + 1) / 2)), dim3(64), 0, st, t, tb, f);
}
// The whole MSM + verdict of a batch in three launches (timed separately by the API's profile scopes).
bool verify_wp_supported(size_t nb, size_t nvar, int c, size_t n) {
  const size_t total = (2 + 2 * n) * (252 / c + 1);
  return nb >= 1 && total <= 16384 && nvar && (c == 8 || c == 16 || c == 20);
}
static TablesArgs wp_args(const VerifyWp &v, JacRaw **winsum, unsigned *blocks) {
  TablesArgs t{};
  t.points = v.points_abi; t.nb = v.nb; t.nvar = v.nvar; t.lanes = (v.nvar + TNP - 1) / TNP; t.bad = v.bad;
  const size_t nblk = (v.nb * t.lanes + 63) / 64;
  uint8_t *sp = (uint8_t *)v.scratch;
  t.scratch = (int32_t *)sp; sp += nblk * TNP * SE * STE * 64 * 4;
  t.tab = (AffRaw *)sp; sp += v.nb * v.nvar * SE * sizeof(AffRaw);
  *winsum = (JacRaw *)(((uintptr_t)sp + 63) & ~(uintptr_t)63);
  *blocks = (unsigned)nblk;
  return t;
}
// tables of the proof points | table-lookup MSMs over the generators (fixed scalars as for fixed_msm)
void verify_wp_tabfix(hipStream_t st, const VerifyWp &v, int c, const AffDev *table, size_t n, size_t cap,
                      const uint32_t *fixed_scalars, size_t sc_stride, JacRaw *out_fixed) {
  JacRaw *winsum;
  unsigned blocks;
  TablesArgs t = wp_args(v, &winsum, &blocks);
  FixedSmallArgs f{table, n, cap, fixed_scalars, sc_stride, out_fixed, v.nb};
  if (c == 8) launch_tabfix<8>(st, t, blocks, f);
  else if (c == 16) launch_tabfix<16>(st, t, blocks, f);
  else launch_tabfix<20>(st, t, blocks, f);
}
// var_scalars: nb x nvar x 8 words in operand order
void verify_wp_windows(hipStream_t st, const VerifyWp &v, const uint32_t *var_scalars) {
  JacRaw *winsum;
  unsigned blocks;
  TablesArgs t = wp_args(v, &winsum, &blocks);
  hipLaunchKernelGGL(k_verify_windows, dim3(v.nb), dim3(64), 0, st, t.tab, var_scalars, v.nvar, winsum);
}
void verify_wp_horner(hipStream_t st, const VerifyWp &v, const JacRaw *fixed, int32_t *ok, Words8 *mega) {
  JacRaw *winsum;
  unsigned blocks;
  (void)wp_args(v, &winsum, &blocks);
  static_assert(num_windows<SW>() == 64, "64 window sums per proof");
  static const bool grouped = !(getenv("BPGPU_HORNER_GROUPS") && atoi(getenv("BPGPU_HORNER_GROUPS")) == 0);
  if (grouped) {
    hipLaunchKernelGGL(k_verify_horner_groups, dim3((v.nb * (64 / HG) + 63) / 64), dim3(64), 0, st, winsum, v.nb);
    hipLaunchKernelGGL(k_verify_horner, dim3((v.nb + 63) / 64), dim3(64), 0, st, winsum, fixed, v.nb, ok, mega, 64 / HG, HG, SW * HG);
  } else {
    hipLaunchKernelGGL(k_verify_horner, dim3((v.nb + 63) / 64), dim3(64), 0, st, winsum, fixed, v.nb, ok, mega, 64, 1, SW);
  }
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
