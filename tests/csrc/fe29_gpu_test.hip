// GPU twin of fe29_host_test.cpp: the same field / point operations through the DEVICE code path of fe29.cuh / ec29.cuh
// (asm MAD chains, out-of-line exact group-law branches), one element per lane, results as canonical bytes.
// Test infrastructure only -- never linked into the product.  Built by __graft_entry__.build() into
// tests/csrc/libfe29_gpu.so; tests/test_gpu_arith.py compares its outputs with Python big integers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../mpc_bulletproof_amd/csrc/ec29.cuh"
#include "../../mpc_bulletproof_amd/csrc/ec29_quad.cuh"
#include "../../mpc_bulletproof_amd/csrc/ec29_row.cuh"
using namespace bp;

template <class F> __device__ bool d_load(Fe<F> &out, const uint32_t *w) {
  uint32_t t[8];
  for (int j = 0; j < 8; j++) t[j] = w[j];
  if (!words_lt_mod<F>(t)) return false;
  out = to_mont(unpack<F>(t));
  return true;
}
template <class F> __device__ void d_store(uint32_t *w, const Fe<F> &x) {
  uint32_t t[8];
  pack(t, from_mont(x));
  for (int j = 0; j < 8; j++) w[j] = t[j];
}
template <class F> __global__ void k_field(int op, const uint32_t *a, const uint32_t *b, uint32_t *out, int *rc, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> x, y, r;
  if (!d_load(x, a + 8 * i) || !d_load(y, b + 8 * i)) { rc[i] = -1; return; }
  switch (op) {   // same numbering as fe29_host_test.cpp
    case 0: r = add(x, y); break;
    case 1: r = sub(x, y); break;
    case 2: r = mul(x, y); break;
    case 3: r = sqr(x); break;
    case 4: r = inv(x); break;
    case 5: r = neg(x); break;
    case 6: r = mul_small<8>(x); break;
    case 7: r = mul(sub(x, y), add(x, y)); break;
    case 8: r = sqr(sub(sub(x, y), y)); break;
    case 9: r = mul(norm(add_nr(add_nr(x, x), x)), sub(y, x)); break;
    case 10: r = inv_gcd(x); break;
    default: rc[i] = -2; return;
  }
  d_store(out + 8 * i, r);
  rc[i] = 0;
}
// products of RAW limb vectors (9 x int32 each, any representation the multiplication promises to accept):
// out = canonical bytes of the integer  value(a) * value(b) / 2^261  mod m
template <class F> __global__ void k_rawmul(int sq, const int32_t *a, const int32_t *b, uint32_t *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> x, y;
  for (int j = 0; j < NL; j++) { x.v[j] = a[NL * i + j]; y.v[j] = b[NL * i + j]; }
  Fe<F> r = sq ? sqr(x) : mul(x, y);
  uint32_t t[8];
  pack(t, canon(r));
  for (int j = 0; j < 8; j++) out[8 * i + j] = t[j];
}
// op 0: madd(a, b)  1: add(jac a, jac b) with both operands rescaled to non-trivial Z  2: dbl(a)   (as h29_point)
// op 3: ((identity + a) + b) through the extended-Jacobian accumulator (xyzz_madd), 4: (a with a non-trivial ZZ / ZZZ) + b by
// xyzz_madd_nzq -- the additions of the fixed-base walks and window sums, incl. their out-of-line exact cases
__global__ void k_point(int op, const uint32_t *a, const uint32_t *b, uint32_t *out, int *rc, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wa[16], wb[16];
  for (int j = 0; j < 16; j++) { wa[j] = a[16 * i + j]; wb[j] = b[16 * i + j]; }
  Aff p, q;
  if (!aff_from_boundary(p, wa) || !aff_from_boundary(q, wb)) { rc[i] = -1; return; }
  Jac pj = jac_from_aff(p), r;
  if (op == 0) r = jac_madd(pj, q);
  else if (op == 1) {
    Jac qj = jac_from_aff(q);
    const uint32_t zw[8] = {0x12345, 7, 9, 0, 0, 0, 0, 0};
    Fp z = to_mont(unpack<FP>(zw));
    Fp z2 = sqr(z), z3 = mul(z2, z);
    if (!jac_is_inf(qj)) { qj.X = mul(qj.X, z2); qj.Y = mul(qj.Y, z3); qj.Z = mul(qj.Z, z); }
    if (!jac_is_inf(pj)) { Fp w = add(z, z2), w2 = sqr(w), w3 = mul(w2, w); pj.X = mul(pj.X, w2); pj.Y = mul(pj.Y, w3); pj.Z = mul(pj.Z, w); }
    r = jac_add(pj, qj);
  } else if (op == 3) {
    r = xyzz_to_jac(xyzz_madd(xyzz_madd(xyzz_inf(), p), q));
  } else if (op == 4) {
    if (aff_is_inf(q)) { rc[i] = -3; return; }
    Jac t = pj;
    const uint32_t zw[8] = {0x54321, 11, 5, 0, 0, 0, 0, 0};
    Fp z = to_mont(unpack<FP>(zw));
    if (!jac_is_inf(t)) { Fp z2 = sqr(z), z3 = mul(z2, z); t.X = mul(t.X, z2); t.Y = mul(t.Y, z3); t.Z = mul(t.Z, z); }
    r = xyzz_to_jac(xyzz_madd_nzq(xyzz_from_jac(t), q));
  } else r = jac_dbl(pj);
  uint32_t wo[16];
  aff_to_boundary(wo, jac_to_aff(r));
  for (int j = 0; j < 16; j++) out[16 * i + j] = wo[j];
  rc[i] = 0;
}

// Quad-cooperative group law (ec29_quad.cuh): element i is handled by the 4 lanes of quad i.
// op 0: a + b (both operands rescaled to non-trivial Z)   1: 2 a   2: ((a 2^32 + b) 2^32 + b) 2^4 + a  (a Horner stretch)
__global__ void __launch_bounds__(64) k_q4(int op, const uint32_t *a, const uint32_t *b, uint32_t *out, int *rc, size_t n) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t i = t >> 2;
  const int role = (int)(t & 3);
  if (i >= n) i = n - 1;          // whole quads stay active (n is padded to a multiple of 16 quads by the caller anyway)
  uint32_t wa[16], wb[16];
  for (int j = 0; j < 16; j++) { wa[j] = a[16 * i + j]; wb[j] = b[16 * i + j]; }
  Aff p, q;
  if (!aff_from_boundary(p, wa) || !aff_from_boundary(q, wb)) { if (role == 0) rc[i] = -1; return; }
  Jac pj = jac_from_aff(p), qj = jac_from_aff(q);
  const uint32_t zw[8] = {0x12345, 7, 9, 0, 0, 0, 0, 0};
  Fp z = to_mont(unpack<FP>(zw));
  Fp z2 = sqr(z), z3 = mul(z2, z);
  if (!jac_is_inf(qj)) { qj.X = mul(qj.X, z2); qj.Y = mul(qj.Y, z3); qj.Z = mul(qj.Z, z); }
  if (!jac_is_inf(pj)) { Fp w = add(z, z2), w2 = sqr(w), w3 = mul(w2, w); pj.X = mul(pj.X, w2); pj.Y = mul(pj.Y, w3); pj.Z = mul(pj.Z, w); }
  JacT P = jact_from_jac(pj), Q = jact_from_jac(qj), r;
  if (op == 0) r = q4_add(P, Q, role);
  else if (op == 1) r = q4_dbl(P, role);
  else {
    r = P;
    for (int rep = 0; rep < 2; rep++) {
      for (int d = 0; d < 32; d++) r = q4_dbl(r, role);
      r = q4_add(r, Q, role);
    }
    for (int d = 0; d < 4; d++) r = q4_dbl(r, role);
    r = q4_add(r, P, role);
  }
  // Th must be -Z^4 / 2 on exit (it feeds the next doubling)
  Jac rj = jact_to_jac(r);
  const bool t_ok = is_zero_exact(add(add_nr(r.Th, r.Th), sqr(sqr(r.Z))));
  uint32_t wo[16];
  aff_to_boundary(wo, jac_to_aff(jac_is_inf(rj) || is_zero_exact(rj.Z) ? jac_inf() : rj));
  if (role == 0) {
    for (int j = 0; j < 16; j++) out[16 * i + j] = wo[j];
    rc[i] = t_ok ? 0 : -3;
  }
}

// ---- row-distributed arithmetic (ec29_row.cuh): one element per 16-lane row, one point per wave
// products of RAW limb vectors: element i lives in row (i & 3) of wave (i >> 2)
__global__ void __launch_bounds__(64) k_rowmul(const int32_t *a, const int32_t *b, uint32_t *out, size_t n) {
  const RowK K = rowk_init();
  size_t i = (size_t)blockIdx.x * 4 + K.row;
  const bool live = i < n;
  if (!live) i = n - 1;
  const Rfe x = rload(K, a + NL * i), y = rload(K, b + NL * i);
  const Rfe r = rmul(K, x, y);
  const int32_t hi_lanes = r;                      // lanes 9..15 must read 0 (element invariant)
  uint32_t t[8];
  pack(t, canon(rgather(r)));
  if (live && K.lane == 0) for (int j = 0; j < 8; j++) out[8 * i + j] = t[j];
  if (live && K.lane >= NL && hi_lanes != 0) out[8 * i] = 0xDEADBEEFu;
}
// rbc4 / rgather / rscatter / rhalf_nr / rnorm probes: out[64 * 8] ints per wave
__global__ void __launch_bounds__(64) k_rowprobe(const int32_t *a, int32_t *out) {
  const RowK K = rowk_init();
  const int t = threadIdx.x;
  const R4 b = rbc4(t);
  out[t] = b.r0; out[64 + t] = b.r1; out[128 + t] = b.r2; out[192 + t] = b.r3;
  const Rfe x = rload(K, a + NL * K.row);
  const Fp g = rgather(x);
  int ok = 1;
  for (int j = 0; j < NL; j++) ok &= g.v[j] == a[NL * K.row + j];
  ok &= rscatter(K, g) == x;
  out[256 + t] = ok;
  out[320 + t] = rhalf_nr(K, x);
  out[384 + t] = rnorm(K, x);
}
// op 0: a + b (both operands rescaled to non-trivial Z)   1: 2 a   2: ((a 2^32 + b) 2^32 + b) 2^4 + a  (a Horner stretch)   3: 32 b + a from the identity, Th-less addends
__global__ void __launch_bounds__(64) k_rowpoint(int op, const uint32_t *a, const uint32_t *b, uint32_t *out, int *rc, size_t n) {
  const RowK K = rowk_init();
  const size_t i = blockIdx.x;
  uint32_t wa[16], wb[16];
  for (int j = 0; j < 16; j++) { wa[j] = a[16 * i + j]; wb[j] = b[16 * i + j]; }
  Aff p, q;
  if (!aff_from_boundary(p, wa) || !aff_from_boundary(q, wb)) { if (threadIdx.x == 0) rc[i] = -1; return; }
  Jac pj = jac_from_aff(p), qj = jac_from_aff(q);
  const uint32_t zw[8] = {0x12345, 7, 9, 0, 0, 0, 0, 0};
  Fp z = to_mont(unpack<FP>(zw));
  Fp z2 = sqr(z), z3 = mul(z2, z);
  if (!jac_is_inf(qj)) { qj.X = mul(qj.X, z2); qj.Y = mul(qj.Y, z3); qj.Z = mul(qj.Z, z); }
  if (!jac_is_inf(pj)) { Fp w = add(z, z2), w2 = sqr(w), w3 = mul(w2, w); pj.X = mul(pj.X, w2); pj.Y = mul(pj.Y, w3); pj.Z = mul(pj.Z, w); }
  const JacR P = jacr_scatter(K, pj), Q = jacr_scatter(K, qj);
  JacR r;
  if (op == 0) r = radd(K, P, Q);
  else if (op == 1) r = rdbl(K, P);
  else if (op == 3) {          // a Horner pass whose top windows are empty: identity + addends that carry no Th (as loaded from a JacRaw)
    JacR qa = Q, pa = P;
    qa.Th = 0; pa.Th = 0;
    r = jacr_inf(K);
#pragma unroll 1
    for (int d = 0; d < 3; d++) r = rdbl(K, r);
    r = radd(K, r, qa);
#pragma unroll 1
    for (int d = 0; d < 5; d++) r = rdbl(K, r);
    r = radd(K, r, pa);
  } else {
    r = P;
    for (int rep = 0; rep < 2; rep++) {
#pragma unroll 1
      for (int d = 0; d < 32; d++) r = rdbl(K, r);
      r = radd(K, r, Q);
    }
#pragma unroll 1
    for (int d = 0; d < 4; d++) r = rdbl(K, r);
    r = radd(K, r, P);
  }
  const Jac rj = jacr_gather(K, r);
  const Fp th = rgather(r.Th);
  const bool t_ok = is_zero_exact(add(add_nr(th, th), sqr(sqr(rj.Z))));
  // the four rows must agree (the state is replicated) and lanes 9..15 must hold 0
  const R4 xb = rbc4(r.X), yb = rbc4(r.Y), zb = rbc4(r.Z), tb = rbc4(r.Th);
  bool rep_ok = xb.r0 == xb.r1 && xb.r0 == xb.r2 && xb.r0 == xb.r3 && yb.r0 == yb.r1 && yb.r0 == yb.r2 && yb.r0 == yb.r3 &&
                zb.r0 == zb.r1 && zb.r0 == zb.r2 && zb.r0 == zb.r3 && tb.r0 == tb.r1 && tb.r0 == tb.r2 && tb.r0 == tb.r3;
  if (K.lane >= NL) rep_ok = rep_ok && r.X == 0 && r.Y == 0 && r.Z == 0 && r.Th == 0;
  const bool all_rep = __ballot(!rep_ok) == 0;
  uint32_t wo[16];
  aff_to_boundary(wo, jac_to_aff(jac_is_inf(rj) || is_zero_exact(rj.Z) ? jac_inf() : rj));
  if (threadIdx.x == 0) {
    for (int j = 0; j < 16; j++) out[16 * i + j] = wo[j];
    rc[i] = !t_ok ? -3 : (!all_rep ? -4 : 0);
  }
}

// latency probe: ONE wave runs nd dependent doublings (+ one addition per 32) from the generator, in the quad form (form 0: 16
// quads, quad 0 is checked) or the row form (form 1); the result is written so that the chain cannot be dropped
__global__ void __launch_bounds__(64) k_chain(int form, int nd, const uint32_t *a, uint32_t *out) {
  uint32_t wa[16];
  for (int j = 0; j < 16; j++) wa[j] = a[j];
  Aff p;
  if (!aff_from_boundary(p, wa)) return;
  const Jac pj = jac_from_aff(p);
  Jac rj;
  if (form == 0) {
    const int role = threadIdx.x & 3;
    const JacT P = jact_from_jac(pj);
    JacT r = P;
#pragma unroll 1
    for (int d = 0; d < nd; d++) { r = q4_dbl(r, role); if ((d & 31) == 31) r = q4_add(r, P, role); }
    rj = jact_to_jac(r);
  } else {
    const RowK K = rowk_init();
    const JacR P = jacr_scatter(K, pj);
    JacR r = P;
#pragma unroll 1
    for (int d = 0; d < nd; d++) { r = rdbl(K, r); if ((d & 31) == 31) r = radd(K, r, P); }
    rj = jacr_gather(K, r);
  }
  uint32_t wo[16];
  aff_to_boundary(wo, jac_to_aff(rj));
  if (threadIdx.x == 0) for (int j = 0; j < 16; j++) out[j] = wo[j];
}

namespace {
struct DevBuf {
  void *p = nullptr;
  explicit DevBuf(size_t bytes) { if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) p = nullptr; }
  ~DevBuf() { if (p) (void)hipFree(p); }
};
bool up(DevBuf &d, const void *h, size_t bytes) { return d.p && hipMemcpy(d.p, h, bytes, hipMemcpyHostToDevice) == hipSuccess; }
bool down(void *h, const DevBuf &d, size_t bytes) { return d.p && hipMemcpy(h, d.p, bytes, hipMemcpyDeviceToHost) == hipSuccess; }
}  // namespace

extern "C" {
// all: 0 = ran (per-element status in rc where present), -100 = no HIP device / runtime error
int g29_field(int field, int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, int *rc) {
  DevBuf da(32 * n), db(32 * n), dout(32 * n), drc(4 * n);
  if (!up(da, a, 32 * n) || !up(db, b, 32 * n) || !dout.p || !drc.p) return -100;
  dim3 g((unsigned)((n + 63) / 64)), t(64);
  if (field == 0) hipLaunchKernelGGL(k_field<FP>, g, t, 0, 0, op, (const uint32_t *)da.p, (const uint32_t *)db.p, (uint32_t *)dout.p, (int *)drc.p, n);
  else hipLaunchKernelGGL(k_field<FN>, g, t, 0, 0, op, (const uint32_t *)da.p, (const uint32_t *)db.p, (uint32_t *)dout.p, (int *)drc.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 32 * n) && down(rc, drc, 4 * n) ? 0 : -100;
}
int g29_rawmul(int field, int sq, const int32_t *a, const int32_t *b, size_t n, uint8_t *out) {
  DevBuf da(36 * n), db(36 * n), dout(32 * n);
  if (!up(da, a, 36 * n) || !up(db, b, 36 * n) || !dout.p) return -100;
  dim3 g((unsigned)((n + 63) / 64)), t(64);
  if (field == 0) hipLaunchKernelGGL(k_rawmul<FP>, g, t, 0, 0, sq, (const int32_t *)da.p, (const int32_t *)db.p, (uint32_t *)dout.p, n);
  else hipLaunchKernelGGL(k_rawmul<FN>, g, t, 0, 0, sq, (const int32_t *)da.p, (const int32_t *)db.p, (uint32_t *)dout.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 32 * n) ? 0 : -100;
}
int g29_q4(int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, int *rc) {
  DevBuf da(64 * n), db(64 * n), dout(64 * n), drc(4 * n);
  if (!up(da, a, 64 * n) || !up(db, b, 64 * n) || !dout.p || !drc.p) return -100;
  hipLaunchKernelGGL(k_q4, dim3((unsigned)((4 * n + 63) / 64)), dim3(64), 0, 0, op, (const uint32_t *)da.p, (const uint32_t *)db.p,
                     (uint32_t *)dout.p, (int *)drc.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 64 * n) && down(rc, drc, 4 * n) ? 0 : -100;
}
int g29_rowmul(const int32_t *a, const int32_t *b, size_t n, uint8_t *out) {
  DevBuf da(36 * n), db(36 * n), dout(32 * n);
  if (!up(da, a, 36 * n) || !up(db, b, 36 * n) || !dout.p) return -100;
  hipLaunchKernelGGL(k_rowmul, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, 0, (const int32_t *)da.p, (const int32_t *)db.p, (uint32_t *)dout.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 32 * n) ? 0 : -100;
}
int g29_rowprobe(const int32_t *a /* 4 x 9 limbs */, int32_t *out /* 448 */) {
  DevBuf da(36 * 4), dout(4 * 448);
  if (!up(da, a, 36 * 4) || !dout.p) return -100;
  hipLaunchKernelGGL(k_rowprobe, dim3(1), dim3(64), 0, 0, (const int32_t *)da.p, (int32_t *)dout.p);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 4 * 448) ? 0 : -100;
}
int g29_rowpoint(int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, int *rc) {
  DevBuf da(64 * n), db(64 * n), dout(64 * n), drc(4 * n);
  if (!up(da, a, 64 * n) || !up(db, b, 64 * n) || !dout.p || !drc.p) return -100;
  hipLaunchKernelGGL(k_rowpoint, dim3((unsigned)n), dim3(64), 0, 0, op, (const uint32_t *)da.p, (const uint32_t *)db.p,
                     (uint32_t *)dout.p, (int *)drc.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 64 * n) && down(rc, drc, 4 * n) ? 0 : -100;
}
// nd doublings (+ nd / 32 additions) on one wave; returns the kernel's duration in microseconds (HIP events, best of 5), < 0 on error
double g29_chain(int form, int nd, const uint8_t *a, uint8_t *out) {
  DevBuf da(64), dout(64);
  if (!up(da, a, 64) || !dout.p) return -100;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -100;
  float best = 1e30f;
  for (int rep = 0; rep < 6; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, form, nd, (const uint32_t *)da.p, (uint32_t *)dout.p);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -100;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return down(out, dout, 64) ? (double)best * 1e3 : -100;
}
int g29_point(int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, int *rc) {
  DevBuf da(64 * n), db(64 * n), dout(64 * n), drc(4 * n);
  if (!up(da, a, 64 * n) || !up(db, b, 64 * n) || !dout.p || !drc.p) return -100;
  hipLaunchKernelGGL(k_point, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, (const uint32_t *)da.p, (const uint32_t *)db.p,
                     (uint32_t *)dout.p, (int *)drc.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 64 * n) && down(rc, drc, 4 * n) ? 0 : -100;
}
}
