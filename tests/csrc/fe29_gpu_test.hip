// GPU twin of fe29_host_test.cpp: the same field / point operations through the DEVICE code path of fe29.cuh / ec29.cuh
// (asm MAD chains, out-of-line exact group-law branches), one element per lane, results as canonical bytes.
// Test infrastructure only -- never linked into the product.  Built by __graft_entry__.build() into
// tests/csrc/libfe29_gpu.so; tests/test_gpu_arith.py compares its outputs with Python big integers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../mpc_bulletproof_amd/csrc/ec29.cuh"
#include "../../mpc_bulletproof_amd/csrc/ec29_quad.cuh"
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
int g29_point(int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, int *rc) {
  DevBuf da(64 * n), db(64 * n), dout(64 * n), drc(4 * n);
  if (!up(da, a, 64 * n) || !up(db, b, 64 * n) || !dout.p || !drc.p) return -100;
  hipLaunchKernelGGL(k_point, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, (const uint32_t *)da.p, (const uint32_t *)db.p,
                     (uint32_t *)dout.p, (int *)drc.p, n);
  if (hipDeviceSynchronize() != hipSuccess) return -100;
  return down(out, dout, 64 * n) && down(rc, drc, 4 * n) ? 0 : -100;
}
}
