// k_ark.hip -- arkworks in-memory forms <-> this library's formats, on the device (last paragraph).
//
// The reference's arithmetic layer (mpc-stark 0.2 over ark-ff / ark-ec 0.4) keeps a Scalar as ark_ff::Fp256<MontBackend<_, 4>>
// -- four little-endian u64 limbs holding x * 2^256 mod n -- and a StarkPoint as ark_ec::short_weierstrass::Projective --
// Jacobian (X : Y : Z), the point is (X / Z^2, Y / Z^3), each coordinate four u64 limbs holding c * 2^256 mod p, identity
// Z = 0.  A Rust caller can hand those bytes over as they lie in memory ([u64; 4] = 32 little-endian bytes): one Montgomery
// multiplication per element here replaces a de-Montgomery + big-endian serialisation per scalar and an inversion per point
// on the host.  [The layouts are the published ones of ark-ff / ark-ec 0.4; confirm against the pinned crate versions when a
// Rust toolchain is available: shim/examples/gen_fixtures.rs.]
#include "ec_dev.cuh"

using namespace bp;

namespace bpk {

template <class F> __device__ __forceinline__ Fe<F> limbs_const(const int32_t (&c)[NL]) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < NL; j++) r.v[j] = c[j];
  return r;
}
__device__ __forceinline__ Fn fn_ark_in() { constexpr int32_t C[NL] = FN_ARK_IN; return limbs_const<FN>(C); }
__device__ __forceinline__ Fn fn_ark_out() { constexpr int32_t C[NL] = FN_ARK_OUT; return limbs_const<FN>(C); }
__device__ __forceinline__ Fp fp_ark_in() { constexpr int32_t C[NL] = FP_ARK_IN; return limbs_const<FP>(C); }
__device__ __forceinline__ Fp fp_ark_out() { constexpr int32_t C[NL] = FP_ARK_OUT; return limbs_const<FP>(C); }

// ark Montgomery limbs (x 2^256 mod n) -> plain canonical words
__global__ void __launch_bounds__(256) k_scalars_from_ark(const Words8 *in, Words8 *out, size_t n, int *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = in[i].w[j];
  if (!words_lt_mod<FN>(w)) { atomicOr(bad, 1); for (int j = 0; j < 8; j++) out[i].w[j] = 0; return; }
  pack(w, canon(mul(unpack<FN>(w), fn_ark_in())));
#pragma unroll
  for (int j = 0; j < 8; j++) out[i].w[j] = w[j];
}
// plain canonical words -> ark Montgomery limbs
__global__ void __launch_bounds__(256) k_scalars_to_ark(const Words8 *in, Words8 *out, size_t n, int *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 8; j++) w[j] = in[i].w[j];
  if (!words_lt_mod<FN>(w)) { atomicOr(bad, 1); for (int j = 0; j < 8; j++) out[i].w[j] = 0; return; }
  pack(w, canon(mul(unpack<FN>(w), fn_ark_out())));
#pragma unroll
  for (int j = 0; j < 8; j++) out[i].w[j] = w[j];
}
// ark Jacobian Montgomery (3 x 32 B per point) -> JacRaw (this library's Montgomery limbs), validated: canonical
// coordinates and Y^2 = X^3 + X Z^4 + b Z^6; Z = 0 -> identity
__global__ void __launch_bounds__(128) k_points_from_ark(const Words8 *in, JacRaw *out, size_t n, int *bad) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[8], wy[8], wz[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { wx[j] = in[3 * i].w[j]; wy[j] = in[3 * i + 1].w[j]; wz[j] = in[3 * i + 2].w[j]; }
  Jac p = jac_inf();
  bool ok = words_lt_mod<FP>(wx) && words_lt_mod<FP>(wy) && words_lt_mod<FP>(wz);
  uint32_t zany = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) zany |= wz[j];
  if (ok && zany) {
    const Fp k = fp_ark_in();
    p.X = mul(unpack<FP>(wx), k); p.Y = mul(unpack<FP>(wy), k); p.Z = mul(unpack<FP>(wz), k);
    Fp B;
    constexpr int32_t CB[NL] = CURVE_B_MONT;
    for (int j = 0; j < NL; j++) B.v[j] = CB[j];
    const Fp Z2 = sqr(p.Z), Z4 = sqr(Z2), Z6 = mul(Z4, Z2);
    const Fp rhs = norm(add_nr(add_nr(mul(sqr(p.X), p.X), mul(p.X, Z4)), mul(B, Z6)));
    ok = is_zero_exact(sub(sqr(p.Y), rhs));
    if (!ok) p = jac_inf();
  }
  if (!ok) atomicOr(bad, 1);
  raw_store(&out[i], p);
}
// JacRaw -> ark Jacobian Montgomery; the identity goes out as (1 : 1 : 0) like Projective::zero()
__global__ void __launch_bounds__(128) k_points_to_ark(const JacRaw *in, Words8 *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Jac p = raw_load(&in[i]);
  if (jac_is_inf(p) || is_zero_exact(p.Z)) { p.X = fe_one<FP>(); p.Y = fe_one<FP>(); p.Z = fe_zero<FP>(); }
  const Fp k = fp_ark_out();
  uint32_t w[8];
  pack(w, canon(mul(p.X, k)));
#pragma unroll
  for (int j = 0; j < 8; j++) out[3 * i].w[j] = w[j];
  pack(w, canon(mul(p.Y, k)));
#pragma unroll
  for (int j = 0; j < 8; j++) out[3 * i + 1].w[j] = w[j];
  pack(w, canon(mul(p.Z, k)));
#pragma unroll
  for (int j = 0; j < 8; j++) out[3 * i + 2].w[j] = w[j];
}
// JacRaw -> boundary affine bytes happens through batch_normalize + aff_to_boundary elsewhere; AffDev -> JacRaw:
__global__ void __launch_bounds__(256) k_aff_to_jacraw(const AffDev *in, JacRaw *out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  raw_store(&out[i], jac_from_aff(aff_load(&in[i])));
}

void scalars_from_ark(hipStream_t st, const Words8 *in, Words8 *out, size_t n, int *bad) {
  if (n) hipLaunchKernelGGL(k_scalars_from_ark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n, bad);
}
void scalars_to_ark(hipStream_t st, const Words8 *in, Words8 *out, size_t n, int *bad) {
  if (n) hipLaunchKernelGGL(k_scalars_to_ark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n, bad);
}
void points_from_ark(hipStream_t st, const Words8 *in, JacRaw *out, size_t n, int *bad) {
  if (n) hipLaunchKernelGGL(k_points_from_ark, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, in, out, n, bad);
}
void points_to_ark(hipStream_t st, const JacRaw *in, Words8 *out, size_t n) {
  if (n) hipLaunchKernelGGL(k_points_to_ark, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, in, out, n);
}
void aff_to_jacraw(hipStream_t st, const AffDev *in, JacRaw *out, size_t n) {
  if (n) hipLaunchKernelGGL(k_aff_to_jacraw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, n);
}

}  // namespace bpk
