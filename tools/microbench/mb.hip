// Microbenchmark: throughput of the fe29/ec29 primitives on gfx950 (defines the integer-ALU roofline).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../mpc_bulletproof_amd/csrc/ec29.cuh"
using namespace bp;
#define LB_ __launch_bounds__(256)
extern "C" __global__ void LB_ k_fpmul(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fp a = unpack<FP>(io + 16 * t), b = unpack<FP>(io + 16 * t + 8);
  for (int i = 0; i < iters; i++) { a = mul(a, b); b = mul(b, a); }
  pack(io + 16 * t, canon(a)); pack(io + 16 * t + 8, canon(b));
}
extern "C" __global__ void LB_ k_fpsqr(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fp a = unpack<FP>(io + 16 * t), b = unpack<FP>(io + 16 * t + 8);
  for (int i = 0; i < iters; i++) { a = sqr(a); b = sqr(b); }
  pack(io + 16 * t, canon(a)); pack(io + 16 * t + 8, canon(b));
}
extern "C" __global__ void LB_ k_fnmul(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fn a = unpack<FN>(io + 16 * t), b = unpack<FN>(io + 16 * t + 8);
  for (int i = 0; i < iters; i++) { a = mul(a, b); b = mul(b, a); }
  pack(io + 16 * t, canon(a)); pack(io + 16 * t + 8, canon(b));
}
extern "C" __global__ void LB_ k_madd(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Aff q; q.x = unpack<FP>(io + 16 * t); q.y = unpack<FP>(io + 16 * t + 8);
  Jac acc = jac_dbl(jac_from_aff(q));
  for (int i = 0; i < iters; i++) acc = jac_madd(acc, q);
  pack(io + 16 * t, canon(acc.X)); pack(io + 16 * t + 8, canon(acc.Z));
}
extern "C" __global__ void LB_ k_dbl(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Aff q; q.x = unpack<FP>(io + 16 * t); q.y = unpack<FP>(io + 16 * t + 8);
  Jac acc = jac_from_aff(q);
  for (int i = 0; i < iters; i++) acc = jac_dbl(acc);
  pack(io + 16 * t, canon(acc.X)); pack(io + 16 * t + 8, canon(acc.Z));
}
// raw v_mad_u64_u32 issue rate: 8 independent chains per lane
extern "C" __global__ void LB_ k_mad(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t a[8]; uint32_t x = io[t], y = io[t + 1] | 1;
  for (int j = 0; j < 8; j++) a[j] = x + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = (uint64_t)(uint32_t)a[j] * y + a[j];
  }
  uint64_t s = 0; for (int j = 0; j < 8; j++) s ^= a[j];
  io[t] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
extern "C" __global__ void LB_ k_add32(uint32_t* io, int iters) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t a[8]; uint32_t x = io[t], y = io[t + 1] | 1;
  for (int j = 0; j < 8; j++) a[j] = x + j;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = (a[j] ^ y) + (a[j] >> 3);
  }
  uint32_t s = 0; for (int j = 0; j < 8; j++) s ^= a[j];
  io[t] = s;
}
typedef void (*kern_t)(uint32_t*, int);
static double run(const char* name, kern_t k, int blocks, int iters, double ops_per_iter_per_thread, uint32_t* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * iters * ops_per_iter_per_thread;
  printf("%-8s blocks=%5d iters=%6d  %8.3f ms  %10.2f Gop/s\n", name, blocks, iters, ms, ops / ms / 1e6);
  return ops / ms / 1e6;
}
int main() {
  size_t n = 256 * 8 * 256 * 16 + 64;
  std::vector<uint32_t> h(n);
  // valid-ish field elements (< 2^251): random words with top word small; for k_madd use the curve generator
  uint32_t G[16] = {0xc943cfca,0x3d723d8b,0x0d1819e0,0xdeacfd9b,0x5a40f0c7,0x7beced41,0x8599971b,0x01ef15c1,
                    0x36e8dc1f,0x2873000c,0x1abe43a3,0xde53ecd1,0xdf46ec62,0xb7be4801,0x0aa49730,0x00566806};
  for (size_t i = 0; i < n; i++) h[i] = (i % 16 < 16) ? G[i % 16] : 0;
  uint32_t* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int wpc : {1, 2, 4, 8}) {   // blocks of 256 thr = 4 waves = 1 wave/SIMD per block per CU
    int blocks = 256 * wpc;
    printf("--- %d block(s)/CU (%d waves/SIMD)\n", wpc, wpc);
    run("mad64", k_mad, blocks, 20000, 8, d);
    run("add32", k_add32, blocks, 20000, 8 * 3, d);
    run("fpmul", k_fpmul, blocks, 2000, 2, d);
    run("fpsqr", k_fpsqr, blocks, 2000, 2, d);
    run("fnmul", k_fnmul, blocks, 2000, 2, d);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    run("madd", k_madd, blocks, 500, 1, d);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    run("dbl", k_dbl, blocks, 500, 1, d);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  }
  return 0;
}
