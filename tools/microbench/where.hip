// Where do the waves of SMALL concurrent launches land?  S streams each launch a kernel of G workgroups x 64 threads whose waves
// record their hardware position (XCC, SE, CU, SIMD) and then stay resident for `us` microseconds.  Prints, per snapshot, how many
// (XCC, SE, CU, SIMD) slots hold 0, 1, 2, ... of the S x G waves: a balanced placement of 20 x 112 waves over 1 024 SIMDs is 2-3 per
// SIMD; a stacked one leaves SIMDs empty while others hold ten.
// build: hipcc --offload-arch=gfx950 -O2 -o where where.hip      run: ./where [streams] [workgroups] [threads] [us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void k_where(uint32_t *out, long long ticks) {
  const long long t0 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    out[2 * w] = hw;
    out[2 * w + 1] = xcc;
  }
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char **argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 20, G = argc > 2 ? atoi(argv[2]) : 112, T = argc > 3 ? atoi(argv[3]) : 64;
  const int us = argc > 4 ? atoi(argv[4]) : 300;
  const int wpg = T / 64, W = G * wpg;
  std::vector<hipStream_t> st(S);
  std::vector<uint32_t *> d(S);
  for (int i = 0; i < S; i++) {
    if (hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess) return 1;
    if (hipMalloc(&d[i], 8 * W) != hipSuccess) return 1;
  }
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
  const long long ticks = (long long)us * rate / 1000;
  for (int rep = 0; rep < 3; rep++) {
    for (int i = 0; i < S; i++) hipLaunchKernelGGL(k_where, dim3(G), dim3(T), 0, st[i], d[i], ticks);
    hipDeviceSynchronize();
  }
  std::map<uint32_t, int> simd, cu;
  std::vector<uint32_t> h(2 * W);
  std::map<uint32_t, int> per_kernel_cus;
  for (int i = 0; i < S; i++) {
    hipMemcpy(h.data(), d[i], 8 * W, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> mine;
    for (int w = 0; w < W; w++) {
      const uint32_t hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
      const uint32_t simd_id = (hw >> 4) & 3, cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu_id;
      simd[(cukey << 2) | simd_id]++;
      cu[cukey]++;
      mine[cukey]++;
    }
    if (i < 3) {
      printf("launch %d: %zu distinct CUs; first waves at (xcc,se,cu,simd):", i, mine.size());
      for (int w = 0; w < 12 && w < W; w++)
        printf(" (%u,%u,%u,%u)", h[2 * w + 1] & 0xf, (h[2 * w] >> 13) & 7, (h[2 * w] >> 8) & 0xf, (h[2 * w] >> 4) & 3);
      printf("\n");
    }
  }
  std::map<int, int> hs, hc;
  for (auto &kv : simd) hs[kv.second]++;
  for (auto &kv : cu) hc[kv.second]++;
  printf("%d launches x %d workgroups x %d threads (%d waves), resident %d us each\n", S, G, T, S * W, us);
  printf("SIMDs touched: %zu of 1024; waves per touched SIMD -> number of SIMDs:", simd.size());
  for (auto &kv : hs) printf("  %d:%d", kv.first, kv.second);
  printf("\nCUs touched: %zu of 256; waves per touched CU -> number of CUs:", cu.size());
  for (auto &kv : hc) printf("  %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
