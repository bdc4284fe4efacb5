// ec_dev.cuh -- device helpers shared by the elliptic-curve translation units.
#pragma once
#include "ec29.cuh"
#include "kernels.h"

namespace bpk {
using namespace bp;

__device__ __forceinline__ void raw_store(JacRaw *d, const Jac &p) {
#pragma unroll
  for (int j = 0; j < NL; j++) { d->v[j] = p.X.v[j]; d->v[NL + j] = p.Y.v[j]; d->v[2 * NL + j] = p.Z.v[j]; }
}
__device__ __forceinline__ Jac raw_load(const JacRaw *s) {
  Jac p;
#pragma unroll
  for (int j = 0; j < NL; j++) { p.X.v[j] = s->v[j]; p.Y.v[j] = s->v[NL + j]; p.Z.v[j] = s->v[2 * NL + j]; }
  return p;
}
__device__ __forceinline__ Aff aff_load(const AffDev *s) {
  Aff a;
  uint32_t w[16];
#pragma unroll
  for (int j = 0; j < 16; j++) w[j] = s->w[j];
  a.x = unpack<FP>(w);
  a.y = unpack<FP>(w + 8);
  return a;
}
__device__ __forceinline__ void aff_store(AffDev *d, const Aff &a) {
  uint32_t w[16];
  pack(w, canon(a.x));
  pack(w + 8, canon(a.y));
#pragma unroll
  for (int j = 0; j < 16; j++) d->w[j] = w[j];
}

// ------------------------------------------------------------------------------------------------
// block-level point sum: every lane holds `acc`; result valid in lane 0.  smem: 27 * TPB/2 ints.
template <int TPB> __device__ __forceinline__ Jac block_sum(Jac acc, int32_t *smem) {
  const int tid = threadIdx.x;
#pragma unroll 1
  for (int s = TPB / 2; s > 0; s >>= 1) {
    if (tid >= s && tid < 2 * s) {
#pragma unroll
      for (int t = 0; t < NL; t++) {
        smem[t * (TPB / 2) + tid - s] = acc.X.v[t];
        smem[(NL + t) * (TPB / 2) + tid - s] = acc.Y.v[t];
        smem[(2 * NL + t) * (TPB / 2) + tid - s] = acc.Z.v[t];
      }
    }
    __syncthreads();
    if (tid < s) {
      Jac q;
#pragma unroll
      for (int t = 0; t < NL; t++) {
        q.X.v[t] = smem[t * (TPB / 2) + tid];
        q.Y.v[t] = smem[(NL + t) * (TPB / 2) + tid];
        q.Z.v[t] = smem[(2 * NL + t) * (TPB / 2) + tid];
      }
      acc = jac_add(acc, q);
    }
    __syncthreads();
  }
  return acc;
}

}  // namespace bpk
