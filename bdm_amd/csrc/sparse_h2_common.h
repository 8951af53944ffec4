// sparse_h2_common.h -- fp16x3 helpers shared by the sparse first convolution's GEMM (sparse_conv_h2.hip) and the experimental
// one-kernel form (experimental/sparse_conv_fused.hip): two-term fp16 split of records, the activation scale derived from the
// per-shape maximum, and the per-output-channel weight scale.
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
namespace {

__device__ __forceinline__ void split2s(float v, unsigned short &h, unsigned short &l) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  h = __builtin_bit_cast(unsigned short, hi);
  l = __builtin_bit_cast(unsigned short, lo);
}

__device__ __forceinline__ void split_record(const float4 &p, const float4 &q, float s, f16x8 &hi, f16x8 &lo) {
  const float v[8] = {p.x * s, p.y * s, p.z * s, p.w * s, q.x * s, q.y * s, q.z * s, q.w * s};
  unsigned short h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split2s(v[j], h[j], l[j]);
  uint4 ph, pl;
  ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
  pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
  hi = *reinterpret_cast<const f16x8 *>(&ph);
  lo = *reinterpret_cast<const f16x8 *>(&pl);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (vmcnt(0)), i.e.
// it waits for the global loads issued as PREFETCH for later steps and so exposes their whole latency at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// power of two s with amax * s in [2^14, 2^15)  (1 when amax is 0 / not finite)
__device__ __forceinline__ float act_scale_from_max(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 1.f;
  int ex;
  (void)frexpf(amax, &ex);  // amax = f * 2^ex, f in [0.5, 1)
  return ldexpf(1.f, 15 - ex);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// weights (Cout, Cin, 3,3,3) fp32 -> [G][27][2][Cout] records of 8 fp16 (hi / lo of w * 2^e[co]); inv_scale[co] = 2^-e[co]
// ---------------------------------------------------------------------------------------------------
static __global__ void sparse_fused_weight_scale_kernel(int cout, int cin, const float *__restrict__ w, float *__restrict__ scale,
                                                 float *__restrict__ inv_scale) {
  __shared__ float sh[256];
  const int co = blockIdx.x;
  float m = 0.f;
  for (int e = threadIdx.x; e < cin * 27; e += blockDim.x) m = fmaxf(m, fabsf(w[(size_t)co * cin * 27 + e]));
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int ex = 0;
    const float mx = sh[0];
    const bool ok = mx > 0.f && mx < INFINITY;
    if (ok) (void)frexpf(mx, &ex);
    const int e = ok ? 10 - ex : 0;  // mx * 2^e in [2^9, 2^10)
    scale[co] = ldexpf(1.0f, e);
    inv_scale[co] = ldexpf(1.0f, -e);
  }
}
