// rng_ops.hip -- per-shape counter-based random streams for the sampler (SURVEY.md 8e "Partitioning").
//
// The reference seeds ONE global generator per process with seed + rank (experiments/training_utils.py:373-378) and draws
// the DDPM noise (model/model.py:286), the PVD noise (pvd/__init__.py:213,232), the initial cloud (main_blending.py:228) and
// the blend masks (main_blending.py:330-338) from it, so a shape's sample depends on which rank and which batch slot it
// lands in.  Here every shape owns a Philox4x32-10 stream keyed by (run seed, GLOBAL shape index): the value of
//   draw `d` of purpose `p`, element `e` of shape `s`   =   Philox(key(seed, s), counter = (e / 4, 0, d, p))[e % 4]
// depends on nothing else, so results are identical for any rank count, batch size or batch order, and one launch serves
// the whole batch.  Normals: Box-Muller on 24-bit uniforms, (r0, r1) -> elements 4k, 4k+1 and (r2, r3) -> 4k+2, 4k+3.
// oracle/ref_rng.py restates the generator in numpy (bit-exact integers; normals to float32 libm accuracy).
#include "../../include/bdm_hip.h"
#include "common.h"

#pragma clang fp contract(off)

using namespace bdm;

namespace {

struct U4 { uint32_t v[4]; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  U4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

__device__ __forceinline__ float uniform24(uint32_t r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-08f; }  // (0, 1)

// four standard normals of block `blk` of a shape's draw
__device__ __forceinline__ void normal4(unsigned long long key, unsigned long long blk, uint32_t draw, uint32_t purpose,
                                        float z[4]) {
  const U4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), draw, purpose, (uint32_t)key, (uint32_t)(key >> 32));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = uniform24(r.v[2 * h]), u2 = uniform24(r.v[2 * h + 1]);
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    sincosf(6.283185307179586f * u2, &s, &c);
    z[2 * h] = rad * c;
    z[2 * h + 1] = rad * s;
  }
}

__global__ void philox_normal_kernel(long long per_shape, const unsigned long long *__restrict__ keys, uint32_t draw,
                                     uint32_t purpose, float *__restrict__ out) {
  const int bi = blockIdx.y;
  const unsigned long long key = keys[bi];
  const long long blocks = (per_shape + 3) / 4;
  float *o = out + (size_t)bi * per_shape;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < blocks; q += (long long)gridDim.x * blockDim.x) {
    float z[4];
    normal4(key, (unsigned long long)q, draw, purpose, z);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < per_shape) o[4 * q + k] = z[k];
  }
}

__global__ void philox_bits_kernel(long long per_shape, const unsigned long long *__restrict__ keys, uint32_t draw,
                                   uint32_t purpose, long long *__restrict__ out) {
  const int bi = blockIdx.y;
  const unsigned long long key = keys[bi];
  const long long blocks = (per_shape + 3) / 4;
  long long *o = out + (size_t)bi * per_shape;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < blocks; q += (long long)gridDim.x * blockDim.x) {
    const U4 r = philox4x32_10((uint32_t)q, (uint32_t)((unsigned long long)q >> 32), draw, purpose, (uint32_t)key,
                               (uint32_t)(key >> 32));
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < per_shape) o[4 * q + k] = (long long)(r.v[k] & 1u);
  }
}

// mode 0: DDPM (diffusers) step, coefficients (sqrt_beta_prod, sqrt_alpha_prod, c_x0, c_x, sigma); sigma == 0 -> no draw used
// mode 1: PVD step, coefficients (a, b, c1, c2, sigma)
template <int MODE>
// (x and out may be the SAME buffer: the reverse loops step in place; hence no __restrict__ on the two)
__global__ void step_philox_kernel(long long per_shape, const float *x, const float *__restrict__ eps,
                                   const unsigned long long *__restrict__ keys, uint32_t draw, uint32_t purpose, float k0,
                                   float k1, float k2, float k3, float sigma, float *out) {
  const int bi = blockIdx.y;
  const unsigned long long key = keys[bi];
  const long long blocks = (per_shape + 3) / 4;
  const size_t base = (size_t)bi * per_shape;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < blocks; q += (long long)gridDim.x * blockDim.x) {
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (sigma != 0.f) normal4(key, (unsigned long long)q, draw, purpose, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long e = 4 * q + k;
      if (e >= per_shape) break;
      const float xi = x[base + e];
      float v;
      if (MODE == 0) {
        const float x0 = (xi - k0 * eps[base + e]) / k1;
        v = k2 * x0 + k3 * xi;
        if (sigma != 0.f) v = v + sigma * z[k];
      } else {
        const float x0 = k0 * xi - k1 * eps[base + e];
        const float mean = k2 * x0 + k3 * xi;
        v = mean + sigma * z[k];
      }
      out[base + e] = v;
    }
  }
}

inline dim3 stream_grid(int b, long long per_shape) {
  long long g = ((per_shape + 3) / 4 + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return dim3((unsigned)g, (unsigned)b);
}

}  // namespace

extern "C" int bdm_philox_normal(int b, long long per_shape, const unsigned long long *keys, unsigned int draw,
                                 unsigned int purpose, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && per_shape >= 0 && keys != nullptr && out != nullptr, "philox_normal: bad arguments");
  if (b == 0 || per_shape == 0) return BDM_OK;
  BDM_REQUIRE(b <= 65535, "philox_normal: at most 65535 shapes per launch");
  hipLaunchKernelGGL(philox_normal_kernel, stream_grid(b, per_shape), dim3(256), 0, (hipStream_t)stream, per_shape, keys,
                     draw, purpose, out);
  return launch_status("philox_normal");
}

extern "C" int bdm_philox_bits(int b, long long per_shape, const unsigned long long *keys, unsigned int draw,
                               unsigned int purpose, long long *out, void *stream) {
  BDM_REQUIRE(b >= 0 && per_shape >= 0 && keys != nullptr && out != nullptr, "philox_bits: bad arguments");
  if (b == 0 || per_shape == 0) return BDM_OK;
  BDM_REQUIRE(b <= 65535, "philox_bits: at most 65535 shapes per launch");
  hipLaunchKernelGGL(philox_bits_kernel, stream_grid(b, per_shape), dim3(256), 0, (hipStream_t)stream, per_shape, keys,
                     draw, purpose, out);
  return launch_status("philox_bits");
}

extern "C" int bdm_ddpm_step_philox(int b, long long per_shape, const float *x, const float *eps,
                                    const unsigned long long *keys, unsigned int draw, unsigned int purpose,
                                    float sqrt_beta_prod, float sqrt_alpha_prod, float coef_x0, float coef_x, float sigma,
                                    float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && per_shape >= 0 && keys != nullptr, "ddpm_step_philox: bad arguments");
  if (b == 0 || per_shape == 0) return BDM_OK;
  BDM_REQUIRE(b <= 65535, "ddpm_step_philox: at most 65535 shapes per launch");
  hipLaunchKernelGGL(step_philox_kernel<0>, stream_grid(b, per_shape), dim3(256), 0, (hipStream_t)stream, per_shape, x, eps,
                     keys, draw, purpose, sqrt_beta_prod, sqrt_alpha_prod, coef_x0, coef_x, sigma, out);
  return launch_status("ddpm_step_philox");
}

extern "C" int bdm_pvd_step_philox(int b, long long per_shape, const float *x, const float *eps,
                                   const unsigned long long *keys, unsigned int draw, unsigned int purpose,
                                   float sqrt_recip_abar, float sqrt_recipm1_abar, float coef1, float coef2, float sigma,
                                   float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && per_shape >= 0 && keys != nullptr, "pvd_step_philox: bad arguments");
  if (b == 0 || per_shape == 0) return BDM_OK;
  BDM_REQUIRE(b <= 65535, "pvd_step_philox: at most 65535 shapes per launch");
  hipLaunchKernelGGL(step_philox_kernel<1>, stream_grid(b, per_shape), dim3(256), 0, (hipStream_t)stream, per_shape, x, eps,
                     keys, draw, purpose, sqrt_recip_abar, sqrt_recipm1_abar, coef1, coef2, sigma, out);
  return launch_status("pvd_step_philox");
}
