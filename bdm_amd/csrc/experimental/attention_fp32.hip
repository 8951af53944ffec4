// experimental/attention_fp32.hip (built with `make EXPERIMENTAL=1` only) -- the fp32-input MFMA flash attention kernel, superseded
// by the fp16x3 (attention_h2.hip) and bf16x6 (attention_s3.hip) kernels; reached through bdm_attention_core without a workspace.
#include "../../../include/bdm_hip.h"
#include "../common.h"

using namespace bdm;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// (a) flash-style kernel for the 16^3 = 4096-token voxel attention inside sa_layers.1.0.
//     S^T tile = K^T Q is computed with the KEY on the accumulator rows and the QUERY on the MFMA
//     lane, so (i) the softmax over keys is a register + one-partner-lane reduction and (ii) the
//     probability tile is already the B operand of the P*V product (no LDS round trip): step s of
//     that product consumes accumulator register s, whose key index is
//     4*(lane>>5) + (s&3) + 8*(s>>2); the V operand is read from LDS with the same key order.
//     One wave = 32 queries; 4 waves share the K/V tiles of a 32-key step.
template <int CB>  // channel blocks of 32 (C <= 32*CB)
__global__ __launch_bounds__(256) void attn_flash_kernel(int C, int L, const float *__restrict__ q,
                                                         const float *__restrict__ k, const float *__restrict__ v,
                                                         long long bs, int ld, float *__restrict__ out,
                                                         long long bs_o, int ld_o) {
  constexpr int CP = 32 * CB;
  __shared__ float Ks[CP][32 + 1];  // [c][key]
  __shared__ float Vs[CP][32 + 1];  // [c][key]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int bi = blockIdx.y;
  const int i0 = (blockIdx.x * 4 + wave) * 32;  // this wave's first query
  const float *qb = q + (size_t)bi * bs, *kb = k + (size_t)bi * bs, *vb = v + (size_t)bi * bs;

  // Q as B operand: lane holds q[c = 2s+lh][i0+li], s = 0..CP/2-1
  float qreg[CP / 2];
#pragma unroll
  for (int s = 0; s < CP / 2; ++s) {
    const int c = 2 * s + lh;
    qreg[s] = (c < C && i0 + li < L) ? qb[(size_t)c * ld + i0 + li] : 0.f;
  }
  f32x16 o[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
  float run_max = -INFINITY, run_sum = 0.f;

  for (int j0 = 0; j0 < L; j0 += 32) {
    __syncthreads();
    for (int e = tid; e < CP * 32; e += 256) {
      const int c = e >> 5, j = e & 31;
      const bool ok = c < C && j0 + j < L;
      Ks[c][j] = ok ? kb[(size_t)c * ld + j0 + j] : 0.f;
      Vs[c][j] = ok ? vb[(size_t)c * ld + j0 + j] : 0.f;
    }
    __syncthreads();
    // S^T[j][i] = sum_c k[c][j] q[c][i] : A[row=j][kk=c] = Ks[c][j]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < CP / 2; ++s)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[2 * s + lh][li], qreg[s], st, 0, 0, 0);
    // keys beyond L do not exist
    float tile_max = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (j >= L) st[r] = -INFINITY;
      tile_max = fmaxf(tile_max, st[r]);
    }
    tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 32, 64));
    const float new_max = fmaxf(run_max, tile_max);
    const float corr = expf(run_max - new_max);  // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - new_max);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    run_sum = run_sum * corr + psum;
    run_max = new_max;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[cb][r] *= corr;
      // O^T[c][i] += sum_j v[c][j] P[j][i] : A[row=c][kk] = Vs[c][key(s,lh)], B = st[s]
#pragma unroll
      for (int s = 0; s < 16; ++s)
        o[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[cb * 32 + li][(s & 3) + 8 * (s >> 2) + 4 * lh], st[s], o[cb],
                                                     0, 0, 0);
    }
  }
  const float inv = 1.0f / run_sum;
  float *ob = out + (size_t)bi * bs_o;
  if (i0 + li < L) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (c < C) ob[(size_t)c * ld_o + i0 + li] = o[cb][r] * inv;
      }
  }
}


int attention_flash_fp32(int b, int c, int l, const float *q, const float *k, const float *v, long long bs_qkv, int ld_qkv,
                         float *out, long long bs_o, int ld_o, hipStream_t s) {
  dim3 grid(cdiv(l, 128), b);
  if (c <= 32)
    hipLaunchKernelGGL(attn_flash_kernel<1>, grid, dim3(256), 0, s, c, l, q, k, v, bs_qkv, ld_qkv, out, bs_o, ld_o);
  else
    hipLaunchKernelGGL(attn_flash_kernel<2>, grid, dim3(256), 0, s, c, l, q, k, v, bs_qkv, ld_qkv, out, bs_o, ld_o);
  return launch_status("attention_flash");
}
