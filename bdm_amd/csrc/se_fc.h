// se_fc.h -- the two small FC layers of the SE block (se.py:8-19, reduction 8, ReLU, no biases), shared by every kernel that evaluates
// them (se_fc_kernel / se_gate_fused_kernel / devox_gn_fused_kernel in dense_ops.hip, se_rows_fc_kernel in pvconv_compact.hip,
// pv_tail_small_kernel in pvconv_small.hip) so that the gate has the same bits whichever launch computes it.
//
// Round 5: the hidden layer was `hidden` threads each walking a whole row of W1 (c dependent FMAs on strided loads: 8 - 17 us for 8 K
// MACs, a launch-sized cost at every one of the 14 gates of a forward).  Now a WAVE owns a hidden unit: lanes take k = lane, lane + 64,
// ... (W1's row is read as contiguous 256-byte runs), a fixed xor-butterfly adds the 64 partial sums -- deterministic, batch-independent.
#pragma once
#include <hip/hip_runtime.h>

namespace bdm {

// s_mean[c] (LDS) -> s_hid[hidden] (LDS); every thread of the workgroup calls it (blockDim.x a multiple of 64); the caller barriers
// before (s_mean complete) and after (s_hid complete).  A wave owns hidden units w, w + nw, ...; lane l sums k = l, l + 64, ... and the
// 64 partial sums are added by DPP row operations + four readlanes in a fixed order (xor 1, xor 2, half-row mirror, row mirror, then rows
// 0..3): a `__shfl_xor` butterfly is six DEPENDENT ds_bpermute round trips through the LDS crossbar per unit -- measured 5.3 us for the 32
// units of one gate (tools/tail_bench.py phase stamps), more than the rest of the kernel that needed the gate.
__device__ __forceinline__ float se_wave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return ((r0 + r1) + r2) + r3;   // every lane of a 16-lane row holds the row's sum; rows added in order (wave-uniform result)
}

__device__ __forceinline__ void se_hidden_layer(int c, int hidden, const float *__restrict__ w1, const float *s_mean, float *s_hid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // four units of the wave at a time, and up to four k per lane of each, with all their weight loads in flight (clamped addresses,
  // zeroed beyond the matrix): one unit after the other, each waiting for its own loads, was hidden / nw dependent round trips --
  // 11.6 us for a 32 x 256 layer, slower than the scalar loop it replaced.  Per lane k ascending, then se_wave_sum: same sums.
  for (int j0 = wave; j0 < hidden; j0 += 4 * nw) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = lane; k0 < c; k0 += 4 * 64) {
      float w[4][4], sk[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int k = k0 + 64 * v;
        sk[v] = k < c ? s_mean[k] : 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u][v] = w1[(size_t)min(j0 + u * nw, hidden - 1) * c + min(k, c - 1)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (k0 + 64 * v < c) a[u] += w[u][v] * sk[v];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float r = se_wave_sum(a[u]);
      if (lane == 0 && j0 + u * nw < hidden) s_hid[j0 + u * nw] = fmaxf(r, 0.f);
    }
  }
}

// gate of channel ci from the hidden vector (LDS): sigmoid(W2[ci] . hid), k ascending
__device__ __forceinline__ float se_gate_of(int ci, int hidden, const float *__restrict__ w2, const float *s_hid) {
  float a = 0.f;
  for (int k = 0; k < hidden; ++k) a += w2[(size_t)ci * hidden + k] * s_hid[k];
  return 1.0f / (1.0f + expf(-a));
}

}  // namespace bdm
