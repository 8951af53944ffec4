// EXPERIMENTAL family (`make EXPERIMENTAL=1`): built, parity-tested, measured and NOT faster than the pair it replaces:
// gather + GroupNorm-1 + Swish + operand split in one launch, one workgroup per (shape, GroupNorm group) -- 32 us against 14 + 6.7 (FP0) /
// 22 + 6.7 (FP1) for sparse_gather_v4 + to_h2_stats: four dependent trips to the GEMM's output on 128 workgroups lose to two kernels of a
// thousand workgroups each.  Kept as the record of VERDICT r4 next-1 (b'); the default library does not contain it.
#include "../../../include/bdm_hip.h"
#include "../common.h"

using namespace bdm;

// =====================================================================================================================================
// Gather of the sparse first convolution + GroupNorm-1 + Swish + fp16 operand split, one workgroup per (shape, GroupNorm group).
//   y (b, n_max, 27, cout) = the batched GEMM's output (row k = occupied cell k, sparse_conv_h2.hip); occ_index (b, r^3) = row of a
//   cell or -1.  out[v][co] = bias[co] + sum over taps t (ascending) of y[occ_index[v + off(t)]][t][co]  -- the sums of
//   sparse_gather_v4_kernel in its order.  The group's tile (cg channels x r^3 cells) stays in LDS; its statistics are reduced in a
//   fixed order inside the workgroup (per item fp32, across items / lanes / waves fp64: independent of the batch); then every record of
//   8 channels is normalised, Swished, scaled by the power of two act_scale and split into (hi, lo) fp16 in the layout of
//   bdm_group_norm_to_h2 (b, cout/8, 2, r^3) x 8.
// =====================================================================================================================================
namespace {

constexpr int GATHER_T = 1024;   // 16 waves: a (shape, group) tile is 4096 (cell, channel quad) items -- four per thread, i.e. four
                                 // dependent trips to the GEMM's output instead of sixteen (256 threads: 64 us per launch, latency-bound)

__device__ __forceinline__ void split2h(float v, unsigned short &h, unsigned short &l) {   // conv3d_h2.hip's split2
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  h = __builtin_bit_cast(unsigned short, hi);
  l = __builtin_bit_cast(unsigned short, lo);
}

__global__ __launch_bounds__(GATHER_T) void gather_h2_small_kernel(int cout, int r, int n_max, int cg, const float *__restrict__ y,
                                                              const int *__restrict__ occ_index, const float *__restrict__ bias,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                              float act_scale, uint4 *__restrict__ out, unsigned *__restrict__ saturated) {
  extern __shared__ __align__(16) float smem[];
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = cout / cg, bi = blockIdx.x / G, g = blockIdx.x % G, Q = cg >> 2, ldt = r3 + 1;
  float *tile = smem;                                        // [cg][r3 + 1]
  int *oi = reinterpret_cast<int *>(tile + (size_t)cg * ldt);  // [r3]
  __shared__ double s_red[GATHER_T / 64][2];
  __shared__ float s_ab[64][2];
  for (int e = tid; e < r3; e += GATHER_T) oi[e] = occ_index[(size_t)bi * r3 + e];
  __syncthreads();
  const float *yb = y + (size_t)bi * n_max * 27 * cout + g * cg;
  const int q = tid % Q;                                     // the thread's channel quad (GATHER_T % Q == 0: the same for all its items)
  const float4 b4 = bias ? *reinterpret_cast<const float4 *>(bias + g * cg + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  double ds = 0.0, dq = 0.0;
  for (int item = tid; item < r3 * Q; item += GATHER_T) {
    const int v = item / Q;
    const int x = v / r2, yy = (v / r) % r, z = v % r;
    // which taps have an occupied input cell (bit t), then ONLY those row pieces, up to 14 in flight, added in ascending tap order
    // (a trip to the GEMM's output per batch: one for almost every cell -- a cell has 3 - 11 occupied neighbours on these levels)
    unsigned mask = 0u;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int gx = x + t / 9 - 1, gy = yy + (t / 3) % 3 - 1, gz = z + t % 3 - 1;
      const bool in = gx >= 0 && gx < r && gy >= 0 && gy < r && gz >= 0 && gz < r;
      mask |= ((in && oi[in ? (gx * r + gy) * r + gz : 0] >= 0) ? 1u : 0u) << t;
    }
    float4 acc = b4;
    while (mask) {
      float4 vv[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) {
        vv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask) {
          const int t = __ffs((int)mask) - 1;
          mask &= mask - 1;
          const int k = oi[((x + t / 9 - 1) * r + (yy + (t / 3) % 3 - 1)) * r + (z + t % 3 - 1)];
          vv[u] = *reinterpret_cast<const float4 *>(yb + ((size_t)k * 27 + t) * cout + q * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 14; ++u) { acc.x += vv[u].x; acc.y += vv[u].y; acc.z += vv[u].z; acc.w += vv[u].w; }
    }
    float *tp = tile + (size_t)(q * 4) * ldt + v;
    tp[0] = acc.x; tp[ldt] = acc.y; tp[2 * ldt] = acc.z; tp[3 * ldt] = acc.w;
    ds += (double)((acc.x + acc.y) + (acc.z + acc.w));
    dq += (double)((acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
  if (lane == 0) { s_red[wave][0] = ds; s_red[wave][1] = dq; }
  __syncthreads();
  if (tid < cg) {
    double a = 0.0, qq = 0.0;
    for (int w = 0; w < GATHER_T / 64; ++w) { a += s_red[w][0]; qq += s_red[w][1]; }   // waves in order
    const double cnt = (double)cg * r3, mean = a / cnt;
    double var = qq / cnt - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int ch = g * cg + tid;
    const float ga = gamma[ch] * rstd;
    s_ab[tid][0] = ga;
    s_ab[tid][1] = beta[ch] - (float)mean * ga;
  }
  __syncthreads();
  // records: (8 channels) x cell, cells on the lanes
  const int C8 = cout / 8, recs = cg / 8;
  bool sat = false;
  for (int item = tid; item < recs * r3; item += GATHER_T) {
    const int rl = item / r3, v = item - rl * r3;
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cl = rl * 8 + j;
      float t = swishf(tile[(size_t)cl * ldt + v] * s_ab[cl][0] + s_ab[cl][1]) * act_scale;
      sat |= !(fabsf(t) <= 65504.f);
      split2h(t, h[j], l[j]);
    }
    uint4 ph, pl;
    ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
    pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
    uint4 *o = out + ((size_t)bi * C8 + (g * cg) / 8 + rl) * 2 * (size_t)r3;
    o[v] = ph;
    o[(size_t)r3 + v] = pl;
  }
  if (saturated != nullptr && __ballot(sat) != 0ull && lane == __ffsll((long long)__ballot(sat)) - 1) atomicOr(saturated, 1u);
}

}  // namespace

extern "C" int bdm_sparse_conv_gather_h2_small(int b, int cout, int r, int n_max, const float *y, const int *occ_index, const float *bias,
                                               int groups, const float *gamma, const float *beta, float eps, float act_scale, void *out_h2,
                                               unsigned int *saturated, void *stream) {
  const int cg = groups >= 1 && cout % groups == 0 ? cout / groups : 0;
  BDM_REQUIRE(b >= 0 && r >= 1 && n_max >= 1 && y && occ_index && gamma && beta && out_h2 && cg >= 8 && cg % 8 == 0 && cg <= 64 &&
              GATHER_T % (cg / 4) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0,
              "sparse_conv_gather_h2_small: needs 8 | channels per group <= 64, (cg / 4) | 256 (cout=%d groups=%d)", cout, groups);
  {
    int ex = 0;
    BDM_REQUIRE(act_scale > 0.f && act_scale < INFINITY && frexpf(act_scale, &ex) == 0.5f,
                "sparse_conv_gather_h2_small: act_scale must be a power of two (got %g)", (double)act_scale);
  }
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  const size_t smem = sizeof(float) * (size_t)cg * (r3 + 1) + sizeof(int) * (size_t)r3;
  BDM_REQUIRE(smem <= 150 * 1024, "sparse_conv_gather_h2_small: %zu bytes of LDS (r=%d, %d channels per group): not a small grid", smem, r, cg);
  BDM_ALLOW_LDS(gather_h2_small_kernel, smem);
  hipLaunchKernelGGL(gather_h2_small_kernel, dim3(b * groups), dim3(GATHER_T), smem, (hipStream_t)stream, cout, r, n_max, cg, y, occ_index, bias,
                     gamma, beta, eps, act_scale, (uint4 *)out_h2, saturated);
  return launch_status("sparse_conv_gather_h2_small");
}

