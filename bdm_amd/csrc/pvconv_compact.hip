// pvconv_compact.hip -- the voxel branch of a PVConv (pvconv.py:74-97) WITHOUT dense grids (round 4).
//
// The input of the first convolution is non-zero on the occupied cells only, so its output differs from its bias only on the
// once-dilated set D1 (23 % of a 32^3 grid for a Gaussian-like cloud of 4096 points); after GroupNorm + Swish the field is a
// per-channel CONSTANT outside D1, so the second convolution's output differs from one of 27 per-class constants (class = which
// faces of the grid a voxel touches: the zero padding removes taps there) only on the twice-dilated set D2 (38 %); and the
// devoxelisation only ever reads cells next to an occupied one (inside D1).  With the two voxel lists of the plan
// (bdm_voxel_dilate, bdm_voxel_dilate_again: sparse_conv_os.hip) the branch becomes
//   first convolution   rows of D1                       bdm_sparse_conv_dil(_gn), compact = 1          (sparse_conv_os.hip)
//   GroupNorm-1 + Swish + operand split: rows of D1 -> (hi, lo) fp16 rows of D1 + the constant record    bdm_group_norm_to_h2_rows
//   second convolution  rows of D2 from rows of D1       bdm_sparse_conv_dil_h2_gn                      (sparse_conv_os.hip)
//   the 27 class constants and their share of the GroupNorm-2 statistics                                 bdm_conv3d_class_constants
//   SE gate: channel means of Swish(GroupNorm-2(.)) over ALL voxels = rows of D2 + counts x constants    bdm_se_gate_gn_rows(_pf)
//   devoxelisation + gate + point branch: 8 corner rows through D2's index                               bdm_devoxelize_gn_gate_add_rows(_pf)
// Nothing of size r^3 x C is written or read any more (the dense second convolution alone wrote 134 MB and took 338 us at
// 64 channels, 32^3, B = 16).  Same arithmetic as the dense path per voxel (fp16x3 products, fp32 accumulation; statistics in
// fp64 over fixed slices); the class constants are exact fp64 dot products rounded once.  Deterministic.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"
#include "se_fc.h"

using namespace bdm;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

namespace {

__device__ __forceinline__ void split2c(float v, unsigned short &h, unsigned short &l) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  h = __builtin_bit_cast(unsigned short, hi);
  l = __builtin_bit_cast(unsigned short, lo);
}

__device__ __forceinline__ void split_rec(const float v[8], uint4 &ph, uint4 &pl) {
  unsigned short h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split2c(v[j], h[j], l[j]);
  ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
  pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
}

// mean / rstd of ALL G <= 64 groups of shape `bi` at once: 32 lanes per group add slices l, l + 32, ... and butterfly (a fixed order);
// 256 threads take 8 groups per round.  Results in s_mean / s_rstd (valid after the call's trailing barrier).
__device__ __forceinline__ void all_group_stats(const double *__restrict__ partial, int bi, int G, int S, double count, float eps,
                                                float *s_mean, float *s_rstd) {
  const int tid = threadIdx.x, l = tid & 31, gq = tid >> 5, per = blockDim.x >> 5;
  for (int g0 = 0; g0 < G; g0 += per) {
    const int g = g0 + gq;
    double a = 0.0, q = 0.0;
    if (g < G) {
      const double *pp = partial + ((size_t)bi * G + g) * S * 2;
      for (int sl = l; sl < S; sl += 32) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    }
    a = half32_sum(a); q = half32_sum(q);   // (DPP row sums + readlanes: bit-identical to the xor butterfly over the group's 32 lanes)
    if (l == 0 && g < G) {
      const double mu = a / count;
      double var = q / count - mu * mu;
      if (var < 0) var = 0;
      s_mean[g] = (float)mu;
      s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
    }
  }
  __syncthreads();
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm-1 + Swish + fp16 split on the rows of D1
// ---------------------------------------------------------------------------------------------------------------------
// x: compact rows (b, n_rows_max, C) [dense_in = 0] or the dense grid (b, C, V) read at the list's voxels [dense_in = 1: the hoisted
// first convolution of SA0.0 still gathers into a grid].  out rows_h2 (b, C8, 2, n_rows_max) records of 8 fp16; block x = 0 also
// leaves the chunk's constant record (b, C8, 2) and const_f32 (b, C): the value the second convolution sees outside D1.
__global__ __launch_bounds__(256) void to_h2_rows_kernel(int C, int V, int G, int S, int n_rows_max, int tiles_max, int dense_in,
                                                         const float *__restrict__ x, const int *__restrict__ dil_list,
                                                         const int *__restrict__ tile_start, const float *__restrict__ bias,
                                                         const double *__restrict__ partial, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps, int act, float act_scale,
                                                         uint4 *__restrict__ rows_h2, uint4 *__restrict__ const_h2,
                                                         float *__restrict__ const_f32, unsigned *__restrict__ saturated) {
  // grid (row blocks of RB rows, shapes); a thread = (row, 8-channel chunk) with the chunk on the FAST lane axis: the lanes of a row
  // read its C floats as one contiguous run (a thread per row and chunk-per-block read 32 bytes out of every 256: measured 35 us)
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ float s_a[256], s_b[256];
  const int bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int C8 = (C + 7) / 8;
  const int nd = tile_start[((size_t)bi * tiles_max + tiles_max - 1) * 16 + 1];   // entries of this shape's list
  constexpr int RB = 512;
  if ((int)blockIdx.x * RB >= nd && blockIdx.x != 0) return;
  const int cg = C / G;
  // Requested BEFORE the statistics, which they do not depend on (a wave issues in order: behind all_group_stats -- a global round trip, a
  // reduction and a barrier -- the affine parameters and the first rows were two more dependent round trips): this thread's channel
  // parameters (C <= 256 = one channel per thread) and its first four rows.
  const float g_pre = tid < C ? gamma[tid] : 0.f, b_pre = tid < C ? beta[tid] : 0.f;
  const int per_pass = blockDim.x / C8;                       // rows per pass (C8 <= 32)
  const int c8 = tid % C8, rl = tid / C8;
  const int nch = min(8, C - c8 * 8);
  const bool vec = (C & 3) == 0 && nch == 8;
  auto load_row = [&](int j, float (&in)[8]) {
    if (dense_in) {
      const int v = dil_list[(size_t)bi * n_rows_max + j];
      const float *xb = x + ((size_t)bi * C + c8 * 8) * V + v;
#pragma unroll
      for (int u = 0; u < 8; ++u) in[u] = xb[(size_t)min(u, nch - 1) * V];
    } else {
      const float *row = x + ((size_t)bi * n_rows_max + j) * C + c8 * 8;
      if (vec) {
        const float4 p = *reinterpret_cast<const float4 *>(row), q = *reinterpret_cast<const float4 *>(row + 4);
        in[0] = p.x; in[1] = p.y; in[2] = p.z; in[3] = p.w; in[4] = q.x; in[5] = q.y; in[6] = q.z; in[7] = q.w;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) in[u] = row[min(u, nch - 1)];
      }
    }
  };
  const int j_end = min(nd, (int)(blockIdx.x + 1) * RB), j_first = blockIdx.x * RB + rl;
  const bool pre4 = rl < per_pass && j_first + 3 * per_pass < j_end;
  float pin0[8], pin1[8], pin2[8], pin3[8];
  if (pre4) { load_row(j_first, pin0); load_row(j_first + per_pass, pin1); load_row(j_first + 2 * per_pass, pin2); load_row(j_first + 3 * per_pass, pin3); }
  all_group_stats(partial, bi, G, S, (double)cg * V, eps, s_mean, s_rstd);
  for (int ch = tid; ch < C8 * 8; ch += blockDim.x) {   // (one pass: C8 * 8 <= 256)
    float a = 0.f, bsh = 0.f;
    if (ch < C) {
      const int g = ch / cg;
      a = g_pre * s_rstd[g];
      bsh = b_pre - s_mean[g] * a;
    }
    s_a[ch] = a; s_b[ch] = bsh;
  }
  __syncthreads();
  bool sat = false;
  if (blockIdx.x == 0 && tid < C8) {   // the constant records: GroupNorm + Swish of the bias
    const int c8 = tid, nch = min(8, C - c8 * 8);
    float fill[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = c8 * 8 + j;
      float t = (ch < C && bias ? bias[ch] : 0.f) * s_a[ch] + s_b[ch];
      if (act == 1) t = swishf(t);
      fill[j] = ch < C ? t * act_scale : 0.f;
      sat |= !(fabsf(fill[j]) <= 65504.f);
    }
    uint4 ph, pl;
    split_rec(fill, ph, pl);
    const_h2[((size_t)bi * C8 + c8) * 2 + 0] = ph;
    const_h2[((size_t)bi * C8 + c8) * 2 + 1] = pl;
    const f16x8 hh = *reinterpret_cast<const f16x8 *>(&ph), ll = *reinterpret_cast<const f16x8 *>(&pl);
    for (int j = 0; j < nch; ++j) const_f32[(size_t)bi * C + c8 * 8 + j] = ((float)hh[j] + (float)ll[j]) * (1.0f / act_scale);
  }
  float ca[8], cb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ca[j] = s_a[c8 * 8 + j]; cb[j] = s_b[c8 * 8 + j]; }
  if (rl < per_pass) {
    auto emit_row = [&](int j, const float (&in)[8]) {
      float val[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float t = in[u] * ca[u] + cb[u];   // channels >= C: a = b = 0
        if (act == 1) t = swishf(t);
        val[u] = t * act_scale;
        sat |= !(fabsf(val[u]) <= 65504.f);
      }
      uint4 ph, pl;
      split_rec(val, ph, pl);
      rows_h2[(((size_t)bi * C8 + c8) * 2 + 0) * n_rows_max + j] = ph;
      rows_h2[(((size_t)bi * C8 + c8) * 2 + 1) * n_rows_max + j] = pl;
    };
    // four rows per step, their loads in flight together (one row per step was RB / per_pass = 16+ dependent round trips per workgroup)
    int j = j_first;
    if (pre4) {
      emit_row(j, pin0); emit_row(j + per_pass, pin1); emit_row(j + 2 * per_pass, pin2); emit_row(j + 3 * per_pass, pin3);
      j += 4 * per_pass;
    }
    for (; j + 3 * per_pass < j_end; j += 4 * per_pass) {
      float in0[8], in1[8], in2[8], in3[8];
      load_row(j, in0); load_row(j + per_pass, in1); load_row(j + 2 * per_pass, in2); load_row(j + 3 * per_pass, in3);
      emit_row(j, in0); emit_row(j + per_pass, in1); emit_row(j + 2 * per_pass, in2); emit_row(j + 3 * per_pass, in3);
    }
    for (; j < j_end; j += per_pass) {
      float in0[8];
      load_row(j, in0);
      emit_row(j, in0);
    }
  }
  if (saturated != nullptr && __ballot(sat) != 0ull && lane == __ffsll((long long)__ballot(sat)) - 1) atomicOr(saturated, 1u);
}

extern "C" int bdm_group_norm_to_h2_rows(int b, int c, int v, int groups, const float *x, int dense_in, int n_rows_max,
                                         const int *dil_list, const int *tile_start, int tiles_max, const float *bias,
                                         const float *gamma, const float *beta, float eps, int act, float act_scale, void *rows_h2,
                                         void *const_h2, float *const_f32, const void *partial, int slices,
                                         unsigned int *saturated, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && c <= 256 && v >= 1 && groups >= 1 && groups <= 64 && c % groups == 0 && (c / groups) >= 4 &&
                  partial != nullptr && slices >= 1 && x != nullptr && dil_list != nullptr && tile_start != nullptr && tiles_max >= 1 &&
                  n_rows_max >= 1 && rows_h2 && const_h2 && const_f32,
              "group_norm_to_h2_rows: bad arguments (<= 256 channels)");
  {
    int ex = 0;
    BDM_REQUIRE(act_scale > 0.f && act_scale < INFINITY && frexpf(act_scale, &ex) == 0.5f,
                "group_norm_to_h2_rows: act_scale must be a power of two (got %g)", (double)act_scale);
  }
  if (b == 0) return BDM_OK;
  dim3 grid(cdiv(n_rows_max, 512), b);
  hipLaunchKernelGGL(to_h2_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, v, groups, slices, n_rows_max, tiles_max, dense_in, x,
                     dil_list, tile_start, bias, (const double *)partial, gamma, beta, eps, act, act_scale, (uint4 *)rows_h2,
                     (uint4 *)const_h2, const_f32, saturated);
  return launch_status("group_norm_to_h2_rows");
}

// ---------------------------------------------------------------------------------------------------------------------
// class constants of the second convolution
// ---------------------------------------------------------------------------------------------------------------------
// wsum[k][ci][co] = sum over the taps that stay inside the grid for class k of w[co][ci][tap]  (fp64; once per weight tensor)
__global__ void class_weight_sums_kernel(int cout, int cin, const float *__restrict__ w, double *__restrict__ wsum) {
  const long long total = 27ll * cin * cout;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(e % cout), ci = (int)((e / cout) % cin), k = (int)(e / ((long long)cout * cin));
    const int cls[3] = {k / 9, (k / 3) % 3, k % 3};
    double a = 0.0;
    for (int t = 0; t < 27; ++t) {
      const int d[3] = {t / 9 - 1, (t / 3) % 3 - 1, t % 3 - 1};
      bool in = true;
      for (int ax = 0; ax < 3; ++ax) in = in && !(cls[ax] == 0 && d[ax] == -1) && !(cls[ax] == 2 && d[ax] == 1);
      if (in) a += (double)w[((size_t)co * cin + ci) * 27 + t];
    }
    wsum[e] = a;
  }
}
extern "C" size_t bdm_conv3d_class_weight_elems(int cout, int cin) { return (size_t)27 * cin * cout; }
extern "C" int bdm_conv3d_class_weight_sums(int cout, int cin, const float *w, void *wsum, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1 && w != nullptr && wsum != nullptr, "conv3d_class_weight_sums: bad arguments");
  hipLaunchKernelGGL(class_weight_sums_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, cout, cin, w, (double *)wsum);
  return launch_status("conv3d_class_weight_sums");
}

// class_vals (b, 27, cout) = bias + wsum[k] . const_f32[b]; and slice `slice0 + k` of the GroupNorm partials of the convolution's
// output: class_count[b][k] x (sum, sum of squares) of the class's values per group.
__global__ __launch_bounds__(256) void class_constants_kernel(int cin, int cout, const double *__restrict__ wsum, const float *__restrict__ bias,
                                                              const float *__restrict__ const_f32, const int *__restrict__ class_count,
                                                              float *__restrict__ class_vals, int G, int S, int slice0,
                                                              double *__restrict__ gn_partial) {
  __shared__ float s_c[256];
  __shared__ float s_v[256];
  const int k = blockIdx.x, bi = blockIdx.y, tid = threadIdx.x;
  for (int ci = tid; ci < cin; ci += blockDim.x) s_c[ci] = const_f32[(size_t)bi * cin + ci];
  __syncthreads();
  const double *wk = wsum + (size_t)k * cin * cout;
  for (int co = tid; co < cout; co += blockDim.x) {
    double a = bias ? (double)bias[co] : 0.0;
#pragma unroll 8   // (eight weight loads in flight; the sum stays in ascending ci)
    for (int ci = 0; ci < cin; ++ci) a += wk[(size_t)ci * cout + co] * (double)s_c[ci];
    const float v = (float)a;
    class_vals[((size_t)bi * 27 + k) * cout + co] = v;
    s_v[co] = v;
  }
  __syncthreads();
  if (gn_partial != nullptr && tid < G) {
    const int cg = cout / G;
    double a = 0.0, q = 0.0;
    for (int j = 0; j < cg; ++j) { const double v = (double)s_v[tid * cg + j]; a += v; q += v * v; }
    const double n = (double)class_count[(size_t)bi * 27 + k];
    double *dst = gn_partial + (((size_t)bi * G + tid) * S + slice0 + k) * 2;
    dst[0] = n * a;
    dst[1] = n * q;
  }
}
extern "C" int bdm_conv3d_class_constants(int b, int cin, int cout, const void *wsum, const float *bias, const float *const_f32,
                                          const int *class_count, float *class_vals, int groups, void *gn_partial, int slices,
                                          int slice0, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cin <= 256 && cout >= 1 && cout <= 256 && wsum && const_f32 && class_count && class_vals,
              "conv3d_class_constants: bad arguments (<= 256 channels)");
  BDM_REQUIRE(gn_partial == nullptr || (groups >= 1 && groups <= 64 && cout % groups == 0 && slice0 >= 0 && slice0 + 27 <= slices),
              "conv3d_class_constants: bad GroupNorm slice arguments");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(class_constants_kernel, dim3(27, b), dim3(256), 0, (hipStream_t)stream, cin, cout, (const double *)wsum, bias, const_f32,
                     class_count, class_vals, groups, slices, slice0, (double *)gn_partial);
  return launch_status("conv3d_class_constants");
}

// ---------------------------------------------------------------------------------------------------------------------
// SE gate from rows + class constants
// ---------------------------------------------------------------------------------------------------------------------
#define SE_SLABS 64
struct GnFoldC {
  const double *partial;  // (b, G, S, 2) or NULL
  int S, G, l;
  const float *gamma, *beta;
  float eps;
};

// grid (SE_SLABS, b): per-channel affine forms of GroupNorm-2 from the partials (every block; slab 0 stores them), then the slab's
// share of sum_rows swish(a x + b) per channel -> part (b, SE_SLABS, c).  Rows are channel-contiguous: a wave reads 256 bytes of a row.
__global__ __launch_bounds__(256) void se_rows_partial_kernel(int c, int V, int G, int S, int n_rows_max, int tiles_max,
                                                              const float *__restrict__ rows, const int *__restrict__ tile_start,
                                                              const double *__restrict__ partial, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float eps, float2 *__restrict__ coef,
                                                              float *__restrict__ part, GnFoldC pf, float2 *__restrict__ pf_coef) {
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ float s_a[256], s_b[256];
  __shared__ float s_acc[256];
  const int slab = blockIdx.x, bi = blockIdx.y, tid = threadIdx.x;
  const int cg = c / G;
  all_group_stats(partial, bi, G, S, (double)cg * V, eps, s_mean, s_rstd);
  for (int ch = tid; ch < c; ch += blockDim.x) {
    const float ga = gamma[ch] * s_rstd[ch / cg];
    s_a[ch] = ga;
    s_b[ch] = beta[ch] - s_mean[ch / cg] * ga;
  }
  __syncthreads();
  if (slab == 0) {
    for (int ch = tid; ch < c; ch += blockDim.x) coef[(size_t)bi * c + ch] = make_float2(s_a[ch], s_b[ch]);
    if (pf.partial != nullptr) {   // the point branch's GroupNorm (same channel count): affine forms for the devoxelisation kernel
      const int cgp = c / pf.G;
      for (int ch = tid; ch < c; ch += blockDim.x) {
        const int gp = ch / cgp;
        double a = 0.0, q = 0.0;
        const double *p = pf.partial + ((size_t)bi * pf.G + gp) * pf.S * 2;
        for (int s = 0; s < pf.S; ++s) { a += p[2 * s]; q += p[2 * s + 1]; }
        const double cnt = (double)cgp * pf.l, mu = a / cnt;
        double var = q / cnt - mu * mu;
        if (var < 0) var = 0;
        const float ga = pf.gamma[ch] * (float)(1.0 / sqrt(var + (double)pf.eps));
        pf_coef[(size_t)bi * c + ch] = make_float2(ga, pf.beta[ch] - (float)mu * ga);
      }
    }
  }
  const int nd = tile_start[((size_t)bi * tiles_max + tiles_max - 1) * 16 + 1];
  const int per = (nd + SE_SLABS - 1) / SE_SLABS, j_lo = slab * per, j_hi = min(nd, j_lo + per);
  const int CL = c < 256 ? c : 256, RL = 256 / CL;          // channels per pass, row lanes
  const int cl = tid % CL, rl = tid / CL;
  for (int c0 = 0; c0 < c; c0 += CL) {
    const int ch = c0 + cl;
    float acc = 0.f;
    if (rl < RL && ch < c) {
      const float ga = s_a[ch], be = s_b[ch];
      const float *col = rows + (size_t)bi * n_rows_max * c + ch;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;   // rows in flight per thread: eight, then four (one dependent load per step was
      int j = j_lo + rl;                               // latency-bound; a slab of ~200 rows was still twelve trips with four)
      for (; j + 7 * RL < j_hi; j += 8 * RL) {
        const float v0 = col[(size_t)j * c], v1 = col[(size_t)(j + RL) * c], v2 = col[(size_t)(j + 2 * RL) * c], v3 = col[(size_t)(j + 3 * RL) * c];
        const float v4 = col[(size_t)(j + 4 * RL) * c], v5 = col[(size_t)(j + 5 * RL) * c], v6 = col[(size_t)(j + 6 * RL) * c], v7 = col[(size_t)(j + 7 * RL) * c];
        a0 += swishf(v0 * ga + be); a1 += swishf(v1 * ga + be); a2 += swishf(v2 * ga + be); a3 += swishf(v3 * ga + be);
        a0 += swishf(v4 * ga + be); a1 += swishf(v5 * ga + be); a2 += swishf(v6 * ga + be); a3 += swishf(v7 * ga + be);
      }
      for (; j + 3 * RL < j_hi; j += 4 * RL) {
        const float v0 = col[(size_t)j * c], v1 = col[(size_t)(j + RL) * c], v2 = col[(size_t)(j + 2 * RL) * c], v3 = col[(size_t)(j + 3 * RL) * c];
        a0 += swishf(v0 * ga + be); a1 += swishf(v1 * ga + be); a2 += swishf(v2 * ga + be); a3 += swishf(v3 * ga + be);
      }
      for (; j < j_hi; j += RL) a0 += swishf(col[(size_t)j * c] * ga + be);
      acc = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    s_acc[tid] = acc;
    __syncthreads();
    if (tid < CL && c0 + tid < c) {
      float a = 0.f;
      for (int r2 = 0; r2 < RL; ++r2) a += s_acc[r2 * CL + tid];   // row lanes in order
      part[((size_t)bi * SE_SLABS + slab) * c + c0 + tid] = a;
    }
  }
}

// grid (b): channel means = (slabs in order + class_count x swish(a const + b)) / V, then the SE block's two FC layers (se.py:8-19)
__global__ void se_rows_fc_kernel(int c, int h, int V, const float *__restrict__ part, const float2 *__restrict__ coef,
                                  const float *__restrict__ class_vals, const int *__restrict__ class_count,
                                  const float *__restrict__ w1, const float *__restrict__ w2, float *__restrict__ mean,
                                  float *__restrict__ gate) {
  extern __shared__ float sh[];  // s[c], hid[h]
  float *s = sh, *hid = sh + c;
  const int bi = blockIdx.x;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    double a = 0.0;
    for (int sl = 0; sl < SE_SLABS; ++sl) a += (double)part[((size_t)bi * SE_SLABS + sl) * c + i];
    const float2 ab = coef[(size_t)bi * c + i];
    // the 27 class constants are read UNCONDITIONALLY, all in flight at once (a class without voxels has a finite constant and
    // contributes 0 x value: the sum keeps its bits).  Loading them under `if (count)` made 27 dependent round trips of them: 17 us
    // for this 16-workgroup kernel (round 5, rocprofv3 trace).
    float cv[27];
    int cn[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      cn[k] = class_count[(size_t)bi * 27 + k];
      cv[k] = class_vals[((size_t)bi * 27 + k) * c + i];
    }
#pragma unroll
    for (int k = 0; k < 27; ++k)
      if (cn[k]) a += (double)cn[k] * (double)swishf(cv[k] * ab.x + ab.y);
    const float m = (float)(a / (double)V);
    s[i] = m;
    mean[(size_t)bi * c + i] = m;
  }
  __syncthreads();
  if (w1 == nullptr) return;
  se_hidden_layer(c, h, w1, s, hid);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) gate[(size_t)bi * c + i] = se_gate_of(i, h, w2, hid);
}

static int se_rows_impl(int b, int c, int hidden, int v, int groups, const float *rows, int n_rows_max, const int *tile_start,
                        int tiles_max, const float *class_vals, const int *class_count, const void *gn_partial, int slices,
                        const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *part_ws,
                        float *mean_ws, float *coef, float *gate, GnFoldC pf, float *pf_coef, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && c <= 256 && hidden >= 1 && v >= 1 && groups >= 1 && groups <= 64 && c % groups == 0 && slices >= 1 &&
                  rows && tile_start && class_vals && class_count && gn_partial && part_ws && mean_ws && coef,
              "se_gate_gn_rows: bad arguments (<= 256 channels)");
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(se_rows_partial_kernel, dim3(SE_SLABS, b), dim3(256), 0, s, c, v, groups, slices, n_rows_max, tiles_max, rows, tile_start,
                     (const double *)gn_partial, gamma, beta, eps, (float2 *)coef, part_ws, pf, (float2 *)pf_coef);
  int rc = launch_status("se_rows_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(se_rows_fc_kernel, dim3(b), dim3(256), (c + hidden) * sizeof(float), s, c, hidden, v, part_ws, (const float2 *)coef,
                     class_vals, class_count, w1, w2, mean_ws, gate);
  return launch_status("se_rows_fc");
}
extern "C" size_t bdm_se_gate_gn_rows_workspace_elems(int b, int c) { return (size_t)b * SE_SLABS * c; }
extern "C" int bdm_se_gate_gn_rows(int b, int c, int hidden, int v, int groups, const float *rows, int n_rows_max, const int *tile_start,
                                   int tiles_max, const float *class_vals, const int *class_count, const void *gn_partial, int slices,
                                   const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *part_ws,
                                   float *mean_ws, float *coef, float *gate, void *stream) {
  return se_rows_impl(b, c, hidden, v, groups, rows, n_rows_max, tile_start, tiles_max, class_vals, class_count, gn_partial, slices, gamma,
                      beta, eps, w1, w2, part_ws, mean_ws, coef, gate, GnFoldC{}, nullptr, stream);
}
extern "C" int bdm_se_gate_gn_rows_pf(int b, int c, int hidden, int v, int groups, const float *rows, int n_rows_max,
                                      const int *tile_start, int tiles_max, const float *class_vals, const int *class_count,
                                      const void *gn_partial, int slices, const float *gamma, const float *beta, float eps,
                                      const float *w1, const float *w2, float *part_ws, float *mean_ws, float *coef, float *gate,
                                      const void *pf_partial, int pf_slices, int pf_groups, int pf_n, const float *pf_gamma,
                                      const float *pf_beta, float pf_eps, float *pf_coef, void *stream) {
  BDM_REQUIRE(pf_partial != nullptr && pf_slices >= 1 && pf_groups >= 1 && c % pf_groups == 0 && pf_n >= 1 && pf_gamma && pf_beta && pf_coef,
              "se_gate_gn_rows_pf: bad point-branch arguments");
  GnFoldC pf{(const double *)pf_partial, pf_slices, pf_groups, pf_n, pf_gamma, pf_beta, pf_eps};
  return se_rows_impl(b, c, hidden, v, groups, rows, n_rows_max, tile_start, tiles_max, class_vals, class_count, gn_partial, slices, gamma,
                      beta, eps, w1, w2, part_ws, mean_ws, coef, gate, pf, pf_coef, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// devoxelisation from rows: out[b][c][i] = sum_corners w swish(a row[corner][c] + b) gate[c] (+ point branch)
// ---------------------------------------------------------------------------------------------------------------------
// One workgroup = 64 points x all channels (passes of 64): 16 lanes per point, 4 channels per lane (16-byte row pieces: a point's
// corner row is read as 256 contiguous bytes), results through an LDS tile so that the channel-first output is written as 256-byte
// runs of points.  A corner outside the list cannot occur for a point of the cloud (its cell is occupied, so the corner is in the
// once-dilated set); such an index reads the class constant 13 (interior) -- defined, never taken.
__global__ __launch_bounds__(256) void devox_rows_kernel(int c, int n, int r, int n_rows_max, const float *__restrict__ coords,
                                                         const float *__restrict__ rows, const int *__restrict__ dil_index,
                                                         const float *__restrict__ class_vals, const float2 *__restrict__ coef,
                                                         const float *__restrict__ gate, const float *__restrict__ add, long long bs_a,
                                                         int ld_a, const float2 *__restrict__ add_coef, float *__restrict__ out,
                                                         long long bs_o, int ld_o) {
#pragma clang fp contract(off)
  __shared__ float tile[64][65];
  const int bi = blockIdx.y, p0 = blockIdx.x * 64, tid = threadIdx.x, l16 = tid & 15, ps = tid >> 4;   // 16 point slots per pass
  const int r2 = r * r, r3 = r2 * r;
  const float *pc = coords + (size_t)bi * 3 * n;
  const int *di = dil_index + (size_t)bi * r3;
  const float *rb = rows + (size_t)bi * n_rows_max * c;
  const float *cv = class_vals + ((size_t)bi * 27 + 13) * c;
  // the four points of this thread's slot (pass t handles point p0 + t * 16 + ps): weights and corner rows
  float w8[4][8];
  int j8[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = min(p0 + t * 16 + ps, n - 1);
    const float x = pc[i], y = pc[n + i], z = pc[2 * n + i];
    const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
    const float x1 = x - xl, y1 = y - yl, z1 = z - zl;
    const float x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
    w8[t][0] = x0 * y0 * z0; w8[t][1] = x0 * y0 * z1; w8[t][2] = x0 * y1 * z0; w8[t][3] = x0 * y1 * z1;
    w8[t][4] = x1 * y0 * z0; w8[t][5] = x1 * y0 * z1; w8[t][6] = x1 * y1 * z0; w8[t][7] = x1 * y1 * z1;
    const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
    const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
    const int idx[8] = {i000, i000 + sz, i000 + sy, i000 + sy + sz, i000 + sx, i000 + sx + sz, i000 + sx + sy, i000 + sx + sy + sz};
#pragma unroll
    for (int q = 0; q < 8; ++q) j8[t][q] = di[idx[q]];
  }
  for (int c0 = 0; c0 < c; c0 += 64) {
    const int ch = c0 + 4 * l16;
    const bool cok = ch + 3 < c;
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga, gs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (cok) {
      const float2 e0 = coef[(size_t)bi * c + ch], e1 = coef[(size_t)bi * c + ch + 1], e2 = coef[(size_t)bi * c + ch + 2], e3 = coef[(size_t)bi * c + ch + 3];
      ga = make_float4(e0.x, e1.x, e2.x, e3.x);
      gb = make_float4(e0.y, e1.y, e2.y, e3.y);
      if (gate) gs = make_float4(gate[(size_t)bi * c + ch], gate[(size_t)bi * c + ch + 1], gate[(size_t)bi * c + ch + 2], gate[(size_t)bi * c + ch + 3]);
    }
    __syncthreads();   // the previous pass's tile has been written out
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cok) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int j = j8[t][q];
          const float4 g = j >= 0 ? *reinterpret_cast<const float4 *>(rb + (size_t)j * c + ch) : *reinterpret_cast<const float4 *>(cv + ch);
          const float wq = w8[t][q];
          if (q == 0) {
            acc.x = wq * (swishf(g.x * ga.x + gb.x) * gs.x); acc.y = wq * (swishf(g.y * ga.y + gb.y) * gs.y);
            acc.z = wq * (swishf(g.z * ga.z + gb.z) * gs.z); acc.w = wq * (swishf(g.w * ga.w + gb.w) * gs.w);
          } else {
            acc.x += wq * (swishf(g.x * ga.x + gb.x) * gs.x); acc.y += wq * (swishf(g.y * ga.y + gb.y) * gs.y);
            acc.z += wq * (swishf(g.z * ga.z + gb.z) * gs.z); acc.w += wq * (swishf(g.w * ga.w + gb.w) * gs.w);
          }
        }
      }
      const int pl = t * 16 + ps;
      tile[4 * l16 + 0][pl] = acc.x; tile[4 * l16 + 1][pl] = acc.y; tile[4 * l16 + 2][pl] = acc.z; tile[4 * l16 + 3][pl] = acc.w;
    }
    // 64 channels x 64 points: a wave writes one channel's 64 points (256 bytes) per step.  The point branch's values of the wave's 16
    // channels are fetched BEFORE the barrier, all in flight (they do not depend on the tile): read inside the store loop they were 16
    // dependent round trips per pass
    const int lane = tid & 63, wv = tid >> 6;
    float av[16];
    float2 ac[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int ci = c0 + wv + 4 * u, i = p0 + lane;
      const bool ok = ci < c && i < n;
      av[u] = 0.f; ac[u] = make_float2(0.f, 0.f);
      if (add) av[u] = add[(size_t)bi * bs_a + (size_t)(ok ? ci : 0) * ld_a + (ok ? i : 0)];
      if (add_coef) ac[u] = add_coef[(size_t)bi * c + (ok ? ci : 0)];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int cl = wv + 4 * u, ci = c0 + cl, i = p0 + lane;
      if (ci >= c || i >= n) continue;
      float acc = tile[cl][lane];
      if (add) {
        float a2 = av[u];
        if (add_coef) a2 = swishf(a2 * ac[u].x + ac[u].y);
        acc += a2;
      }
      out[(size_t)bi * bs_o + (size_t)ci * ld_o + i] = acc;
    }
  }
}

static int devox_rows_launch(int b, int c, int n, int r, const float *coords, const float *rows, int n_rows_max, const int *dil_index,
                             const float *class_vals, const float *coef, const float *gate, const float *add, long long bs_a, int ld_a,
                             const float *add_coef, float *out, long long bs_o, int ld_o, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 4 && (c & 3) == 0 && n >= 1 && r >= 1 && coords && rows && dil_index && class_vals && coef && out &&
                  (reinterpret_cast<size_t>(rows) & 15) == 0 && (reinterpret_cast<size_t>(class_vals) & 15) == 0,
              "devoxelize_gn_gate_add_rows: bad arguments (4 | channels, 16-byte aligned rows)");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(devox_rows_kernel, dim3(cdiv(n, 64), b), dim3(256), 0, (hipStream_t)stream, c, n, r, n_rows_max, coords, rows, dil_index,
                     class_vals, (const float2 *)coef, gate, add, bs_a, ld_a, (const float2 *)add_coef, out, bs_o, ld_o);
  return launch_status("devoxelize_gn_gate_add_rows");
}
extern "C" int bdm_devoxelize_gn_gate_add_rows(int b, int c, int n, int r, const float *coords, const float *rows, int n_rows_max,
                                               const int *dil_index, const float *class_vals, const float *coef, const float *gate,
                                               const float *add, long long bs_a, int ld_a, float *out, long long bs_o, int ld_o,
                                               void *stream) {
  return devox_rows_launch(b, c, n, r, coords, rows, n_rows_max, dil_index, class_vals, coef, gate, add, bs_a, ld_a, nullptr, out, bs_o,
                           ld_o, stream);
}
extern "C" int bdm_devoxelize_gn_gate_add_rows_pf(int b, int c, int n, int r, const float *coords, const float *rows, int n_rows_max,
                                                  const int *dil_index, const float *class_vals, const float *coef, const float *gate,
                                                  const float *add, long long bs_a, int ld_a, const float *add_coef, float *out,
                                                  long long bs_o, int ld_o, void *stream) {
  BDM_REQUIRE(add != nullptr && add_coef != nullptr, "devoxelize_gn_gate_add_rows_pf: add / add_coef is NULL");
  return devox_rows_launch(b, c, n, r, coords, rows, n_rows_max, dil_index, class_vals, coef, gate, add, bs_a, ld_a, add_coef, out, bs_o,
                           ld_o, stream);
}
