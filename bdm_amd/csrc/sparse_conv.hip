// sparse_conv.hip -- the FIRST 3x3x3 convolution of every PVConv, evaluated on the occupied voxels only.
//
// Its input is the freshly voxelised point cloud (pvconv.py:93-94): at most N of the r^3 cells are non-zero
// (measured 6.5 % at r = 32 / N = 4096, 10-25 % at r = 16, 9-18 % at r = 8).  Convolution is linear, so
//     out[:, v] = bias + sum_{tap} W_tap . vox[:, v + tap]
// only receives contributions from occupied cells u = v + tap.  Instead of 27*Cin*Cout MACs for each of the r^3
// output cells (22.1 GFLOP per shape for the 390 -> 32 convolution of sa_layers.0.0), the work is
//   1. compact the occupied cells of each shape (occ_list / occ_index), gather their mean features
//      Xc (B, Cin, n_occ) with the voxeliser's deterministic per-voxel summation order (values are bit-identical to
//      the dense grid's);
//   2. ONE batched fp32-MFMA GEMM  Y[b] (n_occ x 27*Cout) = Xc[b]^T (n_occ x Cin) . Wt (Cin x 27*Cout)
//      -- N_occ * 27 * Cin * Cout MACs, 8-15x fewer than the dense convolution;
//   3. an output-stationary gather: every output cell sums, in fixed tap order 0..26, the Y rows of its occupied
//      neighbours (occ_index lookup); rows whose 3x3 neighbourhood of grid rows is empty are pure bias.
// Deterministic (no atomics); differs from the dense kernel only in fp32 summation order.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"
#include "s3_split.h"

using namespace bdm;

// ---------------------------------------------------------------------------------------------------
// 1a. compaction of the occupied cells: cnt (B, r^3) -> occ_index (B, r^3) [-1 = empty], occ_list (B, n_max), n_occ (B)
// ---------------------------------------------------------------------------------------------------
__global__ void vox_compact_kernel(int r3, int n_max, const int *__restrict__ cnt, int *__restrict__ occ_index,
                                   int *__restrict__ occ_list, int *__restrict__ n_occ) {
  __shared__ int wave_tot[16];
  const int bi = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int *c = cnt + (size_t)bi * r3;
  int *oi = occ_index + (size_t)bi * r3;
  int *ol = occ_list + (size_t)bi * n_max;
  const int per = (r3 + T - 1) / T;
  const int lo = min(tid * per, r3), hi = min(lo + per, r3);
  int local = 0;
  for (int v = lo; v < hi; ++v) local += c[v] > 0;
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int off = 0, total = 0;
  for (int w = 0; w < (T >> 6); ++w) { if (w < wave) off += wave_tot[w]; total += wave_tot[w]; }
  int run = off + incl - local;
  for (int v = lo; v < hi; ++v) {
    if (c[v] > 0) { oi[v] = run; ol[run] = v; ++run; }
    else oi[v] = -1;
  }
  if (tid == 0) n_occ[bi] = total;
}

extern "C" int bdm_voxel_compact(int b, int r, int n_max, const int *cnt, int *occ_index, int *occ_list, int *n_occ,
                                 void *stream) {
  BDM_REQUIRE(b >= 0 && r >= 1 && n_max >= 1, "voxel_compact: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(vox_compact_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, r * r * r, n_max, cnt, occ_index,
                     occ_list, n_occ);
  return launch_status("voxel_compact");
}

// rowocc (b, r*r) uint8: does grid row (x, y) hold an occupied cell?  (the gather skips empty rows of the neighbourhood)
__global__ void row_occupancy_kernel(int r, const int *__restrict__ cnt, unsigned char *__restrict__ rowocc) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
  if (row >= r * r) return;
  const int *c = cnt + ((size_t)bi * r * r + row) * r;
  int any = 0;
  for (int z = 0; z < r; ++z) any |= c[z];
  rowocc[(size_t)bi * r * r + row] = any ? 1 : 0;
}
extern "C" int bdm_voxel_row_occupancy(int b, int r, const int *cnt, unsigned char *rowocc, void *stream) {
  BDM_REQUIRE(b >= 0 && r >= 1, "voxel_row_occupancy: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(row_occupancy_kernel, dim3(cdiv(r * r, 256), b), dim3(256), 0, (hipStream_t)stream, r, cnt, rowocc);
  return launch_status("voxel_row_occupancy");
}

// ---------------------------------------------------------------------------------------------------
// 1b. mean features of the occupied cells: Xc (B, C, n_max) channel-first; columns >= n_occ are zero
// ---------------------------------------------------------------------------------------------------
__global__ void sparse_vox_features_kernel(int c, int n, int r3, int n_max, const float *__restrict__ feat, long long bs_f,
                                           int ld_f, const int *__restrict__ cnt, const int *__restrict__ start,
                                           const int *__restrict__ sorted, const int *__restrict__ occ_list,
                                           const int *__restrict__ n_occ, float *__restrict__ xc) {
#pragma clang fp contract(off)  // same arithmetic as vox_reduce_kernel: the values equal the dense grid's bit for bit
  const int k = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.z;
  if (k >= n_max) return;
  const bool live = k < n_occ[bi];
  int cv = 0, s = 0;
  if (live) {
    const int v = occ_list[(size_t)bi * n_max + k];
    cv = cnt[(size_t)bi * r3 + v];
    s = start[(size_t)bi * r3 + v];
  }
  const int *so = sorted + (size_t)bi * n + s;
  const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
  const float *fb = feat + (size_t)bi * bs_f;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) {
    const float *f = fb + (size_t)ci * ld_f;
    float acc = 0.f;
    for (int q = 0; q < cv; ++q) acc = acc + f[so[q]] * inv;
    xc[((size_t)bi * c + ci) * n_max + k] = acc;
  }
}

extern "C" int bdm_sparse_voxel_features(int b, int c, int n, int r, int n_max, const float *features, long long bs_f,
                                         int ld_f, const int *cnt, const void *plan_workspace, const int *occ_list,
                                         const int *n_occ, float *xc, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && n >= 1 && r >= 1 && n_max >= 1, "sparse_voxel_features: bad sizes");
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(const_cast<void *>(plan_workspace), b, n, r3);
  dim3 grid(cdiv(n_max, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(sparse_vox_features_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, r3, n_max, features, bs_f,
                     ld_f, cnt, w.start, w.sorted, occ_list, n_occ, xc);
  return launch_status("sparse_voxel_features");
}

// weights (Cout, Cin, 3,3,3) -> Wt (Cin, 27*Cout): Wt[ci][tap*Cout + co]
__global__ void sparse_pack_kernel(int cout, int cin, const float *__restrict__ w, float *__restrict__ wt) {
  const long long total = (long long)cin * 27 * cout;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(e % cout), tap = (int)((e / cout) % 27), ci = (int)(e / (27ll * cout));
    wt[e] = w[((size_t)co * cin + ci) * 27 + tap];
  }
}
extern "C" int bdm_sparse_conv_pack_weights(int cout, int cin, const float *w, float *wt, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1, "sparse_conv_pack_weights: bad sizes");
  hipLaunchKernelGGL(sparse_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, cout, cin, w, wt);
  return launch_status("sparse_conv_pack_weights");
}

// ---------------------------------------------------------------------------------------------------
// 2'. bf16x6 form of steps 1b + 2 (the default): both GEMM operands are stored pre-split ("S3", s3_split.h) so the
//     GEMM is pure v_mfma_f32_32x32x16_bf16 work -- six exact 8x8-bit partial products per fp32 product, fp32
//     accumulation, smallest terms first; ~2.6x the rate of the fp32-input MFMA at fp32-grade accuracy (~2^-22).
//       Xs (B, G, 3, n_max) records : G = ceil(Cin/8) channel groups of the occupied cells' mean features
//       Ws (G, 3, 27*Cout) records  : Ws[g][s][tap*Cout+co] = split_s(w[co][8g..8g+7][tap])
// ---------------------------------------------------------------------------------------------------
// XCD-aware launch: the 8 feature rows of a (shape, channel group) unit (8 x 4N bytes) are gathered 4 bytes at a time by
// all k-blocks of the unit; consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the
// 1-D grid is decoded such that every k-block of a unit has the same id mod 8: the rows are fetched into ONE L2 once
// instead of missing in all eight (measured 329 -> see DESIGN.md for the 390-channel level).
__global__ void sparse_vox_features_s3_kernel(int c, int n, int r3, int n_max, int G, int units, int kblocks,
                                              const float *__restrict__ feat,
                                              long long bs_f, int ld_f, const int *__restrict__ cnt,
                                              const int *__restrict__ start, const int *__restrict__ sorted,
                                              const int *__restrict__ occ_list, const int *__restrict__ n_occ,
                                              unsigned short *__restrict__ xs) {
#pragma clang fp contract(off)  // same arithmetic as vox_reduce_kernel: the values equal the dense grid's bit for bit
  const int wg = blockIdx.x, span = 8 * kblocks;
  const int unit = (wg / span) * 8 + (wg % span) % 8, kb = (wg % span) / 8;
  if (unit >= units) return;
  const int bi = unit / G, g = unit % G;
  const int k = kb * blockDim.x + threadIdx.x;
  if (k >= n_max) return;
  const int nocc = n_occ[bi];
  // rows beyond the GEMM's last (128-row) tile that holds an occupied cell are never read: whole blocks of them are skipped
  if (kb * (int)blockDim.x >= ((nocc + 127) & ~127)) return;
  const bool live = k < nocc;
  int cv = 0, s = 0;
  if (live) {
    const int v = occ_list[(size_t)bi * n_max + k];
    cv = cnt[(size_t)bi * r3 + v];
    s = start[(size_t)bi * r3 + v];
  }
  const int *so = sorted + (size_t)bi * n + s;
  const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
  const float *fb = feat + (size_t)bi * bs_f + (size_t)g * 8 * ld_f;
  const int nch = min(8, c - g * 8);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  // four points of the cell at a time: their indices, then their 32 feature values, are fetched as independent loads (a
  // point-at-a-time loop pays two dependent round trips per point); the sums keep the list order
  for (int q0 = 0; q0 < cv; q0 += 4) {
    int p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = so[min(q0 + u, cv - 1)];
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[u][j] = fb[(size_t)min(j, nch - 1) * ld_f + p[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q0 + u < cv) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < nch) acc[j] = acc[j] + v[u][j] * inv;
      }
  }
  store_s3(xs + ((size_t)bi * G + g) * 3 * (size_t)n_max * 8, (size_t)k, (size_t)n_max, acc);
}

// LDS-cached form of the feature gather (n <= 4096 points): one workgroup per (shape, channel group) first copies the
// group's 8 feature rows (8 x 4n bytes, coalesced) into LDS and then gathers from LDS.  The global-memory form above issues
// one scattered 4-byte load per (point, channel): the working set of the waves resident on a CU (128 KB per unit) thrashes the
// vector L1 (PMC: 30 % hit rate, 38 M L1->L2 requests for 102 MB of features at the 390-channel level) and the kernel is
// bound by the texture-address unit (81 % of its wave cycles are issue stalls).  Same sums in the same order: same bits.
//   OUT = 0: bf16-triple records (sparse_gemm_s3);  OUT = 1: fp32 records + max |value| (sparse_gemm_h2 path)
// With few channel groups (32-channel layers: 4 groups x 16 shapes = 64 units) `ksplit` workgroups share a unit, each with
// its own copy of the rows and a slice of `cells_per` cells.
template <int OUT>
__global__ __launch_bounds__(1024) void sparse_vox_features_lds_kernel(int c, int n, int r3, int n_max, int G, int ksplit,
                                                                       int cells_per,
                                                                       const float *__restrict__ feat, long long bs_f, int ld_f,
                                                                       const int *__restrict__ cnt, const int *__restrict__ start,
                                                                       const int *__restrict__ sorted,
                                                                       const int *__restrict__ occ_list,
                                                                       const int *__restrict__ n_occ, void *__restrict__ outp,
                                                                       unsigned *__restrict__ amax) {
#pragma clang fp contract(off)  // same arithmetic as vox_reduce_kernel: the values equal the dense grid's bit for bit
  extern __shared__ float frows[];  // [8][n]
  constexpr int CPT = 4;            // cells per thread: n_max <= 4 * blockDim.x
  const int unit = blockIdx.x / ksplit, k0 = (blockIdx.x % ksplit) * cells_per, bi = unit / G, g = unit % G;
  const int tid = threadIdx.x, T = blockDim.x;
  const int nocc = min(n_occ[bi], n_max);
  // rows the consumers read: the GEMM's last 128-row tile that holds an occupied cell (S3) / every row (fp32 records: split pass)
  const int lim = min(OUT == 0 ? min((nocc + 127) & ~127, n_max) : n_max, k0 + cells_per);
  if (k0 >= lim) return;  // nothing to write in this slice
  const int nch = min(8, c - g * 8);
  // The group's rows go to registers FIRST (independent loads, 16 bytes each): a wave issues in order, so behind the index chain below
  // (cell -> count / start -> first list entries: three dependent round trips, each ending in a wait) they would not even be requested
  // before the chain is through -- four serial trips instead of three overlapped with one.
  const float *fb = feat + (size_t)bi * bs_f + (size_t)g * 8 * ld_f;
  const bool vec = (n & 3) == 0 && (ld_f & 3) == 0 && ((reinterpret_cast<size_t>(fb) & 15) == 0) && (n >> 2) <= T;   // 16-byte row pieces, one per thread and row
  const int n4 = n >> 2;
  float4 rowv[8];
  if (vec) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rowv[j] = reinterpret_cast<const float4 *>(fb + (size_t)min(j, nch - 1) * ld_f)[min(tid, n4 - 1)];
  }
  // the cells' point lists (two dependent index loads) are fetched while the rows stream in
  int cs[CPT], cc[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int k = k0 + tid + i * T;
    cs[i] = 0; cc[i] = 0;
    if (k < nocc && k < lim) {
      const int v = occ_list[(size_t)bi * n_max + k];
      cc[i] = cnt[(size_t)bi * r3 + v];
      cs[i] = start[(size_t)bi * r3 + v];
    }
  }
  // ... and so are the first four entries of each cell's point list (cells hold 1 - 3 points on these levels): after the barrier the
  // common cell needs LDS only (its list read there was one more dependent round trip per cell, two per thread at 32^3)
  int pre[CPT][4];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int *so = sorted + (size_t)bi * n + cs[i];
#pragma unroll
    for (int u = 0; u < 4; ++u) pre[i][u] = cc[i] > 0 ? so[min(u, cc[i] - 1)] : 0;
  }
  if (vec) {
    if (tid < n4) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        reinterpret_cast<float4 *>(frows + j * n)[tid] = j < nch ? rowv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else if ((n & 3) == 0 && (ld_f & 3) == 0 && ((reinterpret_cast<size_t>(fb) & 15) == 0)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float4 *src = reinterpret_cast<const float4 *>(fb + (size_t)min(j, nch - 1) * ld_f);
      float4 *dst = reinterpret_cast<float4 *>(frows + j * n);
      for (int p = tid; p < n4; p += T) {
        const float4 v = src[p];
        dst[p] = j < nch ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  } else {
    for (int j = 0; j < 8; ++j)
      for (int p = tid; p < n; p += T) frows[j * n + p] = j < nch ? fb[(size_t)j * ld_f + p] : 0.f;
  }
  __syncthreads();
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int k = k0 + tid + i * T;
    if (k >= lim) continue;
    const int cv = cc[i];
    const int *so = sorted + (size_t)bi * n + cs[i];
    const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int q0 = 0; q0 < cv; q0 += 4) {
      int p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] = q0 == 0 ? pre[i][u] : so[min(q0 + u, cv - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q0 + u < cv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = acc[j] + frows[j * n + p[u]] * inv;  // rows >= nch hold zeros
        }
    }
    if (OUT == 0) {
      store_s3(reinterpret_cast<unsigned short *>(outp) + ((size_t)bi * G + g) * 3 * (size_t)n_max * 8, (size_t)k, (size_t)n_max, acc);
    } else {
      float4 *o = reinterpret_cast<float4 *>(outp) + (((size_t)bi * G + g) * n_max + k) * 2;
      o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(acc[j]));
    }
  }
  if (OUT == 1) {
    m = wave_max(m);
    if ((tid & 63) == 0 && m > 0.f) atomicMax(amax + bi, __float_as_uint(m));  // per shape; non-negative floats order as unsigned ints
  }
}

// shared launcher (also used by bdm_sparse_voxel_features_f32 in sparse_conv_fused.hip); false: shape not covered
bool bdm_sparse_features_lds_launch(int out_kind, int b, int c, int n, int r3, int n_max, const float *features, long long bs_f,
                                    int ld_f, const int *cnt, const int *start, const int *sorted, const int *occ_list,
                                    const int *n_occ, void *out, unsigned *amax, hipStream_t stream, int *rc) {
  if (bdm_staging_choice() == 0 || n > 4096 || n_max > 4096) return false;  // BDM_STAGING=0 keeps the global-memory gather (common.h)
  const int G = (c + 7) / 8, units = b * G;
  // workgroups per unit: every extra one refills the rows, which costs more than the parallelism gains (measured: 32-channel
  // layer at 32^3, 8 slices: 132 -> 207 us)
  const int ksplit = 1;
  const int cells_per = ((n_max + ksplit - 1) / ksplit + 63) & ~63;
  const int T = cells_per > 1024 ? 1024 : 256;  // 4 cells per thread cover the slice
  const size_t smem = sizeof(float) * 8 * (size_t)n;
  *rc = BDM_OK;
  if (out_kind == 0) {
    static int granted0[16] = {0};  // per device: the attribute call costs microseconds of host time, once is enough
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (smem > 48 * 1024 && (dev < 0 || dev >= 16 || granted0[dev] < (int)smem)) {
      hipError_t e = hipFuncSetAttribute((const void *)sparse_vox_features_lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return false;
      if (dev >= 0 && dev < 16) granted0[dev] = (int)smem;
    }
    hipLaunchKernelGGL(sparse_vox_features_lds_kernel<0>, dim3(units * ksplit), dim3(T), smem, stream, c, n, r3, n_max, G, ksplit,
                       cells_per, features, bs_f, ld_f, cnt, start, sorted, occ_list, n_occ, out, amax);
  } else {
    static int granted1[16] = {0};  // per device: the attribute call costs microseconds of host time, once is enough
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (smem > 48 * 1024 && (dev < 0 || dev >= 16 || granted1[dev] < (int)smem)) {
      hipError_t e = hipFuncSetAttribute((const void *)sparse_vox_features_lds_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess) return false;
      if (dev >= 0 && dev < 16) granted1[dev] = (int)smem;
    }
    hipLaunchKernelGGL(sparse_vox_features_lds_kernel<1>, dim3(units * ksplit), dim3(T), smem, stream, c, n, r3, n_max, G, ksplit,
                       cells_per, features, bs_f, ld_f, cnt, start, sorted, occ_list, n_occ, out, amax);
  }
  return true;
}

extern "C" int bdm_sparse_voxel_features_s3(int b, int c, int n, int r, int n_max, const float *features, long long bs_f,
                                            int ld_f, const int *cnt, const void *plan_workspace, const int *occ_list,
                                            const int *n_occ, void *xs, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && n >= 1 && r >= 1 && n_max >= 1, "sparse_voxel_features_s3: bad sizes");
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(const_cast<void *>(plan_workspace), b, n, r3);
  int rc_lds = BDM_OK;
  if (bdm_sparse_features_lds_launch(0, b, c, n, r3, n_max, features, bs_f, ld_f, cnt, w.start, w.sorted, occ_list, n_occ, xs, nullptr,
                                     (hipStream_t)stream, &rc_lds))
    return launch_status("sparse_voxel_features_s3");
  const int G = (c + 7) / 8, units = b * G, kblocks = cdiv(n_max, 128);
  hipLaunchKernelGGL(sparse_vox_features_s3_kernel, dim3(cdiv(units, 8) * 8 * kblocks), dim3(128), 0, (hipStream_t)stream, c,
                     n, r3, n_max, G, units, kblocks, features, bs_f, ld_f, cnt, w.start, w.sorted, occ_list, n_occ,
                     (unsigned short *)xs);
  return launch_status("sparse_voxel_features_s3");
}

__global__ void sparse_pack_s3_kernel(int cout, int cin, const float *__restrict__ w, unsigned short *__restrict__ ws) {
  const int n27 = 27 * cout, G = (cin + 7) / 8;
  const long long total = (long long)G * n27;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(e / n27), col = (int)(e % n27), tap = col / cout, co = col % cout;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = g * 8 + j;
      v[j] = ci < cin ? w[((size_t)co * cin + ci) * 27 + tap] : 0.f;
    }
    store_s3(ws + (size_t)g * 3 * n27 * 8, (size_t)col, (size_t)n27, v);
  }
}
extern "C" size_t bdm_sparse_conv_s3_weight_elems(int cout, int cin) { return (size_t)((cin + 7) / 8) * 3 * 27 * cout * 8; }
extern "C" int bdm_sparse_conv_pack_weights_s3(int cout, int cin, const float *w, void *ws, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1, "sparse_conv_pack_weights_s3: bad sizes");
  hipLaunchKernelGGL(sparse_pack_s3_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, cout, cin, w, (unsigned short *)ws);
  return launch_status("sparse_conv_pack_weights_s3");
}

// Y[b] (M x N) = Xs[b]^T . Ws : 128 x 128 tile per workgroup (2 x 2 waves of 64 x 64), K = 32 (four record groups)
// per stage, register-prefetched: the 12 global loads of stage c+1 are in flight during the 48 MFMAs of stage c.
__global__ __launch_bounds__(256) void sparse_gemm_s3_kernel(int M, int G, int N, const uint4 *__restrict__ A,
                                                             const uint4 *__restrict__ Bw, const int *__restrict__ m_count,
                                                             const float *__restrict__ col_bias, long long bs_cb, float *__restrict__ Y) {
  constexpr int BM = 128, BN = 128, AI = 12 * BM / 256, BI = 12 * BN / 256;
  __shared__ uint4 As[12 * BM], Bs[12 * BN];  // [group-in-stage][split][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, bi = blockIdx.z;
  if (m_count && m0 >= m_count[bi]) return;  // rows beyond this shape's occupied cells
  const uint4 *Ab = A + (size_t)bi * G * 3 * M;
  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  float cbv[2] = {0.f, 0.f};   // (bdm_sparse_conv_gemm_s3_cb's addend of this lane's two columns, requested before the K loop)
  if (col_bias != nullptr) {
#pragma unroll
    for (int y = 0; y < 2; ++y) cbv[y] = col_bias[(size_t)bi * bs_cb + min(n0 + (wc * 2 + y) * 32 + li, N - 1)];
  }
  // Staging registers are native vectors and the loads carry neither a branch nor a select on the loaded value (either
  // makes the wave wait for memory inside the load phase, i.e. no prefetch): rows >= M / columns >= N read a clamped
  // address and only feed outputs that are never stored; the channel groups >= G of the last stage are zeroed when the
  // registers go to LDS.
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v ar[AI], br[BI];
  auto load_stage = [&](int g0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256, row = e % BM, gs = e / BM, g = min(g0 + gs / 3, G - 1), sp = gs % 3;
      ar[i] = *reinterpret_cast<const u32x4v *>(&Ab[(unsigned)((g * 3 + sp) * M + min(m0 + row, M - 1))]);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e = tid + i * 256, col = e % BN, gs = e / BN, g = min(g0 + gs / 3, G - 1), sp = gs % 3;
      br[i] = *reinterpret_cast<const u32x4v *>(&Bw[(unsigned)((g * 3 + sp) * N + min(n0 + col, N - 1))]);
    }
  };
  load_stage(0);
  for (int g0 = 0; g0 < G; g0 += 4) {
    __syncthreads();
    const u32x4v zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256, gs = e / BM;
      *reinterpret_cast<u32x4v *>(&As[e]) = (g0 + gs / 3 < G) ? ar[i] : zero;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e = tid + i * 256, gs = e / BN;
      *reinterpret_cast<u32x4v *>(&Bs[e]) = (g0 + gs / 3 < G) ? br[i] : zero;
    }
    __syncthreads();
    if (g0 + 4 < G) load_stage(g0 + 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          a[x][sp] = *reinterpret_cast<const bf16x8 *>(&As[((2 * kk + lh) * 3 + sp) * BM + (wr * 2 + x) * 32 + li]);
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          b[y][sp] = *reinterpret_cast<const bf16x8 *>(&Bs[((2 * kk + lh) * 3 + sp) * BN + (wc * 2 + y) * 32 + li]);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          f32x16 c = acc[x][y];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][2], b[y][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][1], b[y][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][0], b[y][0], c, 0, 0, 0);
          acc[x][y] = c;
        }
    }
  }
  // C/D map: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 -> 128-byte runs along n
  float *Yb = Y + (size_t)bi * M * N;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int nn = n0 + (wc * 2 + y) * 32 + li;
      const float cb = cbv[y];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wr * 2 + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && nn < N) Yb[(size_t)m * N + nn] = acc[x][y][r] + cb;
      }
    }
}

static int sparse_gemm_s3_launch(int b, int n_max, int cin, int n27, const void *xs, const void *ws, const int *n_occ,
                                 const float *col_bias, long long bs_cb, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && n_max >= 1 && cin >= 1 && n27 >= 1, "sparse_conv_gemm_s3: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(sparse_gemm_s3_kernel, dim3(cdiv(n27, 128), cdiv(n_max, 128), b), dim3(256), 0, (hipStream_t)stream,
                     n_max, (cin + 7) / 8, n27, (const uint4 *)xs, (const uint4 *)ws, n_occ, col_bias, bs_cb, y);
  return launch_status("sparse_conv_gemm_s3");
}

extern "C" int bdm_sparse_conv_gemm_s3(int b, int n_max, int cin, int n27, const void *xs, const void *ws, const int *n_occ,
                                       float *y, void *stream) {
  return sparse_gemm_s3_launch(b, n_max, cin, n27, xs, ws, n_occ, nullptr, 0, y, stream);
}

// + a per-shape column addend col_bias (b, n27): see bdm_sparse_conv_gemm_h2_cb
extern "C" int bdm_sparse_conv_gemm_s3_cb(int b, int n_max, int cin, int n27, const void *xs, const void *ws, const int *n_occ,
                                          const float *col_bias, long long bs_cb, float *y, void *stream) {
  BDM_REQUIRE(col_bias != nullptr && bs_cb >= n27, "sparse_conv_gemm_s3_cb: col_bias is null or its batch stride < n27");
  return sparse_gemm_s3_launch(b, n_max, cin, n27, xs, ws, n_occ, col_bias, bs_cb, y, stream);
}

// ---------------------------------------------------------------------------------------------------
// 2''. Steps 1b + 2 for an input whose feature channels are a GATHER of a per-image map (the projection conditioning of PC^2:
//      x_in[i] = [xyz_i, F[pix_i]] with F the hoisted conditioning image, projection_model.py:179-231).  The GEMM commutes with the
//      gather and with the voxel mean:  Y[k] = mean_{i in cell k} (W . [xyz_i, F[pix_i]]) = mean_i (Wx . xyz_i + Hmap[pix_i])  with
//      Hmap = F . Wf^T (HW x 27*Cout) computed ONCE per trajectory (the image is step-invariant, like the image encoder itself).
//      Per step the 390-channel feature gather (step 1b) and the K = 390 GEMM (step 2) become ONE pass that reads a 27*Cout-float
//      row per point: 226 MB instead of the features' 102 MB + a 56 GFLOP bf16x6 GEMM at B = 16.  Same sums up to the
//      reassociation of each dot product (<= 1e-6 relative, tests/test_hip_dense.py); deterministic (ascending point index).
//      One workgroup = 8 cells of one shape; a lane owns 4 consecutive columns.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sparse_rows_from_map_kernel(int n, int r3, int n_max, int n27, int hw,
                                                                   const float *__restrict__ hmap, const int *__restrict__ pix,
                                                                   const float *__restrict__ xyz, const float *__restrict__ wx,
                                                                   const int *__restrict__ cnt, const int *__restrict__ start,
                                                                   const int *__restrict__ sorted, const int *__restrict__ occ_list,
                                                                   const int *__restrict__ n_occ, float *__restrict__ y) {
  const int bi = blockIdx.y, k0 = blockIdx.x * 8, nocc = min(n_occ[bi], n_max);
  if (k0 >= nocc) return;
  const int c4 = threadIdx.x, c4c = min(c4, (n27 >> 2) - 1);  // float4 column (threads beyond the row only help with the index chase)
  const float4 w0 = make_float4(wx[(c4c * 4 + 0) * 3 + 0], wx[(c4c * 4 + 1) * 3 + 0], wx[(c4c * 4 + 2) * 3 + 0], wx[(c4c * 4 + 3) * 3 + 0]);
  const float4 w1 = make_float4(wx[(c4c * 4 + 0) * 3 + 1], wx[(c4c * 4 + 1) * 3 + 1], wx[(c4c * 4 + 2) * 3 + 1], wx[(c4c * 4 + 3) * 3 + 1]);
  const float4 w2 = make_float4(wx[(c4c * 4 + 0) * 3 + 2], wx[(c4c * 4 + 1) * 3 + 2], wx[(c4c * 4 + 2) * 3 + 2], wx[(c4c * 4 + 3) * 3 + 2]);
  const float *px = xyz + (size_t)bi * 3 * n;
  const int *pp = pix + (size_t)bi * n;
  const float4 *hb = reinterpret_cast<const float4 *>(hmap + (size_t)bi * hw * n27);
  const int n27q = n27 >> 2;
  // The index chase of the workgroup's 8 cells (cell -> count / list start -> point -> owning pixel + coordinates) is done ONCE, by 8 and
  // then 32 threads, into LDS; every thread then has the map rows of four cells' points in flight together.  Chased by every thread for
  // one cell after the other it was ~5 dependent round trips per cell, 40 per workgroup: 78 us for the level-0 layer (round 5).
  __shared__ int s_cv[8], s_st[8], s_pt[8][4], s_pix[8][4];
  __shared__ float s_xyz[8][4][3];
  if (threadIdx.x < 8) {
    const int k = k0 + threadIdx.x;
    int cv = 0, st = 0;
    if (k < nocc) {
      const int v = occ_list[(size_t)bi * n_max + k];
      cv = cnt[(size_t)bi * r3 + v];
      st = start[(size_t)bi * r3 + v];
    }
    s_cv[threadIdx.x] = cv; s_st[threadIdx.x] = st;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int cell = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (q < s_cv[cell]) {
      const int i = sorted[(size_t)bi * n + s_st[cell] + q];
      s_pt[cell][q] = i;
      s_pix[cell][q] = pp[i];
      s_xyz[cell][q][0] = px[i]; s_xyz[cell][q][1] = px[n + i]; s_xyz[cell][q][2] = px[2 * n + i];
    }
  }
  __syncthreads();
  if (c4 * 4 >= n27) return;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int half = 0; half < 2; ++half) {
    float4 h[4][4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cell = half * 4 + cl;
        const int p = q < s_cv[cell] ? s_pix[cell][q] : -1;
        h[cl][q] = p >= 0 ? hb[(size_t)p * n27q + c4] : zero;
      }
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      const int cell = half * 4 + cl, k = k0 + cell;
      if (k >= nocc) continue;
      const int cv = s_cv[cell];
      const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
      float4 acc = zero;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (q < cv) {
          const float x0 = s_xyz[cell][q][0], y0 = s_xyz[cell][q][1], z0 = s_xyz[cell][q][2];
          const float4 h0 = h[cl][q];
          float4 t;
          t.x = h0.x + (w0.x * x0 + w1.x * y0 + w2.x * z0); t.y = h0.y + (w0.y * x0 + w1.y * y0 + w2.y * z0);
          t.z = h0.z + (w0.z * x0 + w1.z * y0 + w2.z * z0); t.w = h0.w + (w0.w * x0 + w1.w * y0 + w2.w * z0);
          acc.x += t.x * inv; acc.y += t.y * inv; acc.z += t.z * inv; acc.w += t.w * inv;
        }
      const int *so = sorted + (size_t)bi * n + s_st[cell];
      for (int q = 4; q < cv; ++q) {   // a cell with more than four points: the rest one by one (rare on the denoisers' levels)
        const int i0 = so[q], p0 = pp[i0];
        const float4 h0 = p0 >= 0 ? hb[(size_t)p0 * n27q + c4] : zero;
        const float x0 = px[i0], y0 = px[n + i0], z0 = px[2 * n + i0];
        float4 t;
        t.x = h0.x + (w0.x * x0 + w1.x * y0 + w2.x * z0); t.y = h0.y + (w0.y * x0 + w1.y * y0 + w2.y * z0);
        t.z = h0.z + (w0.z * x0 + w1.z * y0 + w2.z * z0); t.w = h0.w + (w0.w * x0 + w1.w * y0 + w2.w * z0);
        acc.x += t.x * inv; acc.y += t.y * inv; acc.z += t.z * inv; acc.w += t.w * inv;
      }
      reinterpret_cast<float4 *>(y + ((size_t)bi * n_max + k) * n27)[c4] = acc;
    }
  }
}

extern "C" int bdm_sparse_conv_rows_from_map(int b, int n, int r, int n_max, int n27, int hw, const float *hmap, const int *pix,
                                             const float *xyz, const float *wx, const int *cnt, const void *plan_workspace,
                                             const int *occ_list, const int *n_occ, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && r >= 1 && n_max >= 1 && n27 >= 4 && (n27 & 3) == 0 && n27 <= 1024 && hw >= 1,
              "sparse_conv_rows_from_map: bad sizes (n27 = %d must be a multiple of 4, <= 1024)", n27);
  BDM_REQUIRE((reinterpret_cast<size_t>(hmap) & 15) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0, "sparse_conv_rows_from_map: map and output must be 16-byte aligned");
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(const_cast<void *>(plan_workspace), b, n, r3);
  hipLaunchKernelGGL(sparse_rows_from_map_kernel, dim3(cdiv(n_max, 8), b), dim3(256), 0, (hipStream_t)stream, n, r3, n_max, n27, hw,
                     hmap, pix, xyz, wx, cnt, w.start, w.sorted, occ_list, n_occ, y);
  return launch_status("sparse_conv_rows_from_map");
}

// ---------------------------------------------------------------------------------------------------
// 3. output-stationary gather.  One workgroup = one grid row (x, y): r cells x Cout channels.
//    A wave sums one cell at a time over 64 channels (coalesced 256-B reads of a Y row segment); the tile is
//    transposed through LDS so that the channel-first output is written as contiguous z-runs.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sparse_gather_kernel(int cout, int r, int n_max, const float *__restrict__ y,
                                                            const int *__restrict__ occ_index,
                                                            const unsigned char *__restrict__ rowocc,
                                                            const float *__restrict__ bias, float *__restrict__ out) {
  extern __shared__ float tile[];  // [r][cout + 1] floats, then 9 * (r + 2) neighbour indices
  const int row = blockIdx.x, bi = blockIdx.y, x = row / r, yy = row % r;
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ldt = cout + 1, rs = r + 2;
  int *nbr = reinterpret_cast<int *>(tile + r * ldt);  // nbr[t9][1 + z], -1 = empty / outside the grid
  float *ob = out + (size_t)bi * cout * r3;
  const unsigned char *ro = rowocc + (size_t)bi * r2;
  bool any = false;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int gx = x + t9 / 3 - 1, gy = yy + t9 % 3 - 1;
    if (gx >= 0 && gx < r && gy >= 0 && gy < r) any |= ro[gx * r + gy] != 0;
  }
  if (!any) {  // no occupied cell anywhere under this row's 3x3x3 stencils: pure bias
    for (int e = tid; e < cout * r; e += 256) {
      const int co = e / r, z = e % r;
      ob[(size_t)co * r3 + row * r + z] = bias ? bias[co] : 0.f;
    }
    return;
  }
  const int *oi = occ_index + (size_t)bi * r3;
  for (int e = tid; e < 9 * rs; e += 256) {
    const int t9 = e / rs, zz = e % rs - 1;
    const int gx = x + t9 / 3 - 1, gy = yy + t9 % 3 - 1;
    int k = -1;
    if (gx >= 0 && gx < r && gy >= 0 && gy < r && zz >= 0 && zz < r && ro[gx * r + gy]) k = oi[(gx * r + gy) * r + zz];
    nbr[e] = k;
  }
  __syncthreads();
  const float *yb = y + (size_t)bi * n_max * 27 * cout;
  const int cblocks = (cout + 63) / 64;
  for (int item = wave; item < r * cblocks; item += 4) {
    const int z = item / cblocks, co = (item % cblocks) * 64 + lane;
    const bool cok = co < cout;
    // lane t < 27 looks up tap t's neighbour; the wave then walks the occupied taps in ascending order
    int kt = -1;
    if (lane < 27) kt = nbr[(lane / 3) * rs + 1 + z + lane % 3 - 1];
    unsigned long long mask = __ballot(kt >= 0);
    float acc = (cok && bias) ? bias[co] : 0.f;
    while (mask) {  // up to four independent row reads in flight
      int tp[4], kk[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tp[u] = mask ? __ffsll((long long)mask) - 1 : -1;
        if (mask) mask &= mask - 1;
        kk[u] = tp[u] >= 0 ? __shfl(kt, tp[u], 64) : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (kk[u] >= 0 && cok) ? yb[((size_t)kk[u] * 27 + tp[u]) * cout + co] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (cok) tile[z * ldt + co] = acc;
  }
  __syncthreads();
  for (int e = tid; e < cout * r; e += 256) {
    const int co = e / r, z = e % r;
    ob[(size_t)co * r3 + row * r + z] = tile[z * ldt + co];
  }
}

// Same sums in the same order, with more memory parallelism (Cout % 4 == 0): a lane owns (cell z, 4 channels) and
// reads its cell's occupied taps as 16-byte pieces of the Y rows, four loads in flight; a wave therefore works on
// 64 / (Cout/4) cells at once instead of one, and issues a quarter of the load instructions.
template <int GU>  // row reads in flight per lane
__global__ __launch_bounds__(256) void sparse_gather_v4_kernel(int cout, int r, int n_max, const float *__restrict__ y,
                                                               const int *__restrict__ occ_index,
                                                               const unsigned char *__restrict__ rowocc,
                                                               const float *__restrict__ bias, float *__restrict__ out,
                                                               int gn_cg, double *__restrict__ gn_partial) {
  extern __shared__ float tile[];  // [r][cout + 1] floats, 9 * (r + 2) neighbour indices, r tap masks, [256][2] GroupNorm partials
  // r and cout / 4 are powers of two on the denoisers' layers: shifts instead of integer divisions (~40 instructions each, a
  // visible share of a workgroup that owns only r x cout values); the general form stays for odd sizes
  const bool p2 = (r & (r - 1)) == 0 && ((cout >> 2) & ((cout >> 2) - 1)) == 0;
  const int rsh = __ffs(r) - 1, csh = __ffs(cout >> 2) - 1;
  const int row = blockIdx.x, bi = blockIdx.y, x = p2 ? row >> rsh : row / r, yy = p2 ? row & (r - 1) : row % r;
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x;
  const int ldt = cout + 1, rs = r + 2;
  int *nbr = reinterpret_cast<int *>(tile + r * ldt);  // nbr[t9][1 + z], -1 = empty / outside the grid
  unsigned *zmask = reinterpret_cast<unsigned *>(nbr + 9 * rs);
  float *red = reinterpret_cast<float *>(zmask + r);
  float *ob = out + (size_t)bi * cout * r3;
  // neighbour rows straight from occ_index (-1 for an empty cell): no dependent look at the row-occupancy bytes first, and the
  // "nothing under this row's stencils" test comes out of the same barrier
  const int *oi = occ_index + (size_t)bi * r3;
  int mine = 0;
  const float inv_rs = 1.0f / (float)rs;
  for (int e = tid; e < 9 * rs; e += 256) {
    const int t9 = (int)(((float)e + 0.5f) * inv_rs), zz = e - t9 * rs - 1;  // e / rs, e % rs - 1 (e < 9 * 34: exact)
    const int gx = x + t9 / 3 - 1, gy = yy + t9 % 3 - 1;
    int k = -1;
    if (gx >= 0 && gx < r && gy >= 0 && gy < r && zz >= 0 && zz < r) k = oi[(gx * r + gy) * r + zz];
    nbr[e] = k;
    mine |= k >= 0;
  }
  const int any = __syncthreads_or(mine);
  if (!any) {  // no occupied cell anywhere under this row's 3x3x3 stencils: pure bias
    for (int e = tid; e < cout * r; e += 256) {
      const int co = p2 ? e >> rsh : e / r, z = p2 ? e & (r - 1) : e % r;
      ob[(size_t)co * r3 + row * r + z] = bias ? bias[co] : 0.f;
    }
    if (gn_partial != nullptr && tid < cout / gn_cg) {  // this row's share of the GroupNorm statistics: r cells of bias
      double a = 0.0, q = 0.0;
      for (int j = 0; j < gn_cg; ++j) {
        const double bv = bias ? (double)bias[tid * gn_cg + j] : 0.0;
        a += bv; q += bv * bv;
      }
      double *dst = gn_partial + (((size_t)bi * (cout / gn_cg) + tid) * r2 + row) * 2;
      dst[0] = a * r;
      dst[1] = q * r;
    }
    return;
  }
  if (tid < r) {
    unsigned mk = 0u;
#pragma unroll
    for (int t = 0; t < 27; ++t) mk |= (nbr[(t / 3) * rs + 1 + tid + t % 3 - 1] >= 0 ? 1u : 0u) << t;
    zmask[tid] = mk;
  }
  __syncthreads();
  const float *yb = y + (size_t)bi * n_max * 27 * cout;
  const int c4n = cout >> 2;
  float gs = 0.f, gq = 0.f;  // GroupNorm partials of this thread's items
  for (int item = tid; item < r * c4n; item += 256) {
    const int z = p2 ? item >> csh : item / c4n, co = (p2 ? item & (c4n - 1) : item % c4n) * 4;
    unsigned mask = zmask[z];
    float4 acc = bias ? make_float4(bias[co], bias[co + 1], bias[co + 2], bias[co + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    while (mask) {
      float4 v[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask) {
          const int t = __ffs((int)mask) - 1;
          mask &= mask - 1;
          const int k = nbr[(t / 3) * rs + 1 + z + t % 3 - 1];
          v[u] = *reinterpret_cast<const float4 *>(yb + ((size_t)k * 27 + t) * cout + co);
        }
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    float *tp = tile + z * ldt + co;
    tp[0] = acc.x; tp[1] = acc.y; tp[2] = acc.z; tp[3] = acc.w;
    gs += (acc.x + acc.y) + (acc.z + acc.w);
    gq += (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
  }
  if (gn_partial != nullptr) {
    // GroupNorm(cout / gn_cg groups) statistics of this row of the output.  A thread's items all lie in ONE group (256 is a
    // multiple of cout / 4 and gn_cg of 4).  Lane l of a wave holds channel quad l % c4n: butterflies over the lane bits that
    // enumerate copies of a quad (>= c4n) and quads of one group (< gn_cg / 4) leave each group's wave sum in its first lane;
    // the four waves are then added in order, in fp64 -- a fixed order, deterministic.
    const int lane = tid & 63, wave = tid >> 6, q4 = gn_cg >> 2;
    for (int o = c4n; o < 64; o <<= 1) { gs += __shfl_xor(gs, o, 64); gq += __shfl_xor(gq, o, 64); }
    for (int o = 1; o < q4 && o < 64; o <<= 1) { gs += __shfl_xor(gs, o, 64); gq += __shfl_xor(gq, o, 64); }
    const int lq = lane % c4n;
    if (lane < c4n && (lq % q4) == 0) { red[(wave * 64 + lq / q4) * 2] = gs; red[(wave * 64 + lq / q4) * 2 + 1] = gq; }
  }
  __syncthreads();
  for (int e = tid; e < cout * r; e += 256) {
    const int co = p2 ? e >> rsh : e / r, z = p2 ? e & (r - 1) : e % r;
    ob[(size_t)co * r3 + row * r + z] = tile[z * ldt + co];
  }
  if (gn_partial != nullptr && tid < cout / gn_cg) {
    double a = 0.0, q = 0.0;
    for (int w = 0; w < 4; ++w) { a += (double)red[(w * 64 + tid) * 2]; q += (double)red[(w * 64 + tid) * 2 + 1]; }
    double *dst = gn_partial + (((size_t)bi * (cout / gn_cg) + tid) * r2 + row) * 2;
    dst[0] = a;
    dst[1] = q;
  }
}

static int sparse_gather_launch(int b, int cout, int r, int n_max, const float *y, const int *occ_index,
                                const unsigned char *rowocc, const float *bias, float *out, int gn_cg, double *gn_partial,
                                void *stream) {
  BDM_REQUIRE(b >= 0 && cout >= 1 && r >= 1 && n_max >= 1, "sparse_conv_gather: bad sizes");
  if (b == 0) return BDM_OK;
  const size_t smem = sizeof(float) * (size_t)r * (cout + 1) + sizeof(int) * 9 * (size_t)(r + 2);
  if ((cout & 3) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0) {
    const int gu = 8;  // row reads in flight per lane (4 measured 1.5 % slower)
    const size_t sm4 = smem + sizeof(unsigned) * r + sizeof(float) * 512;
    if (gu == 8)
      hipLaunchKernelGGL(sparse_gather_v4_kernel<8>, dim3(r * r, b), dim3(256), sm4, (hipStream_t)stream, cout, r, n_max, y,
                         occ_index, rowocc, bias, out, gn_cg, gn_partial);
    else
      hipLaunchKernelGGL(sparse_gather_v4_kernel<4>, dim3(r * r, b), dim3(256), sm4, (hipStream_t)stream, cout, r, n_max, y,
                         occ_index, rowocc, bias, out, gn_cg, gn_partial);
    return launch_status("sparse_conv_gather");
  }
  BDM_REQUIRE(gn_partial == nullptr, "sparse_conv_gather_gn: needs cout %% 4 == 0 and a 16-byte aligned intermediate");
  hipLaunchKernelGGL(sparse_gather_kernel, dim3(r * r, b), dim3(256), smem, (hipStream_t)stream, cout, r, n_max, y,
                     occ_index, rowocc, bias, out);
  return launch_status("sparse_conv_gather");
}

extern "C" int bdm_sparse_conv_gather(int b, int cout, int r, int n_max, const float *y, const int *occ_index,
                                      const unsigned char *rowocc, const float *bias, float *out, void *stream) {
  return sparse_gather_launch(b, cout, r, n_max, y, occ_index, rowocc, bias, out, 0, nullptr, stream);
}

// The same gather, also leaving the GroupNorm(groups) statistics of its OUTPUT: r*r slice partials per (shape, group) in
// gn_partial (b, groups, r*r, 2 doubles) -- slice = grid row (x, y).  bdm_group_norm_to_h2_stats consumes them, so the first
// GroupNorm of a PVConv (pvconv.py:80) needs no statistics pass over the grid.
extern "C" int bdm_sparse_conv_gather_gn(int b, int cout, int r, int n_max, const float *y, const int *occ_index,
                                         const unsigned char *rowocc, const float *bias, float *out, int groups,
                                         void *gn_partial, void *stream) {
  const int cg = groups >= 1 && cout % groups == 0 ? cout / groups : 0;
  BDM_REQUIRE(gn_partial != nullptr && cg >= 4 && cg % 4 == 0 && (cout & 3) == 0 && 256 % (cout / 4) == 0 && groups <= 64,
              "sparse_conv_gather_gn: needs 4 | channels per group and (cout / 4) | 256 (cout=%d groups=%d)", cout, groups);
  return sparse_gather_launch(b, cout, r, n_max, y, occ_index, rowocc, bias, out, cg, (double *)gn_partial, stream);
}
