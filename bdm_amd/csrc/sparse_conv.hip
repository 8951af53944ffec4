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
#include "../../include/bdm_hip.h"
#include "common.h"

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

extern "C" int bdm_sparse_conv_gather(int b, int cout, int r, int n_max, const float *y, const int *occ_index,
                                      const unsigned char *rowocc, const float *bias, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && cout >= 1 && r >= 1 && n_max >= 1, "sparse_conv_gather: bad sizes");
  if (b == 0) return BDM_OK;
  const size_t smem = sizeof(float) * (size_t)r * (cout + 1) + sizeof(int) * 9 * (size_t)(r + 2);
  hipLaunchKernelGGL(sparse_gather_kernel, dim3(r * r, b), dim3(256), smem, (hipStream_t)stream, cout, r, n_max, y,
                     occ_index, rowocc, bias, out);
  return launch_status("sparse_conv_gather");
}
