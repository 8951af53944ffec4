// sparse_conv_h2.hip -- fp16x3 form of the sparse first convolution's GEMM (the default on the 8^3 levels and the wide 16^3 layers):
// occupied cells' mean features as fp32 records + per-SHAPE max |value| (bdm_sparse_voxel_features_f32), their two-term fp16 split
// with a power-of-two scale per shape (bdm_sparse_split_h2), weights with a per-output-channel scale
// (bdm_sparse_conv_pack_weights_h2) and the batched GEMM Y = X . W on v_mfma_f32_32x32x16_f16 (bdm_sparse_conv_gemm_h2).
// The gather that follows is sparse_conv.hip's.  Arithmetic: an fp32 operand times a power of two is stored as hi + lo (two fp16
// terms, 22 signed bits), a product is lo.hi + hi.lo + hi.hi accumulated in fp32: fp32-grade (<= 3e-7 relative L2 vs fp64).
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

#include "sparse_h2_common.h"

// LDS-cached feature gather shared with sparse_conv.hip (defined there)
bool bdm_sparse_features_lds_launch(int out_kind, int b, int c, int n, int r3, int n_max, const float *features, long long bs_f,
                                    int ld_f, const int *cnt, const int *start, const int *sorted, const int *occ_list,
                                    const int *n_occ, void *out, unsigned *amax, hipStream_t stream, int *rc);

// ---------------------------------------------------------------------------------------------------
// occupied cells' mean features as fp32 records: xr (B, G, n_max) records of 8 channels (32 bytes), rows >= n_occ zero;
// amax[bi] = max |value| of shape bi (b slots, zero on entry).  Same XCD-aware unit decoding and the same
// summation order as sparse_vox_features_s3_kernel (sparse_conv.hip): values equal the dense voxel grid's bit for bit.
// ---------------------------------------------------------------------------------------------------
__global__ void sparse_vox_features_f32_kernel(int c, int n, int r3, int n_max, int G, int units, int kblocks,
                                               const float *__restrict__ feat, long long bs_f, int ld_f,
                                               const int *__restrict__ cnt, const int *__restrict__ start,
                                               const int *__restrict__ sorted, const int *__restrict__ occ_list,
                                               const int *__restrict__ n_occ, float4 *__restrict__ xr,
                                               unsigned *__restrict__ amax) {
#pragma clang fp contract(off)
  const int wg = blockIdx.x, span = 8 * kblocks;
  const int unit = (wg / span) * 8 + (wg % span) % 8, kb = (wg % span) / 8;
  if (unit >= units) return;
  const int bi = unit / G, g = unit % G;
  const int k = kb * blockDim.x + threadIdx.x;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (k < n_max && k < n_occ[bi]) {
    const int v = occ_list[(size_t)bi * n_max + k];
    const int cv = cnt[(size_t)bi * r3 + v];
    const int *so = sorted + (size_t)bi * n + start[(size_t)bi * r3 + v];
    const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
    const float *fb = feat + (size_t)bi * bs_f + (size_t)g * 8 * ld_f;
    const int nch = min(8, c - g * 8);
    for (int q0 = 0; q0 < cv; q0 += 4) {  // four points at a time, independent loads, list order kept (see sparse_conv.hip)
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
  }
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(acc[j]));
  if (k < n_max) {
    float4 *o = xr + (((size_t)bi * G + g) * n_max + k) * 2;
    o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax + bi, __float_as_uint(m));  // per shape; non-negative floats order as unsigned ints
}

extern "C" int bdm_sparse_voxel_features_f32(int b, int c, int n, int r, int n_max, const float *features, long long bs_f,
                                             int ld_f, const int *cnt, const void *plan_workspace, const int *occ_list,
                                             const int *n_occ, void *xr, float *amax, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && n >= 1 && r >= 1 && n_max >= 1 && amax != nullptr, "sparse_voxel_features_f32: bad arguments");
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(const_cast<void *>(plan_workspace), b, n, r3);
  int rc_lds = BDM_OK;
  if (bdm_sparse_features_lds_launch(1, b, c, n, r3, n_max, features, bs_f, ld_f, cnt, w.start, w.sorted, occ_list, n_occ, xr,
                                     (unsigned *)amax, (hipStream_t)stream, &rc_lds))
    return launch_status("sparse_voxel_features_f32");
  const int G = (c + 7) / 8, units = b * G, kblocks = cdiv(n_max, 128);
  hipLaunchKernelGGL(sparse_vox_features_f32_kernel, dim3(cdiv(units, 8) * 8 * kblocks), dim3(128), 0, (hipStream_t)stream, c,
                     n, r3, n_max, G, units, kblocks, features, bs_f, ld_f, cnt, w.start, w.sorted, occ_list, n_occ,
                     (float4 *)xr, (unsigned *)amax);
  return launch_status("sparse_voxel_features_f32");
}

// ---------------------------------------------------------------------------------------------------
// fp16x3 form of the batched GEMM of sparse_conv.hip (step 2): Y[b] (n_occ x 27*Cout) = X[b] . W with HALF the matrix work
// of the bf16x6 GEMM.  A = the fp32 feature records of bdm_sparse_voxel_features_f32, scaled by the power of two derived
// from amax and split into (hi, lo) fp16 while they are staged into LDS; B = weights packed as [G][2][27*Cout] fp16 records
// with a per-output-channel scale (column n -> channel n % Cout).  Tile BM x 128 (BM = 128, or 64 for levels with <= 256
// occupied rows per shape: less padding), K = 32 per stage, register-prefetched.  The gather (step 3) is unchanged.
// ---------------------------------------------------------------------------------------------------
__global__ void sparse_gemm_h2_pack_kernel(int cout, int cin, const float *__restrict__ w, const float *__restrict__ scale,
                                           unsigned short *__restrict__ wq) {
  const int G = (cin + 7) / 8, n27 = 27 * cout;
  const long long total = (long long)G * n27;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int col = (int)(e % n27), g = (int)(e / n27), tap = col / cout, co = col % cout;
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = g * 8 + j;
      split2s(ci < cin ? w[((size_t)co * cin + ci) * 27 + tap] * scale[co] : 0.f, h[j], l[j]);
    }
    unsigned short *ph = wq + (((size_t)g * 2 + 0) * n27 + col) * 8, *pl = wq + (((size_t)g * 2 + 1) * n27 + col) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ph[j] = h[j]; pl[j] = l[j]; }
  }
}
extern "C" size_t bdm_sparse_conv_h2_weight_elems(int cout, int cin) { return (size_t)((cin + 7) / 8) * 2 * 27 * cout * 8; }
extern "C" int bdm_sparse_conv_pack_weights_h2(int cout, int cin, const float *w, void *packed, float *scale_ws,
                                               float *inv_scale, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1 && scale_ws != nullptr && inv_scale != nullptr, "sparse_conv_pack_weights_h2: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sparse_fused_weight_scale_kernel, dim3(cout), dim3(256), 0, s, cout, cin, w, scale_ws, inv_scale);
  hipLaunchKernelGGL(sparse_gemm_h2_pack_kernel, dim3(512), dim3(256), 0, s, cout, cin, w, scale_ws, (unsigned short *)packed);
  return launch_status("sparse_conv_pack_weights_h2");
}

// fp32 records -> (hi, lo) fp16 records of x * 2^e (e from amax), layout [b][g][2][M]: done ONCE per call, so that the
// GEMM's column tiles do not each repeat the split while staging
__global__ void sparse_split_h2_kernel(long long rows_total, int M, const float4 *__restrict__ xr, const float *__restrict__ amax,
                                       uint4 *__restrict__ xh, int G) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < rows_total; e += (long long)gridDim.x * blockDim.x) {
    const long long bg = e / M;
    const int row = (int)(e % M);
    const float sx = act_scale_from_max(amax[bg / G]);  // one power-of-two scale per SHAPE: results do not depend on batch-mates
    f16x8 hi, lo;
    split_record(xr[e * 2], xr[e * 2 + 1], sx, hi, lo);
    xh[(bg * 2 + 0) * M + row] = *reinterpret_cast<const uint4 *>(&hi);
    xh[(bg * 2 + 1) * M + row] = *reinterpret_cast<const uint4 *>(&lo);
  }
}

template <int BM>
__global__ __launch_bounds__(256) void sparse_gemm_h2_kernel(int M, int G, int N, int Cout, const uint4 *__restrict__ A,
                                                             const float *__restrict__ amax, const uint4 *__restrict__ Bw,
                                                             const float *__restrict__ inv_sw, const int *__restrict__ m_count,
                                                             const float *__restrict__ col_bias, long long bs_cb, float *__restrict__ Y) {
  constexpr int BN = 128, MX = BM / 64, BI = 8 * BN / 256;  // 4 groups x 2 splits per stage
  __shared__ uint4 As[8 * BM], Bs[8 * BN];  // [group-in-stage][split][row]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;  // 2 x 2 waves: rows wr * (BM/2), columns wc * 64
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, bi = blockIdx.z;
  if (m_count && m0 >= m_count[bi]) return;  // rows beyond this shape's occupied cells
  const float sx = act_scale_from_max(amax[bi]);
  const uint4 *Ab = A + (size_t)bi * G * 2 * M;
  f32x16 acc[MX][2];
#pragma unroll
  for (int x = 0; x < MX; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
  // the per-shape column addend of this lane's two columns, requested BEFORE the K loop (read in the epilogue it was one more dependent
  // round trip between the last MFMA and the stores of a workgroup that lives for two K stages)
  float cbv[2] = {0.f, 0.f}, postv[2];   // (likewise the per-column weight scales)
#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int nn = min(n0 + (wc * 2 + y) * 32 + li, N - 1);
    postv[y] = inv_sw[nn % Cout];
    if (col_bias != nullptr) cbv[y] = col_bias[(size_t)bi * bs_cb + nn];
  }
  constexpr int AI = 8 * BM / 256;
  // native-vector staging registers, loads without branch or select (see sparse_gemm_s3_kernel): rows / columns beyond the
  // matrix read a clamped address, channel groups >= G are zeroed on the way to LDS
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v ar[AI], br[BI];
  auto load_stage = [&](int g0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256, row = e % BM, gs = e / BM, g = min(g0 + gs / 2, G - 1), sp = gs % 2;
      ar[i] = *reinterpret_cast<const u32x4v *>(&Ab[(unsigned)((g * 2 + sp) * M + min(m0 + row, M - 1))]);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e = tid + i * 256, col = e % BN, gs = e / BN, g = min(g0 + gs / 2, G - 1), sp = gs % 2;
      br[i] = *reinterpret_cast<const u32x4v *>(&Bw[(unsigned)((g * 2 + sp) * N + min(n0 + col, N - 1))]);
    }
  };
  load_stage(0);
  for (int g0 = 0; g0 < G; g0 += 4) {
    __syncthreads();
    const u32x4v zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256, gs = e / BM;
      *reinterpret_cast<u32x4v *>(&As[e]) = (g0 + gs / 2 < G) ? ar[i] : zero;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e = tid + i * 256, gs = e / BN;
      *reinterpret_cast<u32x4v *>(&Bs[e]) = (g0 + gs / 2 < G) ? br[i] : zero;
    }
    __syncthreads();
    if (g0 + 4 < G) load_stage(g0 + 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 a[MX][2], b[2][2];
#pragma unroll
      for (int x = 0; x < MX; ++x)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
          a[x][sp] = *reinterpret_cast<const f16x8 *>(&As[((2 * kk + lh) * 2 + sp) * BM + (wr * MX + x) * 32 + li]);
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
          b[y][sp] = *reinterpret_cast<const f16x8 *>(&Bs[((2 * kk + lh) * 2 + sp) * BN + (wc * 2 + y) * 32 + li]);
      // term-major over the independent accumulators: lo.hi, hi.lo, hi.hi
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int x = 0; x < MX; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[x][term == 0 ? 1 : 0], b[y][term == 1 ? 1 : 0], acc[x][y], 0, 0, 0);
    }
  }
  float *Yb = Y + (size_t)bi * M * N;
  const float inv_sx = 1.0f / sx;
#pragma unroll
  for (int x = 0; x < MX; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) {
      const int nn = n0 + (wc * 2 + y) * 32 + li;
      const float post = postv[y] * inv_sx;
      const float cb = cbv[y];  // per-shape column addend (see the _cb entry point)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wr * MX + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && nn < N) Yb[(size_t)m * N + nn] = acc[x][y][r] * post + cb;
      }
    }
}

// xr (b, G, n_max) fp32 records + amax -> xh (b, G, 2, n_max) fp16 records (hi, lo of x * 2^e)
extern "C" int bdm_sparse_split_h2(int b, int cin, int n_max, const void *xr, const float *amax, void *xh, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && n_max >= 1 && amax != nullptr, "sparse_split_h2: bad arguments");
  if (b == 0) return BDM_OK;
  const long long rows = (long long)b * ((cin + 7) / 8) * n_max;
  long long grid = (rows + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(sparse_split_h2_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, rows, n_max, (const float4 *)xr,
                     amax, (uint4 *)xh, (cin + 7) / 8);
  return launch_status("sparse_split_h2");
}

// col_bias (b, 27 * cout) or null: a per-shape addend of every row's column (tap, co) -- the share of input channels that are CONSTANT over
// a shape's occupied cells (the time embedding concatenated to the features, pvcnn.py:103: avg_voxelize of a constant is the constant on
// every occupied cell), u[b][tap][co] = sum_c W[co][c][tap] t[b][c]; the gather then adds it once per OCCUPIED neighbour, which is what
// the zero-padded convolution of the concatenated grid does.
static int sparse_gemm_h2_launch(int b, int n_max, int cin, int cout, const void *xh, const float *amax, const void *packed_w,
                                 const float *inv_scale, const int *n_occ, const float *col_bias, long long bs_cb, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && n_max >= 1 && cin >= 1 && cout >= 1 && amax != nullptr && inv_scale != nullptr, "sparse_conv_gemm_h2: bad arguments");
  if (b == 0) return BDM_OK;
  const int n27 = 27 * cout, G = (cin + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  if (n_max <= 256)
    hipLaunchKernelGGL(sparse_gemm_h2_kernel<64>, dim3(cdiv(n27, 128), cdiv(n_max, 64), b), dim3(256), 0, s, n_max, G, n27, cout,
                       (const uint4 *)xh, amax, (const uint4 *)packed_w, inv_scale, n_occ, col_bias, bs_cb, y);
  else
    hipLaunchKernelGGL(sparse_gemm_h2_kernel<128>, dim3(cdiv(n27, 128), cdiv(n_max, 128), b), dim3(256), 0, s, n_max, G, n27, cout,
                       (const uint4 *)xh, amax, (const uint4 *)packed_w, inv_scale, n_occ, col_bias, bs_cb, y);
  return launch_status("sparse_conv_gemm_h2");
}

extern "C" int bdm_sparse_conv_gemm_h2(int b, int n_max, int cin, int cout, const void *xh, const float *amax, const void *packed_w,
                                       const float *inv_scale, const int *n_occ, float *y, void *stream) {
  return sparse_gemm_h2_launch(b, n_max, cin, cout, xh, amax, packed_w, inv_scale, n_occ, nullptr, 0, y, stream);
}

extern "C" int bdm_sparse_conv_gemm_h2_cb(int b, int n_max, int cin, int cout, const void *xh, const float *amax, const void *packed_w,
                                          const float *inv_scale, const int *n_occ, const float *col_bias, long long bs_cb, float *y, void *stream) {
  BDM_REQUIRE(col_bias != nullptr && bs_cb >= 27LL * cout, "sparse_conv_gemm_h2_cb: col_bias is null or its batch stride < 27 * cout");
  return sparse_gemm_h2_launch(b, n_max, cin, cout, xh, amax, packed_w, inv_scale, n_occ, col_bias, bs_cb, y, stream);
}
