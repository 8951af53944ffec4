// conv3d_s3.hip -- the 3x3x3 voxel convolution at fp32 accuracy on the BF16 matrix cores ("bf16x6").
//
// gfx950's f32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md, Matrix cores) and the dense
// fp32 kernel of conv3d.hip already sits at the sustained-clock MFMA limit (~120 TFLOP/s).  Here every fp32
// operand x is split EXACTLY into three bf16 terms  x = x1 + x2 + x3  (8 + 8 + 8 mantissa bits) and the product
// a*b is evaluated as the six leading partial products
//       a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2)
// each of which is exact in fp32 (8 x 8 bits) and accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The three
// dropped terms are <= 2^-23 |a b| (below one fp32 rounding of the product), so the result has fp32-level
// accuracy (measured <= 3e-7 relative L2 vs fp64, same as the fp32-MFMA kernel) at 16/6 = 2.7x its matrix rate.
//
// Layouts ("S3" = split-in-3):
//   activations  (B, ceil(C/8), 3, r^3, 8) bf16 : per (shape, 8-channel group, split) a run of 16-byte voxel
//                records -> a halo row is one contiguous copy, and a lane's B fragment (8 consecutive k =
//                8 channels of ONE voxel) is a single conflict-free ds_read_b128.  Produced directly by the
//                voxeliser (bdm_avg_voxelize_s3) and by the GroupNorm+Swish kernel (bdm_group_norm_to_s3).
//   weights      [ceil(Cin/8)][14 tap pairs][3 splits][2][Cout][8 ci] bf16 : the K = 16 of one MFMA is
//                (2 taps) x (8 channels); lane half h addresses tap 2p+h, so both operands stay 16-byte reads.
//                Tap 27 is a zero pad.
// Output: fp32, channel-first (B, Cout, r^3) -- what GroupNorm statistics, attention, SE and devoxelisation read.
#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;
#include "s3_split.h"

#define S3_PAIRS 14

// ---------------------------------------------------------------------------------------------------
// weight packing: (Cout, Cin, 3,3,3) fp32 -> [C8][14][3][2][Cout][8] bf16
// ---------------------------------------------------------------------------------------------------
__global__ void pack_s3_kernel(int cout, int cin, const float *__restrict__ w, unsigned short *__restrict__ wq) {
  const int c8n = (cin + 7) / 8;
  const long long total = (long long)c8n * S3_PAIRS * 2 * cout * 8;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e % 8);
    const int co = (int)((e / 8) % cout);
    const int h = (int)((e / (8ll * cout)) % 2);
    const int p = (int)((e / (16ll * cout)) % S3_PAIRS);
    const int c8 = (int)(e / (16ll * cout * S3_PAIRS));
    const int ci = c8 * 8 + j, tap = 2 * p + h;
    const float v = (ci < cin && tap < 27) ? w[((size_t)co * cin + ci) * 27 + tap] : 0.f;
    unsigned short s[3];
    split3(v, s[0], s[1], s[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) wq[((((size_t)(c8 * S3_PAIRS + p) * 3 + q) * 2 + h) * cout + co) * 8 + j] = s[q];
  }
}
extern "C" size_t bdm_conv3d_s3_weight_elems(int cout, int cin) {
  return (size_t)((cin + 7) / 8) * S3_PAIRS * 3 * 2 * cout * 8;
}
extern "C" int bdm_conv3d_s3_pack_weights(int cout, int cin, const float *w, void *packed, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1, "conv3d_s3_pack_weights: bad sizes");
  hipLaunchKernelGGL(pack_s3_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, cout, cin, w, (unsigned short *)packed);
  return launch_status("conv3d_s3_pack_weights");
}

// ---------------------------------------------------------------------------------------------------
// the convolution
// ---------------------------------------------------------------------------------------------------
template <int MI, int NI, int R, int TX, int TY>
__global__ __launch_bounds__(256) void conv3d_s3_kernel(int C8, int Cout, const float4 *__restrict__ x,
                                                        const float4 *__restrict__ wq, const float *__restrict__ bias,
                                                        float *__restrict__ y) {
  extern __shared__ __align__(16) float4 smem4[];
  constexpr int BM = 32 * MI;
  constexpr int RSV = R + 2;                 // voxel records per halo row (one zero pad at each end)
  constexpr int ROWS = (TX + 2) * (TY + 2);
  constexpr int HALO = ROWS * RSV;           // records per split
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int XV = 3 * ROWS * R, XI = (XV + 255) / 256;          // 16-byte pieces of the input tile
  constexpr int WV = S3_PAIRS * 3 * 2 * BM, WI = (WV + 255) / 256;  // 16-byte pieces of the weight tile
  float4 *Xs = smem4;              // [3][HALO]
  float4 *Ws = smem4 + 3 * HALO;   // [14][3][2][BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  constexpr int tiles_y = R / TY;
  const int X0 = (blockIdx.x / tiles_y) * TX, Y0 = (blockIdx.x % tiles_y) * TY;
  const int m0 = blockIdx.y * BM, bi = blockIdx.z;
  const float4 *xb = x + (size_t)bi * C8 * 3 * R3;
  float *yb = y + (size_t)bi * Cout * R3;

  constexpr int rpb = 32 / R;
  const int dyl = li / R, zl = li % R;
  constexpr int blocks_per_plane = TY / rpb;
  int lbase[NI], gvox[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int nb = q * 4 + wave;
    const int tx = nb / blocks_per_plane, ty = (nb % blocks_per_plane) * rpb + dyl;
    lbase[q] = ((tx + 1) * (TY + 2) + (ty + 1)) * RSV + 1 + zl;
    gvox[q] = ((X0 + tx) * R + (Y0 + ty)) * R + zl;
  }
  // this lane half's tap of pair p is 2p + lh; record offset of that tap (pad tap 27 reuses tap 26's address)
  f32x16 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int q = 0; q < NI; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][q][i] = 0.f;

  for (int e = tid; e < 3 * HALO; e += 256) Xs[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 xr[XI], wr[WI];
  auto load_chunk = [&](int c8) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + i * 256;
      const int z = e % R, row = (e / R) % ROWS, s = e / (R * ROWS);
      const int gx = X0 + row / (TY + 2) - 1, gy = Y0 + row % (TY + 2) - 1;
      xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < XV && gx >= 0 && gx < R && gy >= 0 && gy < R)
        xr[i] = xb[((size_t)c8 * 3 + s) * R3 + (gx * R + gy) * R + z];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * 256;
      const int m = e % BM, psh = e / BM;  // psh = (p*3 + s)*2 + h
      wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < WV && m0 + m < Cout) wr[i] = wq[((size_t)c8 * (S3_PAIRS * 6) + psh) * Cout + m0 + m];
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + i * 256;
      const int z = e % R, row = (e / R) % ROWS, s = e / (R * ROWS);
      const int gx = X0 + row / (TY + 2) - 1, gy = Y0 + row % (TY + 2) - 1;
      if (e < XV && gx >= 0 && gx < R && gy >= 0 && gy < R) Xs[s * HALO + row * RSV + 1 + z] = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * 256;
      if (e < WV) Ws[e] = wr[i];
    }
  };

  load_chunk(0);
  for (int c8 = 0; c8 < C8; ++c8) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (c8 + 1 < C8) load_chunk(c8 + 1);
#pragma unroll
    for (int p = 0; p < S3_PAIRS; ++p) {
      // tap of this lane half: 2p + lh  (compile-time pair, run-time half -> select between two constants)
      const int t0 = 2 * p, t1 = (2 * p + 1 < 27) ? 2 * p + 1 : 26;
      const int off0 = ((t0 / 9 - 1) * (TY + 2) + ((t0 / 3) % 3 - 1)) * RSV + (t0 % 3 - 1);
      const int off1 = ((t1 / 9 - 1) * (TY + 2) + ((t1 / 3) % 3 - 1)) * RSV + (t1 % 3 - 1);
      const int toff = lh ? off1 : off0;
      bf16x8 a[MI][3], b[NI][3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float4 t = Ws[((p * 3 + s) * 2 + lh) * BM + mi * 32 + li];
          a[mi][s] = *reinterpret_cast<const bf16x8 *>(&t);
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
          const float4 t = Xs[s * HALO + lbase[q] + toff];
          b[q][s] = *reinterpret_cast<const bf16x8 *>(&t);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < NI; ++q) {
          f32x16 c = acc[mi][q];
          // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[q][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[q][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[q][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[q][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[q][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[q][0], c, 0, 0, 0);
          acc[mi][q] = c;
        }
    }
  }
#pragma unroll
  for (int p = 0; p < MI; ++p)
#pragma unroll
    for (int q = 0; q < NI; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + p * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (m < Cout) yb[(size_t)m * R3 + gvox[q]] = acc[p][q][i] + (bias ? bias[m] : 0.f);
      }
}

extern "C" int bdm_conv3d_3x3x3_s3(int b, int cin, int cout, int r, const void *x_s3, const void *packed_w,
                                   const float *bias, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1, "conv3d_s3: bad sizes");
  if (r != 8 && r != 16 && r != 32) {
    set_error("conv3d_s3: resolution %d unsupported (8, 16, 32)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const int c8 = (cin + 7) / 8;
  int tx, ty, ni, mi;
  if (r == 32) { tx = 2; ty = 8; ni = 4; }
  else if (r == 16) { tx = 2; ty = 16; ni = 4; }
  else { tx = 4; ty = 8; ni = 2; }
  mi = (cout > 32 && ni == 4) ? 2 : 1;
  const size_t smem = 16 * ((size_t)3 * (tx + 2) * (ty + 2) * (r + 2) + (size_t)S3_PAIRS * 6 * 32 * mi);
  dim3 grid((r / tx) * (r / ty), cdiv(cout, 32 * mi), b);
  hipStream_t s = (hipStream_t)stream;
#define S3_LAUNCH(MI, NI, R, TX, TY)                                                                            \
  do {                                                                                                          \
    BDM_ALLOW_LDS((conv3d_s3_kernel<MI, NI, R, TX, TY>), smem);                                                 \
    hipLaunchKernelGGL((conv3d_s3_kernel<MI, NI, R, TX, TY>), grid, dim3(256), smem, s, c8, cout,               \
                       (const float4 *)x_s3, (const float4 *)packed_w, bias, y);                                \
  } while (0)
  if (r == 32) { if (mi == 2) S3_LAUNCH(2, 4, 32, 2, 8); else S3_LAUNCH(1, 4, 32, 2, 8); }
  else if (r == 16) { if (mi == 2) S3_LAUNCH(2, 4, 16, 2, 16); else S3_LAUNCH(1, 4, 16, 2, 16); }
  else S3_LAUNCH(1, 2, 8, 4, 8);
#undef S3_LAUNCH
  return launch_status("conv3d_s3");
}

// ---------------------------------------------------------------------------------------------------
// producers of the S3 layout
// ---------------------------------------------------------------------------------------------------
// fp32 channel-first (B, C, V)  ->  S3, optionally through GroupNorm (+ Swish): the fused normalise step between
// the two convolutions of a PVConv (pvconv.py:78-82).  stats = per-(shape, group) fp64 (sum, sumsq) slice partials.
__global__ void to_s3_kernel(int C, int V, int G, int S, const float *__restrict__ x, const double *__restrict__ partial,
                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int act,
                             unsigned short *__restrict__ out) {
  __shared__ float s_mean[64], s_rstd[64];
  const int bi = blockIdx.z, c8 = blockIdx.y, C8 = gridDim.y;
  const int cg = C / G;
  if (partial) {
    if (threadIdx.x < G) {
      double a = 0.0, q = 0.0;
      const size_t bg = (size_t)bi * G + threadIdx.x;
      for (int s = 0; s < S; ++s) { a += partial[(bg * S + s) * 2]; q += partial[(bg * S + s) * 2 + 1]; }
      const double cnt = (double)cg * V, mean = a / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0) var = 0;
      s_mean[threadIdx.x] = (float)mean;
      s_rstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
  }
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  float val[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = c8 * 8 + j;
    float t = 0.f;
    if (ch < C) {
      t = x[((size_t)bi * C + ch) * V + v];
      if (partial) {
        const int g = ch / cg;
        t = (t - s_mean[g]) * s_rstd[g] * gamma[ch] + beta[ch];
        if (act == 1) t = swishf(t);
      }
    }
    val[j] = t;
  }
  store_s3(out + ((size_t)bi * C8 + c8) * 3 * (size_t)V * 8, (size_t)v, (size_t)V, val);
}

// stats kernel shared with dense_ops.hip's GroupNorm (declared there)
extern "C" int bdm_group_norm_stats(int b, int c, int l, int groups, const float *x, long long bs_x, void *workspace,
                                    int *slices_out, void *stream);

extern "C" int bdm_group_norm_to_s3(int b, int c, int v, int groups, const float *x, const float *gamma,
                                    const float *beta, float eps, int act, void *out_s3, void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && v >= 1, "group_norm_to_s3: bad sizes");
  BDM_REQUIRE(groups == 0 || (c % groups == 0 && groups <= 64 && workspace != nullptr), "group_norm_to_s3: bad groups %d", groups);
  if (b == 0) return BDM_OK;
  int S = 0;
  if (groups > 0) {
    int rc = bdm_group_norm_stats(b, c, v, groups, x, (long long)c * v, workspace, &S, stream);
    if (rc) return rc;
  }
  dim3 grid(cdiv(v, 256), (c + 7) / 8, b);
  hipLaunchKernelGGL(to_s3_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, v, groups > 0 ? groups : 1, S, x,
                     groups > 0 ? (const double *)workspace : nullptr, gamma, beta, eps, act, (unsigned short *)out_s3);
  return launch_status("group_norm_to_s3");
}

// voxeliser writing S3 directly (same deterministic per-voxel order as bdm_avg_voxelize_forward); uses the plan
// (cnt, start, sorted) that bdm_avg_voxelize_plan left in the workspace.
__global__ void vox_reduce_s3_kernel(int c, int n, int r3, const float *__restrict__ feat, long long bs_f, int ld_f,
                                     const int *__restrict__ cnt, const int *__restrict__ start,
                                     const int *__restrict__ sorted, unsigned short *__restrict__ out) {
#pragma clang fp contract(off)
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z, c8 = blockIdx.y, C8 = gridDim.y;
  if (v >= r3) return;
  const int cv = cnt[(size_t)bi * r3 + v];
  const int s = start[(size_t)bi * r3 + v];
  const int *so = sorted + (size_t)bi * n + s;
  const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  const float *fb = feat + (size_t)bi * bs_f;
  for (int q = 0; q < cv; ++q) {
    const int p = so[q];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = c8 * 8 + j;
      if (ch < c) acc[j] = acc[j] + fb[(size_t)ch * ld_f + p] * inv;
    }
  }
  store_s3(out + ((size_t)bi * C8 + c8) * 3 * (size_t)r3 * 8, (size_t)v, (size_t)r3, acc);
}

extern "C" int bdm_voxelize_plan(int b, int n, int r, const int *coords, int *ind, int *cnt, void *workspace, void *stream);

extern "C" int bdm_avg_voxelize_s3(int b, int c, int n, int r, const float *features, long long bs_f, int ld_f,
                                   const int *coords, void *out_s3, int *ind, int *cnt, void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && n >= 1 && r >= 1 && r <= 32, "avg_voxelize_s3: bad sizes");
  BDM_REQUIRE(workspace != nullptr, "avg_voxelize_s3: workspace is NULL");
  if (b == 0) return BDM_OK;
  int rc = bdm_voxelize_plan(b, n, r, coords, ind, cnt, workspace, stream);
  if (rc) return rc;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(workspace, b, n, r3);
  dim3 grid(cdiv(r3, 256), (c + 7) / 8, b);
  hipLaunchKernelGGL(vox_reduce_s3_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, r3, features, bs_f, ld_f, cnt,
                     w.start, w.sorted, (unsigned short *)out_s3);
  return launch_status("vox_reduce_s3");
}
