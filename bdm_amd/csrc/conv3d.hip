// conv3d.hip -- 3x3x3, stride 1, zero-padded convolution on the dense r^3 voxel grids of PVConv
// (reference: nn.Conv3d in experiments/model/pvcnn/modules/pvconv.py:75-85; ~95 % of the
// denoiser's FLOPs).
//
// Implicit GEMM on the f32-input matrix cores (v_mfma_f32_32x32x2_f32):
//   D[co][voxel] += W[tap][ci][co] * X[ci][voxel + tap]      K = 27 taps x Cin
// * the voxel index sits on the MFMA lane (channel-first grids: 32 consecutive z-cells -- or
//   2 x 16 / 4 x 8 cells of neighbouring rows for r = 16 / 8 -- form one 32-wide column block),
//   so the B operand is a conflict-free ds_read_b32 from a halo tile staged once per
//   8-channel chunk and reused by all 27 taps;
// * weights are pre-packed to [tap][ci][co] (bdm_conv3d_pack_weights) so the A operand is a
//   conflict-free read of 32 consecutive output channels;
// * one 256-thread workgroup = 4 waves side by side along the voxel axis, each holding
//   MI x NI accumulator tiles of 32 x 32.
#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define C3_BKC 8   // input channels per staged chunk
#define C3_ZOFF 4  // interior of a halo row starts 16-B aligned; z = -1 sits at ZOFF-1

__global__ void conv3d_pack_kernel(int cout, int cin, const float *__restrict__ w, float *__restrict__ wp) {
  const long long total = (long long)cout * cin * 27;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(e % cout);
    const int ci = (int)((e / cout) % cin);
    const int tap = (int)(e / ((long long)cout * cin));
    wp[e] = w[((size_t)co * cin + ci) * 27 + tap];
  }
}

extern "C" int bdm_conv3d_pack_weights(int cout, int cin, const float *w, float *packed, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1, "conv3d_pack_weights: bad sizes");
  hipLaunchKernelGGL(conv3d_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, cout, cin, w, packed);
  return launch_status("conv3d_pack_weights");
}

template <int MI, int NI>
__global__ __launch_bounds__(256) void conv3d_kernel(int Cin, int Cout, int r, int TX, int TY,
                                                     const float *__restrict__ x, const float *__restrict__ wp,
                                                     const float *__restrict__ bias, float *__restrict__ y) {
  extern __shared__ __align__(16) float smem[];
  constexpr int BM = 32 * MI;
  const int RS = r + 8;                       // halo row stride (floats)
  const int HALO = (TX + 2) * (TY + 2) * RS;  // floats per input channel
  float *Xs = smem;                           // [C3_BKC][HALO]
  float *Ws = smem + C3_BKC * HALO;           // [27][C3_BKC][BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int r2 = r * r, r3 = r2 * r;
  const int tiles_y = r / TY;
  const int X0 = (blockIdx.x / tiles_y) * TX, Y0 = (blockIdx.x % tiles_y) * TY;
  const int m0 = blockIdx.y * BM, bi = blockIdx.z;
  const float *xb = x + (size_t)bi * Cin * r3;
  float *yb = y + (size_t)bi * Cout * r3;

  // per-lane voxel of each of this wave's NI column blocks
  const int rpb = 32 / r;  // grid rows per 32-wide column block (1, 2 or 4)
  const int dyl = li / r, zl = li % r;
  const int blocks_per_plane = TY / rpb;
  int lbase[NI], gvox[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int nb = wave * NI + q;
    const int tx = nb / blocks_per_plane, ty = (nb % blocks_per_plane) * rpb + dyl;
    lbase[q] = ((tx + 1) * (TY + 2) + (ty + 1)) * RS + C3_ZOFF + zl;
    gvox[q] = ((X0 + tx) * r + (Y0 + ty)) * r + zl;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int q = 0; q < NI; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][q][i] = 0.f;

  // zero the halo tile once: z pads and out-of-grid rows are never written again
  for (int e = tid; e < C3_BKC * HALO; e += 256) Xs[e] = 0.f;

  const int rows_per_ch = (TX + 2) * (TY + 2);
  const int r4 = r / 4;  // float4 per row
  for (int c0 = 0; c0 < Cin; c0 += C3_BKC) {
    __syncthreads();  // previous chunk fully consumed (and the zero fill above is complete)
    // ---- stage the input halo tile: interior cells of in-grid rows
    for (int e = tid; e < C3_BKC * rows_per_ch * r4; e += 256) {
      const int z4 = (e % r4) * 4;
      const int row = (e / r4) % rows_per_ch;
      const int ci = e / (r4 * rows_per_ch);
      const int hx = row / (TY + 2), hy = row % (TY + 2);
      const int gx = X0 + hx - 1, gy = Y0 + hy - 1;
      if (gx >= 0 && gx < r && gy >= 0 && gy < r) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c0 + ci < Cin)
          v = *reinterpret_cast<const float4 *>(xb + (size_t)(c0 + ci) * r3 + (gx * r + gy) * r + z4);
        *reinterpret_cast<float4 *>(&Xs[ci * HALO + row * RS + C3_ZOFF + z4]) = v;
      }
    }
    // ---- stage the weight tile [27][8][BM]
    for (int e = tid; e < 27 * C3_BKC * BM; e += 256) {
      const int m = e % BM;
      const int ci = (e / BM) % C3_BKC;
      const int tap = e / (BM * C3_BKC);
      float v = 0.f;
      if (c0 + ci < Cin && m0 + m < Cout) v = wp[((size_t)tap * Cin + c0 + ci) * Cout + m0 + m];
      Ws[e] = v;
    }
    __syncthreads();
    // ---- 27 taps x 4 k-steps
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz) {
          const int tap = (dx + 1) * 9 + (dy + 1) * 3 + (dz + 1);
          const int toff = (dx * (TY + 2) + dy) * RS + dz;
#pragma unroll
          for (int kk = 0; kk < C3_BKC / 2; ++kk) {
            float a[MI], b[NI];
            const float *wrow = Ws + (tap * C3_BKC + 2 * kk + lh) * BM + li;
            const float *xrow = Xs + (2 * kk + lh) * HALO + toff;
#pragma unroll
            for (int p = 0; p < MI; ++p) a[p] = wrow[p * 32];
#pragma unroll
            for (int q = 0; q < NI; ++q) b[q] = xrow[lbase[q]];
#pragma unroll
            for (int p = 0; p < MI; ++p)
#pragma unroll
              for (int q = 0; q < NI; ++q)
                acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], b[q], acc[p][q], 0, 0, 0);
          }
        }
      }
  }
  // ---- epilogue
#pragma unroll
  for (int p = 0; p < MI; ++p)
#pragma unroll
    for (int q = 0; q < NI; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + p * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
        if (m < Cout) yb[(size_t)m * r3 + gvox[q]] = acc[p][q][i] + (bias ? bias[m] : 0.f);
      }
}

struct C3Cfg { int mi, ni, tx, ty; };
static C3Cfg conv3d_cfg(int cout, int r) {
  C3Cfg c;
  if (r == 32) { c.tx = 2; c.ty = 8; c.ni = 4; }
  else if (r == 16) { c.tx = 2; c.ty = 16; c.ni = 4; }
  else { c.tx = 4; c.ty = 8; c.ni = 2; }  // r == 8
  c.mi = (cout > 32 && c.ni == 4) ? 2 : 1;
  return c;
}
static size_t conv3d_smem(const C3Cfg &c, int r) {
  return sizeof(float) * ((size_t)C3_BKC * (c.tx + 2) * (c.ty + 2) * (r + 8) + (size_t)27 * C3_BKC * 32 * c.mi);
}

extern "C" int bdm_conv3d_3x3x3(int b, int cin, int cout, int r, const float *x, const float *packed_w,
                                const float *bias, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1, "conv3d: bad sizes");
  if (r != 8 && r != 16 && r != 32) {
    set_error("conv3d: resolution %d unsupported (8, 16, 32 are the grids of the PVCNN denoisers)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const C3Cfg c = conv3d_cfg(cout, r);
  const size_t smem = conv3d_smem(c, r);
  dim3 grid((r / c.tx) * (r / c.ty), cdiv(cout, 32 * c.mi), b);
  hipStream_t s = (hipStream_t)stream;
#define C3_LAUNCH(MI, NI)                                                                                       \
  do {                                                                                                          \
    BDM_ALLOW_LDS((conv3d_kernel<MI, NI>), smem);                                                               \
    hipLaunchKernelGGL((conv3d_kernel<MI, NI>), grid, dim3(256), smem, s, cin, cout, r, c.tx, c.ty, x, packed_w, \
                       bias, y);                                                                                \
  } while (0)
  if (c.mi == 2 && c.ni == 4) C3_LAUNCH(2, 4);
  else if (c.mi == 1 && c.ni == 4) C3_LAUNCH(1, 4);
  else C3_LAUNCH(1, 2);
#undef C3_LAUNCH
  return launch_status("conv3d");
}
