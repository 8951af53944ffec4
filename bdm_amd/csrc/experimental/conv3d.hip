// experimental/conv3d.hip (built with `make EXPERIMENTAL=1` only: the fp32-MFMA convolution family, superseded by the fp16x3 /
// bf16x6 kernels of conv3d_h2.hip / conv3d_s3.hip; kept as the arithmetic yardstick of the accuracy tests) -- 3x3x3, stride 1, zero-padded convolution on the dense r^3 voxel grids of PVConv
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
#include "../../../include/bdm_hip.h"
#include "../common.h"

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

// Staging is software-pipelined through registers: the global loads of chunk c+1 are issued before
// the 27 x 4 MFMA steps of chunk c and written to LDS after them, so their latency hides under
// ~14-55k cycles of matrix work even at one wave per SIMD.  Grid resolution R and the tile shape are
// template constants so that every index split is a shift or a multiply-by-constant.
//
// SPARSE variant (first convolution of a PVConv: its input is the freshly voxelised point cloud, at most
// N of the r^3 cells are non-zero): `rowocc` (B, r*r) flags the (x, y) grid rows that hold at least one
// point.  A 32-wide column block shifted by a tap (dx, dy, .) reads only rows whose flag is clear -> its
// B operand is all zeros for every input channel -> the 3 x 4 x MI MFMAs of that (block, dx, dy) are
// skipped (wave-uniform branch).  Exact up to the order of additions of zeros, i.e. bit-identical.
template <int MI, int NI, int R, int TX, int TY, bool SPARSE>
__global__ __launch_bounds__(256) void conv3d_kernel(int Cin, int Cout, const float *__restrict__ x,
                                                     const float *__restrict__ wp, const float *__restrict__ bias,
                                                     const unsigned char *__restrict__ rowocc,
                                                     float *__restrict__ y) {
  extern __shared__ __align__(16) float smem[];
  constexpr int BM = 32 * MI;
  constexpr int RS = R + 8;                  // halo row stride (floats)
  constexpr int ROWS = (TX + 2) * (TY + 2);  // halo rows per channel
  constexpr int HALO = ROWS * RS;            // floats per input channel
  constexpr int R4 = R / 4, R2 = R * R, R3 = R2 * R;
  constexpr int XV = C3_BKC * ROWS * R4, XI = (XV + 255) / 256;      // float4 pieces of the input tile
  constexpr int WV = 27 * C3_BKC * (BM / 4), WI = (WV + 255) / 256;  // float4 pieces of the weight tile
  float *Xs = smem;                          // [C3_BKC][HALO]
  float *Ws = smem + C3_BKC * HALO;          // [27][C3_BKC][BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  constexpr int tiles_y = R / TY;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id.  With occupancy-dependent work (SPARSE) the busy
  // tiles of every shape would otherwise pile up on the same XCDs (same tile index modulo 8), so the tile index is
  // rotated by the group number: each XCD sees every tile residue.  Pure speed; any placement is correct.
  constexpr int NT = (R / TX) * tiles_y;
  int tile = blockIdx.x;
  if constexpr (SPARSE) {
    const int k = blockIdx.y + gridDim.y * blockIdx.z;
    if constexpr (NT % 8 == 0) tile = (tile & ~7) | ((tile + (tile >> 3) + k) & 7);
    else tile = (tile + k) % NT;
  }
  const int X0 = (tile / tiles_y) * TX, Y0 = (tile % tiles_y) * TY;
  const int m0 = blockIdx.y * BM, bi = blockIdx.z;
  const float *xb = x + (size_t)bi * Cin * R3;
  float *yb = y + (size_t)bi * Cout * R3;

  // per-lane voxel of each of this wave's NI column blocks
  constexpr int rpb = 32 / R;  // grid rows per 32-wide column block (1, 2 or 4)
  const int dyl = li / R, zl = li % R;
  constexpr int blocks_per_plane = TY / rpb;
  int lbase[NI], gvox[NI];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int nb = q * 4 + wave;  // interleaved: the 4 waves of a block get similar occupancy
    const int tx = nb / blocks_per_plane, ty = (nb % blocks_per_plane) * rpb + dyl;
    lbase[q] = ((tx + 1) * (TY + 2) + (ty + 1)) * RS + C3_ZOFF + zl;
    gvox[q] = ((X0 + tx) * R + (Y0 + ty)) * R + zl;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int q = 0; q < NI; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][q][i] = 0.f;

  // wave-uniform occupancy bits: bit t9 of occ[q] = "column block q shifted by (dx,dy) = t9 touches a non-empty row"
  unsigned occ[NI];
  if constexpr (SPARSE) {
    const unsigned char *ro = rowocc + (size_t)bi * R2;
    unsigned any = 0;
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int nb = q * 4 + wave;  // interleaved: the 4 waves of a block get similar occupancy
      const int tx = nb / blocks_per_plane, ty0 = (nb % blocks_per_plane) * rpb;
      unsigned m = 0;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int gx = X0 + tx + t9 / 3 - 1;
        unsigned hit = 0;
#pragma unroll
        for (int rr = 0; rr < rpb; ++rr) {
          const int gy = Y0 + ty0 + rr + t9 % 3 - 1;
          if (gx >= 0 && gx < R && gy >= 0 && gy < R) hit |= ro[gx * R + gy];
        }
        m |= (hit ? 1u : 0u) << t9;
      }
      occ[q] = __builtin_amdgcn_readfirstlane(m);
      any |= occ[q];
    }
    if (!__syncthreads_or(any != 0)) {  // nothing but zeros under this tile's halo: the output is the bias
#pragma unroll
      for (int p = 0; p < MI; ++p)
#pragma unroll
        for (int q = 0; q < NI; ++q)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int m = m0 + p * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (m < Cout) yb[(size_t)m * R3 + gvox[q]] = bias ? bias[m] : 0.f;
          }
      return;
    }
  }

  // zero the halo tile once: z pads and out-of-grid rows are never written again
  for (int e = tid; e < C3_BKC * HALO; e += 256) Xs[e] = 0.f;

  // ---- per-thread staging descriptors (chunk independent)
  int x_goff[XI], x_loff[XI];  // global offset within a chunk (-1: nothing to do), LDS offset
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + i * 256;
    const int z4 = (e % R4) * 4, row = (e / R4) % ROWS, ci = e / (R4 * ROWS);
    const int hx = row / (TY + 2), hy = row % (TY + 2);
    const int gx = X0 + hx - 1, gy = Y0 + hy - 1;
    const bool ok = e < XV && gx >= 0 && gx < R && gy >= 0 && gy < R;
    x_goff[i] = ok ? ci * R3 + (gx * R + gy) * R + z4 : -1;
    x_loff[i] = ci * HALO + row * RS + C3_ZOFF + z4;
  }
  float4 xr[XI], wr[WI];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x_goff[i] >= 0 && c0 + (x_goff[i] / R3) < Cin)
        xr[i] = *reinterpret_cast<const float4 *>(xb + (size_t)c0 * R3 + x_goff[i]);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * 256;
      const int m4 = (e % (BM / 4)) * 4, ci = (e / (BM / 4)) % C3_BKC, tap = e / ((BM / 4) * C3_BKC);
      wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < WV && c0 + ci < Cin && m0 + m4 < Cout)
        wr[i] = *reinterpret_cast<const float4 *>(wp + ((size_t)tap * Cin + c0 + ci) * Cout + m0 + m4);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (x_goff[i] >= 0) *reinterpret_cast<float4 *>(&Xs[x_loff[i]]) = xr[i];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * 256;
      if (e < WV) *reinterpret_cast<float4 *>(&Ws[e * 4]) = wr[i];
    }
  };

  load_chunk(0);
  for (int c0 = 0; c0 < Cin; c0 += C3_BKC) {
    __syncthreads();  // previous chunk fully consumed (first pass: the zero fill is complete)
    store_chunk();
    __syncthreads();
    if (c0 + C3_BKC < Cin) load_chunk(c0 + C3_BKC);  // in flight during the MFMA block below
    if constexpr (SPARSE) {
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int toff9 = ((t9 / 3 - 1) * (TY + 2) + (t9 % 3 - 1)) * RS;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
          if ((occ[q] >> t9) & 1u) {
            // 12 k-steps (3 dz x 4 channel pairs) in two batches: all LDS operands of a batch are requested
            // before its MFMAs so the read latency is paid once per batch, not once per MFMA
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              float bv[6], av[6][MI];
#pragma unroll
              for (int j = 0; j < 6; ++j) {
                const int st = h * 6 + j, dz = st / (C3_BKC / 2) - 1, kk = st % (C3_BKC / 2);
                const int tap = t9 * 3 + dz + 1;
                bv[j] = Xs[(2 * kk + lh) * HALO + toff9 + dz + lbase[q]];
#pragma unroll
                for (int p = 0; p < MI; ++p) av[j][p] = Ws[(tap * C3_BKC + 2 * kk + lh) * BM + li + p * 32];
              }
#pragma unroll
              for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int p = 0; p < MI; ++p)
                  acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][p], bv[j], acc[p][q], 0, 0, 0);
            }
          }
        }
      }
    } else {
      // ---- 27 taps x 4 k-steps, operands of step s+1 fetched from LDS ahead of the MFMAs of step s
      float a_cur[MI], b_cur[NI];
      {
        const int toff = (-(TY + 2) - 1) * RS - 1;
  #pragma unroll
        for (int p = 0; p < MI; ++p) a_cur[p] = Ws[lh * BM + li + p * 32];
  #pragma unroll
        for (int q = 0; q < NI; ++q) b_cur[q] = Xs[lh * HALO + toff + lbase[q]];
      }
  #pragma unroll
      for (int s = 0; s < 27 * (C3_BKC / 2); ++s) {
        float a_nxt[MI], b_nxt[NI];
        if (s + 1 < 27 * (C3_BKC / 2)) {
          const int tap = (s + 1) / (C3_BKC / 2), kk = (s + 1) % (C3_BKC / 2);
          const int dx = tap / 9 - 1, dy = (tap / 3) % 3 - 1, dz = tap % 3 - 1;
          const int toff = (dx * (TY + 2) + dy) * RS + dz;
  #pragma unroll
          for (int p = 0; p < MI; ++p) a_nxt[p] = Ws[(tap * C3_BKC + 2 * kk + lh) * BM + li + p * 32];
  #pragma unroll
          for (int q = 0; q < NI; ++q) b_nxt[q] = Xs[(2 * kk + lh) * HALO + toff + lbase[q]];
        }
  #pragma unroll
        for (int p = 0; p < MI; ++p)
  #pragma unroll
          for (int q = 0; q < NI; ++q)
            acc[p][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[p], b_cur[q], acc[p][q], 0, 0, 0);
        if (s + 1 < 27 * (C3_BKC / 2)) {
  #pragma unroll
          for (int p = 0; p < MI; ++p) a_cur[p] = a_nxt[p];
  #pragma unroll
          for (int q = 0; q < NI; ++q) b_cur[q] = b_nxt[q];
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
        if (m < Cout) yb[(size_t)m * R3 + gvox[q]] = acc[p][q][i] + (bias ? bias[m] : 0.f);
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


extern "C" int bdm_conv3d_3x3x3_sparse(int b, int cin, int cout, int r, const float *x, const float *packed_w,
                                       const float *bias, const unsigned char *rowocc, float *y, void *stream);

extern "C" int bdm_conv3d_3x3x3(int b, int cin, int cout, int r, const float *x, const float *packed_w,
                                const float *bias, float *y, void *stream) {
  return bdm_conv3d_3x3x3_sparse(b, cin, cout, r, x, packed_w, bias, nullptr, y, stream);
}

extern "C" int bdm_conv3d_3x3x3_sparse(int b, int cin, int cout, int r, const float *x, const float *packed_w,
                                       const float *bias, const unsigned char *rowocc, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1, "conv3d: bad sizes");
  BDM_REQUIRE(cout % 4 == 0, "conv3d: cout must be a multiple of 4 (GroupNorm(8) widths are)");
  if (r != 8 && r != 16 && r != 32) {
    set_error("conv3d: resolution %d unsupported (8, 16, 32 are the grids of the PVCNN denoisers)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const C3Cfg c = conv3d_cfg(cout, r);
  const size_t smem = conv3d_smem(c, r);
  dim3 grid((r / c.tx) * (r / c.ty), cdiv(cout, 32 * c.mi), b);
  hipStream_t s = (hipStream_t)stream;
#define C3_LAUNCH(MI, NI, R, TX, TY)                                                                              \
  do {                                                                                                            \
    if (rowocc) {                                                                                                 \
      BDM_ALLOW_LDS((conv3d_kernel<MI, NI, R, TX, TY, true>), smem);                                              \
      hipLaunchKernelGGL((conv3d_kernel<MI, NI, R, TX, TY, true>), grid, dim3(256), smem, s, cin, cout, x,        \
                         packed_w, bias, rowocc, y);                                                              \
    } else {                                                                                                      \
      BDM_ALLOW_LDS((conv3d_kernel<MI, NI, R, TX, TY, false>), smem);                                             \
      hipLaunchKernelGGL((conv3d_kernel<MI, NI, R, TX, TY, false>), grid, dim3(256), smem, s, cin, cout, x,       \
                         packed_w, bias, rowocc, y);                                                              \
    }                                                                                                             \
  } while (0)
  if (r == 32) { if (c.mi == 2) C3_LAUNCH(2, 4, 32, 2, 8); else C3_LAUNCH(1, 4, 32, 2, 8); }
  else if (r == 16) { if (c.mi == 2) C3_LAUNCH(2, 4, 16, 2, 16); else C3_LAUNCH(1, 4, 16, 2, 16); }
  else C3_LAUNCH(1, 2, 8, 4, 8);
#undef C3_LAUNCH
  return launch_status("conv3d");
}
