// conv3d_h2.hip -- the 3x3x3 voxel convolution at fp32 accuracy on the FP16 matrix cores ("fp16x3").
//
// Half the matrix work of the bf16x6 form (conv3d_s3.hip) at the same accuracy class.  An fp32 operand is stored as
// TWO fp16 terms  x * 2^s = hi + lo  (hi = RN_11(x 2^s), lo = RN_11(x 2^s - hi); 11 + 11 signed bits capture x to
// <= 2^-24 |x|), and a product is the three partial products  lo.hi + hi.lo + hi.hi , each exact in fp32 (11 x 11
// bits) and accumulated in fp32 by v_mfma_f32_16x16x32_f16; the dropped lo.lo term is <= 2^-24 |a b|.
// fp16 has a 5-bit exponent, so both operands are pre-scaled by exact powers of two that keep hi AND lo in the normal
// range, and the scales are divided out of the fp32 accumulator in the epilogue:
//   weights      per output channel: max |w[co]| -> [2^9, 2^10)  (pack time; inv_scale[co] for the epilogue)
//   activations  one power of two per call, chosen by the caller from what it knows about the tensor: on this path
//                they are GroupNorm + Swish outputs, |y| <= |gamma| |z| + |beta|, and the host picks the scale from
//                the layer's gamma / beta so that a 64-sigma value still fits (larger ones saturate at 65504
//                instead of overflowing to inf)
// Below the normal range an operand keeps an ABSOLUTE error <= 2^-25 in scaled units: 2^-25 / act_scale for
// activations, 2^-35 max|w[co]| for weights -- under the fp32 resolution of any sum those operands take part in.
// Measured <= 3e-7 relative L2 vs fp64 (tests/test_hip_dense.py), the same as the fp32-input MFMA kernel.
//
// Layouts ("H2"): activations (B, ceil(C/8), 2, r^3, 8) fp16; weights [ceil(Cin/8)][14 tap pairs][2][2][Cout][8] fp16.
// Geometry, staging and tap addressing are those of conv3d_s3.hip.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define H2_PAIRS 14

__device__ __forceinline__ void split2(float v, unsigned short &h, unsigned short &l) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);  // saturate instead of overflowing to inf
  const _Float16 hi = (_Float16)v;          // round to nearest even
  const _Float16 lo = (_Float16)(v - (float)hi);  // the remainder is exact in fp32
  h = __builtin_bit_cast(unsigned short, hi);
  l = __builtin_bit_cast(unsigned short, lo);
}

__device__ __forceinline__ void store_h2(unsigned short *base, size_t rec, size_t split_stride, const float v[8]) {
  unsigned short h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split2(v[j], h[j], l[j]);
  uint4 ph, pl;
  ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
  pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
  uint4 *o = reinterpret_cast<uint4 *>(base);
  o[rec] = ph;
  o[split_stride + rec] = pl;
}

// ---------------------------------------------------------------------------------------------------
// weight packing: per-output-channel power-of-two scale, then (Cout, Cin, 3,3,3) fp32 -> [C8][14][2][2][Cout][8] fp16
// ---------------------------------------------------------------------------------------------------
__global__ void h2_weight_scale_kernel(int cout, int cin, const float *__restrict__ w, float *__restrict__ scale,
                                       float *__restrict__ inv_scale) {
  __shared__ float sh[256];
  const int co = blockIdx.x;
  float m = 0.f;
  for (int e = threadIdx.x; e < cin * 27; e += blockDim.x) m = fmaxf(m, fabsf(w[(size_t)co * cin * 27 + e]));
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int ex = 0;
    const float mx = sh[0];
    if (mx > 0.f && mx < INFINITY) (void)frexpf(mx, &ex);  // mx = f * 2^ex, f in [0.5, 1)
    const int e = (mx > 0.f && mx < INFINITY) ? 10 - ex : 0;  // mx * 2^e in [2^9, 2^10)
    scale[co] = ldexpf(1.0f, e);
    inv_scale[co] = ldexpf(1.0f, -e);
  }
}
__global__ void pack_h2_kernel(int cout, int cin, const float *__restrict__ w, const float *__restrict__ scale,
                               unsigned short *__restrict__ wq) {
  const int c8n = (cin + 7) / 8;
  const long long total = (long long)c8n * H2_PAIRS * 2 * cout * 8;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e % 8);
    const int co = (int)((e / 8) % cout);
    const int h = (int)((e / (8ll * cout)) % 2);
    const int p = (int)((e / (16ll * cout)) % H2_PAIRS);
    const int c8 = (int)(e / (16ll * cout * H2_PAIRS));
    const int ci = c8 * 8 + j, tap = 2 * p + h;
    const float v = (ci < cin && tap < 27) ? w[((size_t)co * cin + ci) * 27 + tap] * scale[co] : 0.f;  // exact scaling
    unsigned short s[2];
    split2(v, s[0], s[1]);
#pragma unroll
    for (int q = 0; q < 2; ++q) wq[((((size_t)(c8 * H2_PAIRS + p) * 2 + q) * 2 + h) * cout + co) * 8 + j] = s[q];
  }
}
extern "C" size_t bdm_conv3d_h2_weight_elems(int cout, int cin) {
  return (size_t)((cin + 7) / 8) * H2_PAIRS * 2 * 2 * cout * 8;
}
extern "C" int bdm_conv3d_h2_pack_weights(int cout, int cin, const float *w, void *packed, float *scale_ws,
                                          float *inv_scale, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1 && scale_ws != nullptr && inv_scale != nullptr, "conv3d_h2_pack_weights: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(h2_weight_scale_kernel, dim3(cout), dim3(256), 0, s, cout, cin, w, scale_ws, inv_scale);
  hipLaunchKernelGGL(pack_h2_kernel, dim3(512), dim3(256), 0, s, cout, cin, w, scale_ws, (unsigned short *)packed);
  return launch_status("conv3d_h2_pack_weights");
}

// ---------------------------------------------------------------------------------------------------
// the convolution, on v_mfma_f32_16x16x32_f16 (K = 8 channels x FOUR taps per instruction)
// ---------------------------------------------------------------------------------------------------
// Implicit GEMM: a workgroup owns BM output channels x a TX x TY x R brick of voxels; per 8-channel chunk it stages the brick's
// halo (both fp16 planes) and the chunk's 27 x BM weights in LDS, then walks the 7 tap quads.  Round 3 moved it from
// v_mfma_f32_32x32x16_f16 (tap PAIRS) to the 16x16x32 shape: the chip holds a higher clock on it at the same flops
// (MI355X_MICROARCH.md, DVFS item 7), measured on random data at B = 16: 256->256 at 8^3 94.0 -> 85.0 us, 512->256 190.5 -> 172.6,
// 32->32 at 32^3 124.1 -> 118.4, 128->128 at 16^3 166.0 -> 158.4, 64->64 at 32^3 389.7 -> 380.6 (tools/conv_h2_check.py); and a
// 16 x 16 output tile lets a tiny batch spread a layer over twice the workgroups with half the dependent chain per wave.
//   wave tile   MT x NT tiles of 16 output channels x 16 voxels; a workgroup = NW waves on the same 16 MT channels
//   K step      tap quad Q (7 of them; tap 27 is a zero-weight pad): lane group kg = lane / 16 holds tap 4Q + kg, 8 channels
//   A operand   weights  Ws[pair 2Q + kg/2][split][half kg&1][channel]   (the pair-major image of the packer, read as quads)
//   B operand   voxels   Xs[split][halo record of the lane's voxel + offset of tap 4Q + kg]
// GroupNorm slice partials on a canonical decomposition (independent of the tile): a unit = 16^3: one x-plane, 8^3: one x-plane,
// 32^3: the 2 x 8 tile; summed as: lane (4 rows of ONE 16-voxel block), butterfly over the 16 lanes of the block, the unit's
// blocks in ascending order (fp32), then the 4-row blocks of a group in ascending row order (fp64).
typedef __attribute__((ext_vector_type(4))) float f32x4a;
template <int MT, int NT, int R, int TX, int TY, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void conv3d_h2q_kernel(
    int C8, int Cout, const float4 *__restrict__ x, const float4 *__restrict__ wq, const float *__restrict__ inv_scale, float x_inv_scale,
    const float *__restrict__ bias, float *__restrict__ y, int gn_cg, double *__restrict__ gn_partial) {
  extern __shared__ __align__(16) float4 smem4[];
  constexpr int BM = 16 * MT;
  constexpr int RSV = R + 2;                 // voxel records per halo row (one zero pad at each end)
  constexpr int ROWS = (TX + 2) * (TY + 2);
  constexpr int HALO = ROWS * RSV;           // records per split
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int NT_ = NW * 64;
  constexpr int XV = 2 * ROWS * R, XI = (XV + NT_ - 1) / NT_;          // 16-byte pieces of the input tile
  constexpr int WV = H2_PAIRS * 2 * 2 * BM, WI = (WV + NT_ - 1) / NT_;  // 16-byte pieces of the weight tile
  constexpr int NQ = H2_PAIRS / 2;                                      // tap quads
  static_assert(TX * TY * R == NT * NW * 16, "tile = NT * NW blocks of 16 voxels");
  float4 *Xs = smem4;              // [2][HALO]
  float4 *Ws = smem4 + 2 * HALO;   // [14][2][2][BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  constexpr int tiles_y = R / TY;
  const int X0 = (blockIdx.x / tiles_y) * TX, Y0 = (blockIdx.x % tiles_y) * TY;
  const int m0 = blockIdx.y * BM, bi = blockIdx.z;
  const float4 *xb = x + (size_t)bi * C8 * 2 * R3;
  float *yb = y + (size_t)bi * Cout * R3;

  // voxel of this lane in block q * NW + wave of the tile, (x-plane, y, z) order: the NW blocks of one q cover RQ whole rows, so
  // the halo record of block q is the record of block 0 plus a COMPILE-TIME offset (one register instead of NT)
  constexpr int RQ = NW * 16 / R;
  static_assert(NW * 16 % R == 0 && (RQ % TY == 0 || TY % RQ == 0), "the blocks of one q are whole rows that tile the brick");
  const int v0 = wave * 16 + l16, r0 = v0 / R, tz = v0 % R;
  const int lbase0 = ((r0 / TY + 1) * (TY + 2) + (r0 % TY + 1)) * RSV + 1 + tz;
  auto row_of = [&](int q, int &tx, int &ty) {  // brick row (tx, ty) of block q, relative to the lane's row of block 0
    if (RQ % TY == 0) { tx = q * (RQ / TY); ty = 0; }
    else { tx = (q * RQ) / TY; ty = (q * RQ) % TY; }
  };
  auto lbase = [&](int q) { int tx, ty; row_of(q, tx, ty); return lbase0 + (tx * (TY + 2) + ty) * RSV; };
  // record offset of this lane group's tap in every quad (pad tap 27 reuses tap 26's address: its weights are zero)
  int toff[NQ];
#pragma unroll
  for (int Q = 0; Q < NQ; ++Q) {
    const int t = min(4 * Q + kg, 26);
    toff[Q] = ((t / 9 - 1) * (TY + 2) + ((t / 3) % 3 - 1)) * RSV + (t % 3 - 1);
  }
  // weight record of (quad Q, split s, channel tile mt): wbase + Q * 8 BM + s * 2 BM + mt * 16   (pair 2Q + kg/2, half kg&1)
  const float4 *wbase = Ws + ((kg >> 1) * 4 + (kg & 1)) * BM + l16;
  f32x4a acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[a][q] = f32x4a{0.f, 0.f, 0.f, 0.f};

  // the tile's output scales and biases go to LDS now (visible after the K loop's first barrier): read from global memory in the epilogue they
  // were one more dependent round trip between the last MFMA and the stores (as in sconv_dil_kernel)
  __shared__ float s_osc[BM], s_obi[BM];
  if (tid < BM) {
    const int m = min(m0 + tid, Cout - 1);
    s_osc[tid] = inv_scale[m] * x_inv_scale;
    s_obi[tid] = bias ? bias[m] : 0.f;
  }
  for (int e = tid; e < 2 * HALO; e += NT_) Xs[e] = make_float4(0.f, 0.f, 0.f, 0.f);

  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v xr[XI], wr[WI];
  auto load_chunk = [&](int c8) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + i * NT_;
      const int z = e % R, row = (e / R) % ROWS, s = e / (R * ROWS);
      const int gx = X0 + row / (TY + 2) - 1, gy = Y0 + row % (TY + 2) - 1;
      const bool ok = e < XV && gx >= 0 && gx < R && gy >= 0 && gy < R;
      xr[i] = *reinterpret_cast<const f32x4v *>(&xb[ok ? (unsigned)((c8 * 2 + s) * R3 + (gx * R + gy) * R + z) : 0u]);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * NT_;
      const int m = e % BM, psh = e / BM;  // psh = (p*2 + s)*2 + h
      const bool ok = e < WV && m0 + m < Cout;
      wr[i] = *reinterpret_cast<const f32x4v *>(&wq[ok ? (unsigned)((c8 * (H2_PAIRS * 4) + psh) * Cout + m0 + m) : 0u]);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int e = tid + i * NT_;
      const int z = e % R, row = (e / R) % ROWS, s = e / (R * ROWS);
      const int gx = X0 + row / (TY + 2) - 1, gy = Y0 + row % (TY + 2) - 1;
      if (e < XV && gx >= 0 && gx < R && gy >= 0 && gy < R) *reinterpret_cast<f32x4v *>(&Xs[s * HALO + row * RSV + 1 + z]) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * NT_;
      if (e < WV) *reinterpret_cast<f32x4v *>(&Ws[e]) = wr[i];
    }
  };

  load_chunk(0);
  for (int c8 = 0; c8 < C8; ++c8) {
    __syncthreads();
    store_chunk();
    __syncthreads();
    if (c8 + 1 < C8) load_chunk(c8 + 1);
    // Fragments of quad Q + 1 are read from LDS before the matrix work of quad Q is issued (two register buffers).  The 64 x 64 wave
    // tile (MT = NT = 4) has no registers for two buffers at two waves per SIMD (64 accumulator + 128 fragment + 48 staging
    // registers): its fragments are single-buffered and re-read right after the quad's 48 MFMAs are issued -- the other wave of
    // the SIMD covers that latency.
    constexpr bool DB = MT * NT < 16;
    f16x8 fa[DB ? 2 : 1][MT][2], fb[DB ? 2 : 1][NT][2];
    auto read_quad = [&](int Q, int buf) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4 t = wbase[Q * 8 * BM + s * 2 * BM + mt * 16];
          fa[buf][mt][s] = *reinterpret_cast<const f16x8 *>(&t);
        }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          const float4 t = Xs[s * HALO + lbase(q) + toff[Q]];
          fb[buf][q][s] = *reinterpret_cast<const f16x8 *>(&t);
        }
      }
    };
    read_quad(0, 0);
#pragma unroll
    for (int Q = 0; Q < NQ; ++Q) {
      if (DB && Q + 1 < NQ) read_quad(Q + 1, (Q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      // smallest terms first: lo.hi, hi.lo, hi.hi; term-major: the MT * NT accumulators are independent
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < NT; ++q)
            acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[DB ? (Q & 1) : 0][mt][term == 0 ? 1 : 0], fb[DB ? (Q & 1) : 0][q][term == 1 ? 1 : 0],
                                                               acc[mt][q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (!DB && Q + 1 < NQ) read_quad(Q + 1, 0);
    }
  }
  // Epilogue: scale + bias + store (+ GroupNorm slice partials, see the head comment).  A lane holds rows 4 kg .. 4 kg + 3 of
  // channel tile mt for ONE voxel: a 4-row block never straddles a group (gn_cg >= 4).
  constexpr int UB = (R == 32 ? TX * TY * R : R * R) / 16;   // 16-voxel blocks per canonical unit
  constexpr int NBLK = NT * NW, UN = NBLK / UB;             // blocks of this tile, units of this tile
  constexpr int NB = MT * 4 * 2;                            // [mt][kg][stat]
  static_assert(NBLK % UB == 0 && UN >= 1, "a tile holds whole canonical units");
  float *red = reinterpret_cast<float *>(smem4);            // [NBLK][NB], then [UN][NB]
  if (gn_partial != nullptr) __syncthreads();  // the operand tiles are dead (slower waves may still be reading fragments)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      int qx, qy;
      row_of(q, qx, qy);
      const int gvox = ((X0 + r0 / TY + qx) * R + (Y0 + r0 % TY + qy)) * R + tz;
      float bs = 0.f, bq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + mt * 16 + 4 * kg + i;
        if (m < Cout) {
          const float v = acc[mt][q][i] * s_osc[mt * 16 + 4 * kg + i] + s_obi[mt * 16 + 4 * kg + i];
          yb[(size_t)m * R3 + gvox] = v;
          bs += v;
          bq = __builtin_fmaf(v, v, bq);  // explicitly fused: the same rounding in every tile variant
        }
      }
      if (gn_partial != nullptr) {
        bs = row16_sum(bs);   // (DPP: bit-identical to the xor butterfly over the 16 voxels of the block)
        bq = row16_sum(bq);
        if (l16 == 0) {
          const int nb = q * NW + wave;
          red[nb * NB + (mt * 4 + kg) * 2 + 0] = bs;
          red[nb * NB + (mt * 4 + kg) * 2 + 1] = bq;
        }
      }
    }
  if (gn_partial != nullptr) {
    __syncthreads();
    float *red2 = red + NBLK * NB;
    for (int e = tid; e < UN * NB; e += NT_) {
      const int un = e / NB, k = e % NB;
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < UB; ++j) a += red[(un * UB + j) * NB + k];  // the unit's blocks in ascending order
      red2[e] = a;
    }
    __syncthreads();
    const int ngt = BM / gn_cg;  // groups inside this channel tile
    if (tid < UN * ngt) {
      const int un = tid / ngt, gi = tid % ngt;
      if (m0 + gi * gn_cg < Cout) {
        double a = 0.0, qq = 0.0;
        const int nb4 = gn_cg / 4;  // 4-row blocks per group, ascending rows
        for (int j = 0; j < nb4; ++j) {
          a += (double)red2[un * NB + (gi * nb4 + j) * 2 + 0];
          qq += (double)red2[un * NB + (gi * nb4 + j) * 2 + 1];
        }
        const int G = Cout / gn_cg, g = m0 / gn_cg + gi, S = gridDim.x * UN;
        double *dst = gn_partial + (((size_t)bi * G + g) * S + blockIdx.x * UN + un) * 2;
        dst[0] = a;
        dst[1] = qq;
      }
    }
  }
}

static int conv3d_h2_launch(int b, int cin, int cout, int r, const void *x_h2, float x_inv_scale, const void *packed_w,
                            const float *inv_scale, const float *bias, float *y, int gn_cg, double *gn_partial,
                            int *slices_out, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && inv_scale != nullptr, "conv3d_h2: bad arguments");
  if (r != 8 && r != 16 && r != 32) {
    set_error("conv3d_h2: resolution %d unsupported (8, 16, 32)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const int c8 = (cin + 7) / 8;
  // tile = TX x TY grid rows x R cells = NT * NW blocks of 16 voxels x 32 * mi output channels; 8 waves (two per SIMD) keep the
  // matrix pipe fed while the other wave of the SIMD waits on LDS (round 2 measured 4 waves with twice the tile per wave: 2-15 % slower)
  int tx, ty, mi;
  // 8^3 grids: 128-voxel tiles with 4 waves double the workgroup count; they win when the 256-voxel tiling cannot fill the chip
  // (128 -> 128 at B = 16: 40.6 -> 30.4 us) and lose when it can (256 -> 256: 96 -> 114 us).
  const bool r8_small = r == 8 && (long long)b * 2 * cdiv(cout, 32) < 256;
  // ... and 64-voxel tiles (one 16 x 16 block per wave: twice the workgroups again, half the chain per wave) when the 128-voxel tiling
  // cannot either: the LARGEST tile that still gives a workgroup per CU wins at every (batch, width) measured (round 5, us per launch,
  // 64 / 128 / 256-voxel tiles: B = 8, 128 channels 20.1 / 27.9 / 37.5; B = 4, 256 channels 36.4 / 50.3 / 69.1; B = 8, 256 channels
  // 71.2 / 54.8 / 73.3; B = 16, 128 channels 38.0 / 30.6 / 39.7; B = 16, 256 channels 144.7 / 109.6 / 83.4).  Same K order, same bits.
  const bool r8_tiny = r == 8 && (long long)b * 4 * cdiv(cout, 32) < 256;
  // 16^3 / 32^3 grids of a few shapes (config C1 is ONE shape): 64-row x 512-voxel tiles give 16 / 64 workgroups, each walking
  // the whole K loop with 48 MFMAs per step; 32-row (and at 16^3 256-voxel) tiles put 4x / 2x as many CUs on the problem with a
  // 4x / 2x shorter chain per wave (B = 1: 114 -> 4x us at 16^3).  Same K order, same bits.
  const long long big_wgs = (long long)b * (r == 32 ? 64 : 8) * cdiv(cout, 64);
  // Threshold = a workgroup per CU (round 5, us per launch, big / small tiling: 16^3 128 channels B = 8 126.6 / 89.0, B = 16 149.1 / 190.5;
  // 16^3 64 channels B = 8 66.7 / 26.6, B = 16 71.7 / 50.2; 32^3 64 channels B = 2 75.4 / 46.2, B = 4 84.2 / 89.2).  It was 128: the
  // half-filled chip of 128 big workgroups lost 30 - 60 % (C4's 16^3 levels, C2's 64-channel 16^3 layer).
  const bool small = r != 8 && cout > 32 && big_wgs < 256;
  if (r == 32) { tx = 2; ty = 8; }
  else if (r == 16) { tx = small ? 1 : 2; ty = 16; }
  else { tx = r8_tiny ? 1 : (r8_small ? 2 : 4); ty = 8; }
  mi = (cout > 32 && r != 8 && !small) ? 2 : 1;
  const size_t smem = 16 * ((size_t)2 * (tx + 2) * (ty + 2) * (r + 2) + (size_t)H2_PAIRS * 4 * 32 * mi);
  dim3 grid((r / tx) * (r / ty), cdiv(cout, 32 * mi), b);
  hipStream_t s = (hipStream_t)stream;
  if (gn_partial != nullptr) {
    BDM_REQUIRE(gn_cg >= 4 && (gn_cg & (gn_cg - 1)) == 0 && (32 * mi) % gn_cg == 0 && cout % gn_cg == 0 && (int)grid.x <= 64,
                "conv3d_h2: GroupNorm statistics need a power-of-two channels-per-group dividing %d and <= 64 spatial tiles (got cg=%d)", 32 * mi, gn_cg);
    // canonical slices (independent of the tile): 16^3 and 8^3 one per x-plane, 32^3 one per 2 x 8 tile
    if (slices_out) *slices_out = r == 16 ? 16 : (r == 8 ? 8 : (int)grid.x);
  }
#define H2Q_LAUNCH(MT, NT, R, TX, TY, NW)                                                                       \
  do {                                                                                                          \
    BDM_ALLOW_LDS((conv3d_h2q_kernel<MT, NT, R, TX, TY, NW>), smem);                                            \
    hipLaunchKernelGGL((conv3d_h2q_kernel<MT, NT, R, TX, TY, NW>), grid, dim3(NW * 64), smem, s, c8, cout,      \
                       (const float4 *)x_h2, (const float4 *)packed_w, inv_scale, x_inv_scale, bias, y, gn_cg,  \
                       gn_partial);                                                                             \
  } while (0)
  if (r == 32) { if (mi == 2) H2Q_LAUNCH(4, 4, 32, 2, 8, 8); else H2Q_LAUNCH(2, 4, 32, 2, 8, 8); }
  else if (r == 16) {
    if (small) H2Q_LAUNCH(2, 2, 16, 1, 16, 8);
    else if (mi == 2) H2Q_LAUNCH(4, 4, 16, 2, 16, 8);
    else H2Q_LAUNCH(2, 4, 16, 2, 16, 8);
  } else {
    if (r8_tiny) H2Q_LAUNCH(2, 1, 8, 1, 8, 4);
    else if (r8_small) H2Q_LAUNCH(2, 2, 8, 2, 8, 4);
    else H2Q_LAUNCH(2, 2, 8, 4, 8, 8);
  }
#undef H2Q_LAUNCH
  return launch_status("conv3d_h2");
}

extern "C" int bdm_conv3d_3x3x3_h2(int b, int cin, int cout, int r, const void *x_h2, float x_inv_scale,
                                   const void *packed_w, const float *inv_scale, const float *bias, float *y,
                                   void *stream) {
  return conv3d_h2_launch(b, cin, cout, r, x_h2, x_inv_scale, packed_w, inv_scale, bias, y, 0, nullptr, nullptr, stream);
}

// The same convolution, also leaving the GroupNorm(groups) statistics of its OUTPUT as slice partials in `gn_workspace`
// (layout and size of bdm_group_norm_workspace_bytes(b, groups); *slices_out = slices per (shape, group)).
extern "C" int bdm_conv3d_3x3x3_h2_gn(int b, int cin, int cout, int r, const void *x_h2, float x_inv_scale,
                                      const void *packed_w, const float *inv_scale, const float *bias, float *y, int groups,
                                      void *gn_workspace, int *slices_out, void *stream) {
  BDM_REQUIRE(groups >= 1 && cout % groups == 0 && gn_workspace != nullptr && slices_out != nullptr, "conv3d_h2_gn: bad arguments");
  return conv3d_h2_launch(b, cin, cout, r, x_h2, x_inv_scale, packed_w, inv_scale, bias, y, cout / groups,
                          (double *)gn_workspace, slices_out, stream);
}

// ---------------------------------------------------------------------------------------------------
// producer of the H2 layout: fp32 channel-first (B, C, V) through GroupNorm (+ Swish), times 16, split in two.
// The fused normalise step between the two convolutions of a PVConv (pvconv.py:78-82).
// ---------------------------------------------------------------------------------------------------
__global__ void to_h2_kernel(int C, int V, int G, int S, const float *__restrict__ x, const double *__restrict__ partial,
                             const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int act,
                             float act_scale, unsigned short *__restrict__ out, unsigned *__restrict__ saturated) {
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
  bool sat = false;
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
    val[j] = t * act_scale;  // a power of two: exact
    sat |= !(fabsf(val[j]) <= 65504.f);  // beyond fp16's range (or NaN): split2 clamps -- the caller must be told
  }
  store_h2(out + ((size_t)bi * C8 + c8) * 2 * (size_t)V * 8, (size_t)v, (size_t)V, val);
  // one sticky word per layer; an OR is order-independent, and the word is only ever written when something saturated
  if (saturated != nullptr && __ballot(sat) != 0ull && (threadIdx.x & 63) == __ffsll((long long)__ballot(sat)) - 1)
    atomicOr(saturated, 1u);
}

// The same split with the GroupNorm statistics taken from MANY slice partials (the sparse gather leaves r*r per (shape, group)):
// fat workgroups (8 channels x VB voxels) reduce their group's partials once, in parallel and in a fixed order (thread t adds
// slices t, t + 256, ...; half-wave... whole-wave butterfly; waves added in order), then stream VB voxels.
template <int VB>
__global__ __launch_bounds__(256) void to_h2_stats_kernel(int C, int V, int G, int S, const float *__restrict__ x,
                                                          const double *__restrict__ partial, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps, int act, float act_scale,
                                                          unsigned short *__restrict__ out, unsigned *__restrict__ saturated) {
  __shared__ float s_mean[2], s_rstd[2];
  __shared__ double s_red[2][4][2];
  const int bi = blockIdx.z, c8 = blockIdx.y, C8 = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = C / G;
  const int g_lo = (c8 * 8) / cg, g_hi = min((c8 * 8 + 7) / cg, G - 1);  // the block's 8 channels touch <= 2 groups (cg >= 4)
  for (int gi = g_lo; gi <= g_hi; ++gi) {
    const double *pp = partial + ((size_t)bi * G + gi) * S * 2;
    double a = 0.0, q = 0.0;
    for (int sl = tid; sl < S; sl += 256) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    a = wave_sum_bfly(a); q = wave_sum_bfly(q);   // (DPP + readlanes, bit-identical to the 64-lane xor butterfly)
    if (lane == 0) { s_red[gi - g_lo][wave][0] = a; s_red[gi - g_lo][wave][1] = q; }
  }
  __syncthreads();
  if (tid <= g_hi - g_lo) {
    const double a = ((s_red[tid][0][0] + s_red[tid][1][0]) + s_red[tid][2][0]) + s_red[tid][3][0];
    const double q = ((s_red[tid][0][1] + s_red[tid][1][1]) + s_red[tid][2][1]) + s_red[tid][3][1];
    const double cnt = (double)cg * V, mean = a / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0) var = 0;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  // the block's 8 channels as affine forms (a, b): x -> a x + b, one FMA per element in the loop below (no per-element group
  // lookup, no integer division); channels >= C give 0
  __shared__ float s_ab[8][2];
  if (tid < 8) {
    const int ch = c8 * 8 + tid;
    float a = 0.f, bsh = 0.f;
    if (ch < C) {
      const int g = ch / cg - g_lo;
      a = gamma[ch] * s_rstd[g];
      bsh = beta[ch] - s_mean[g] * a;
    }
    s_ab[tid][0] = a;
    s_ab[tid][1] = bsh;
  }
  __syncthreads();
  float ca[8], cb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ca[j] = s_ab[j][0]; cb[j] = s_ab[j][1]; }
  const float *xb = x + ((size_t)bi * C + c8 * 8) * V;
  const int nch = min(8, C - c8 * 8);
  bool sat = false;
#pragma unroll 2
  for (int it = 0; it < VB / 256; ++it) {
    const int v = blockIdx.x * VB + it * 256 + tid;
    if (v >= V) break;
    float val[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = xb[(size_t)min(j, nch - 1) * V + v] * ca[j] + cb[j];  // rows >= C: a = b = 0
      if (act == 1) t = swishf(t);
      val[j] = t * act_scale;  // a power of two: exact
      sat |= !(fabsf(val[j]) <= 65504.f);
    }
    store_h2(out + ((size_t)bi * C8 + c8) * 2 * (size_t)V * 8, (size_t)v, (size_t)V, val);
  }
  if (saturated != nullptr && __ballot(sat) != 0ull && lane == __ffsll((long long)__ballot(sat)) - 1) atomicOr(saturated, 1u);
}

extern "C" int bdm_group_norm_to_h2_stats(int b, int c, int v, int groups, const float *x, const float *gamma,
                                          const float *beta, float eps, int act, float act_scale, void *out_h2,
                                          const void *partial, int slices, unsigned int *saturated, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && v >= 1 && groups >= 1 && c % groups == 0 && (c / groups) >= 4 && partial != nullptr && slices >= 1,
              "group_norm_to_h2_stats: bad arguments");
  {
    int ex = 0;
    BDM_REQUIRE(act_scale > 0.f && act_scale < INFINITY && frexpf(act_scale, &ex) == 0.5f,
                "group_norm_to_h2_stats: act_scale must be a power of two (got %g)", (double)act_scale);
  }
  if (b == 0) return BDM_OK;
  constexpr int VB = 1024;
  dim3 grid(cdiv(v, VB), (c + 7) / 8, b);
  hipLaunchKernelGGL(to_h2_stats_kernel<VB>, grid, dim3(256), 0, (hipStream_t)stream, c, v, groups, slices, x,
                     (const double *)partial, gamma, beta, eps, act, act_scale, (unsigned short *)out_h2, saturated);
  return launch_status("group_norm_to_h2_stats");
}

// The same producer for the COMPACT output of the first convolution (sparse_conv_os.hip): x holds one row of C floats per entry
// of the dilated voxel list, (B, n_dil_max, C); dil_index[v] = row of voxel v or -1, and every voxel outside the list is `bias`
// (the convolution of zeros) -- its record is the same for the whole grid and is formed once per thread.  Reads 23 % of the grid
// (32^3, Gaussian-like cloud) instead of all of it; the dense fp32 grid is never written.
template <int VB>
__global__ __launch_bounds__(256) void to_h2_stats_compact_kernel(int C, int V, int G, int S, int n_dil_max, const float *__restrict__ xc,
                                                                  const int *__restrict__ dil_index, const float *__restrict__ bias,
                                                                  const double *__restrict__ partial, const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, float eps, int act, float act_scale,
                                                                  unsigned short *__restrict__ out, unsigned *__restrict__ saturated) {
  __shared__ float s_mean[2], s_rstd[2];
  __shared__ double s_red[2][4][2];
  const int bi = blockIdx.z, c8 = blockIdx.y, C8 = gridDim.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = C / G;
  const int g_lo = (c8 * 8) / cg, g_hi = min((c8 * 8 + 7) / cg, G - 1);
  for (int gi = g_lo; gi <= g_hi; ++gi) {
    const double *pp = partial + ((size_t)bi * G + gi) * S * 2;
    double a = 0.0, q = 0.0;
    for (int sl = tid; sl < S; sl += 256) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    a = wave_sum_bfly(a); q = wave_sum_bfly(q);   // (DPP + readlanes, bit-identical to the 64-lane xor butterfly)
    if (lane == 0) { s_red[gi - g_lo][wave][0] = a; s_red[gi - g_lo][wave][1] = q; }
  }
  __syncthreads();
  if (tid <= g_hi - g_lo) {
    const double a = ((s_red[tid][0][0] + s_red[tid][1][0]) + s_red[tid][2][0]) + s_red[tid][3][0];
    const double q = ((s_red[tid][0][1] + s_red[tid][1][1]) + s_red[tid][2][1]) + s_red[tid][3][1];
    const double cnt = (double)cg * V, mean = a / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0) var = 0;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  __shared__ float s_ab[8][3];   // affine form (a, b) of the GroupNorm per channel, and the channel's bias
  if (tid < 8) {
    const int ch = c8 * 8 + tid;
    float a = 0.f, bsh = 0.f, bv = 0.f;
    if (ch < C) {
      const int g = ch / cg - g_lo;
      a = gamma[ch] * s_rstd[g];
      bsh = beta[ch] - s_mean[g] * a;
      bv = bias ? bias[ch] : 0.f;
    }
    s_ab[tid][0] = a;
    s_ab[tid][1] = bsh;
    s_ab[tid][2] = bv;
  }
  __syncthreads();
  float ca[8], cb[8], fill[8];
  bool sat = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ca[j] = s_ab[j][0]; cb[j] = s_ab[j][1];
    float t = s_ab[j][2] * ca[j] + cb[j];
    if (act == 1) t = swishf(t);
    fill[j] = t * act_scale;
    sat |= !(fabsf(fill[j]) <= 65504.f);
  }
  const int nch = min(8, C - c8 * 8);
  const bool vec = (C & 3) == 0 && nch == 8;   // 16-byte pieces of a row
  const int *di = dil_index + (size_t)bi * V;
  const float *xb = xc + (size_t)bi * n_dil_max * C + c8 * 8;
#pragma unroll 2
  for (int it = 0; it < VB / 256; ++it) {
    const int v = blockIdx.x * VB + it * 256 + tid;
    if (v >= V) break;
    const int j = di[v];
    float val[8];
    if (j >= 0) {
      const float *row = xb + (size_t)j * C;
      float in[8];
      if (vec) {
        const float4 p = *reinterpret_cast<const float4 *>(row), q = *reinterpret_cast<const float4 *>(row + 4);
        in[0] = p.x; in[1] = p.y; in[2] = p.z; in[3] = p.w; in[4] = q.x; in[5] = q.y; in[6] = q.z; in[7] = q.w;
      } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) in[u] = row[min(u, nch - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        float t = in[u] * ca[u] + cb[u];   // channels >= C: a = b = 0
        if (act == 1) t = swishf(t);
        val[u] = t * act_scale;
        sat |= !(fabsf(val[u]) <= 65504.f);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) val[u] = fill[u];
    }
    store_h2(out + ((size_t)bi * C8 + c8) * 2 * (size_t)V * 8, (size_t)v, (size_t)V, val);
  }
  if (saturated != nullptr && __ballot(sat) != 0ull && lane == __ffsll((long long)__ballot(sat)) - 1) atomicOr(saturated, 1u);
}

extern "C" int bdm_group_norm_to_h2_stats_compact(int b, int c, int v, int groups, const float *xc, int n_dil_max, const int *dil_index,
                                                  const float *bias, const float *gamma, const float *beta, float eps, int act,
                                                  float act_scale, void *out_h2, const void *partial, int slices,
                                                  unsigned int *saturated, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && v >= 1 && groups >= 1 && c % groups == 0 && (c / groups) >= 4 && partial != nullptr && slices >= 1 &&
                  xc != nullptr && dil_index != nullptr && n_dil_max >= 1,
              "group_norm_to_h2_stats_compact: bad arguments");
  {
    int ex = 0;
    BDM_REQUIRE(act_scale > 0.f && act_scale < INFINITY && frexpf(act_scale, &ex) == 0.5f,
                "group_norm_to_h2_stats_compact: act_scale must be a power of two (got %g)", (double)act_scale);
  }
  if (b == 0) return BDM_OK;
  constexpr int VB = 1024;
  dim3 grid(cdiv(v, VB), (c + 7) / 8, b);
  hipLaunchKernelGGL(to_h2_stats_compact_kernel<VB>, grid, dim3(256), 0, (hipStream_t)stream, c, v, groups, slices, n_dil_max, xc, dil_index,
                     bias, (const double *)partial, gamma, beta, eps, act, act_scale, (unsigned short *)out_h2, saturated);
  return launch_status("group_norm_to_h2_stats_compact");
}

extern "C" int bdm_group_norm_stats(int b, int c, int l, int groups, const float *x, long long bs_x, void *workspace,
                                    int *slices_out, void *stream);

extern "C" int bdm_group_norm_to_h2(int b, int c, int v, int groups, const float *x, const float *gamma,
                                    const float *beta, float eps, int act, float act_scale, void *out_h2, void *workspace,
                                    unsigned int *saturated, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && v >= 1, "group_norm_to_h2: bad sizes");
  {
    int ex = 0;
    BDM_REQUIRE(act_scale > 0.f && act_scale < INFINITY && frexpf(act_scale, &ex) == 0.5f,
                "group_norm_to_h2: act_scale must be a power of two (got %g)", (double)act_scale);
  }
  BDM_REQUIRE(groups == 0 || (c % groups == 0 && groups <= 64 && workspace != nullptr), "group_norm_to_h2: bad groups %d", groups);
  if (b == 0) return BDM_OK;
  int S = 0;
  if (groups > 0) {
    int rc = bdm_group_norm_stats(b, c, v, groups, x, (long long)c * v, workspace, &S, stream);
    if (rc) return rc;
  }
  dim3 grid(cdiv(v, 256), (c + 7) / 8, b);
  hipLaunchKernelGGL(to_h2_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, v, groups > 0 ? groups : 1, S, x,
                     groups > 0 ? (const double *)workspace : nullptr, gamma, beta, eps, act, act_scale, (unsigned short *)out_h2,
                     saturated);
  return launch_status("group_norm_to_h2");
}
