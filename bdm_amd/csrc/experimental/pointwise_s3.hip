// experimental/pointwise_s3.hip (built with `make EXPERIMENTAL=1` only; opt-in at run time with ops.PW_IMPL = "bf16x6").  MEASURED (round 3, B = 16,
// tools/forward_rows.py): no gain over the fp32-MFMA kernel -- 1272 vs 1271 us per forward over the 38 GEMMs; +7 % on the long-K layer
// (128 x 579 x 4096: 120 -> 112 us), slower on the narrow ones (32 x 390 x 4096: 49 -> 73 us).  The 1x1 GEMMs are bound by operand
// staging (global -> LDS) and not by the matrix pipe, so 2.7x the matrix rate buys nothing.  Kept as a negative result.
// -- the 1x1 convolutions (SharedMLP / FP-module MLPs / attention projections / classifier:
// modules/shared_mlp.py:25-30, pvconv.py:21-31, pvcnn.py:62-69) as a bf16x6 GEMM on the 16-bit matrix pipe.
//
//   Y[b] (M x N) = W (M x K) . X'[b] (K x N) + bias ...       X' = X or Swish(GroupNorm(X)) (folded, as in dense_ops.hip)
//
// Why: v_mfma_f32_32x32x2_f32 (dense_ops.hip) peaks at 157 TFLOP/s and needs one LDS read per lane for every TWO k; the long-K
// layers run at 50-85 TFLOP/s against that peak and the short-K ones are bound by LDS reads and MFMA issue slots, not by HBM.
// v_mfma_f32_32x32x16_bf16 takes 16 k per instruction from one 16-byte LDS read per lane.  Every fp32 operand is split EXACTLY
// into three bf16 terms (s3_split.h: hi + mid + lo, 24 mantissa bits, any exponent: no scale, no range limit) and the six
// partial products that matter are accumulated in fp32, smallest first -- the arithmetic of conv3d_s3.hip (error ~2^-22
// relative per product instead of 2^-24; <= 2e-6 vs fp64 on the layers' shapes, tests/test_hip_dense.py).  Six products on a
// pipe 16x faster: 2.7x the matrix rate of the fp32 kernel, a quarter of its LDS read instructions.
//
//   A = weights, split ONCE at pack time into records [ceil(K/8)][3][M] of 8 consecutive k (bdm_pointwise_s3_pack_weights);
//   B = activations, read channel-first (coalesced along the points), transformed (folded GroupNorm + Swish) and split while they
//       go to LDS as records [k/8][3][column] -- the thread that owns (k-group, column) loads its 8 rows' values itself, so the
//       8-k record needs no transposition pass;
//   tile (32 MI) x (128 NI), four waves side by side along N, 32 (16 for the 256-column tiles) k per stage, register-prefetched (same structure, same epilogue
//   -- bias, per-shape bias, activation, residual, canonical GroupNorm partials, per-shape amax -- as pw_gemm_kernel).
#include <stdlib.h>

#include "../../../include/bdm_hip.h"
#include "../common.h"
#include "../pointwise_common.h"
#include "../s3_split.h"

using namespace bdm;

// ---- weights -> records -------------------------------------------------------------------------------------------------
__global__ void pw_s3_pack_kernel(int M, int K, int ldw, const float *__restrict__ w, unsigned short *__restrict__ out) {
  const int G = (K + 7) / 8;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < G * M; e += gridDim.x * blockDim.x) {
    const int g = e / M, m = e % M;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = g * 8 + j < K ? w[(size_t)m * ldw + g * 8 + j] : 0.f;
    store_s3(out + (size_t)g * 3 * M * 8, (size_t)m, (size_t)M, v);
  }
}

extern "C" size_t bdm_pointwise_s3_weight_elems(int m, int k) { return (size_t)((k + 7) / 8) * 3 * m * 8; }

extern "C" int bdm_pointwise_s3_pack_weights(int m, int k, const float *w, int ldw, void *packed, void *stream) {
  BDM_REQUIRE(m >= 1 && k >= 1 && ldw >= k && w != nullptr && packed != nullptr, "pointwise_s3_pack_weights: bad arguments");
  hipLaunchKernelGGL(pw_s3_pack_kernel, dim3(cdiv(((k + 7) / 8) * m, 256)), dim3(256), 0, (hipStream_t)stream, m, k, ldw, w,
                     (unsigned short *)packed);
  return launch_status("pointwise_s3_pack_weights");
}

// ---- the GEMM --------------------------------------------------------------------------------------------------------------
template <int MI, int NI, bool FOLD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void pw_s3_kernel(
    int M, int K, int N, const uint4 *__restrict__ Wp, const float *__restrict__ X, long long bsx, int ldx,
    const float *__restrict__ bias, const float *__restrict__ bbias, int ldbb, const float *__restrict__ R, long long bsr, int ldr,
    float *__restrict__ Y, long long bsy, int ldy, int act, float slope, PwGn gn) {
  constexpr int BM = 32 * MI, BN = 128 * NI;
  constexpr int GS = NI == 2 ? 2 : 4;  // 8-k record groups per stage: 32 k for the 128-column tiles, 16 k for the 256-column ones (LDS)
  __shared__ uint4 As[3 * GS * BM];  // [group-in-stage * 3 + split][row]
  __shared__ uint4 Bs[3 * GS * BN];  // [group-in-stage * 3 + split][column]
  __shared__ float2 coef_s[FOLD ? 1024 : 1];  // per input channel: swish(coef.x * x + coef.y) = Swish(GroupNorm(x))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, bi = blockIdx.z;
  const int G = (K + 7) / 8;
  if constexpr (FOLD) {  // statistics of the input's groups from the producer's slice partials (as pw_gemm_kernel)
    __shared__ float s_mr[16];
    const int g = tid >> 5, l = tid & 31, cgi = K / gn.in_G;
    double a = 0.0, q = 0.0;
    if (g < gn.in_G) {
      const double *pp = gn.in_partial + ((size_t)bi * gn.in_G + g) * gn.in_S * 2;
      for (int sl = l; sl < gn.in_S; sl += 32) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
    if (l == 0 && g < gn.in_G) {
      const double cnt = (double)cgi * N, mean = a / cnt;
      double var = q / cnt - mean * mean;
      if (var < 0) var = 0;
      s_mr[2 * g] = (float)mean;
      s_mr[2 * g + 1] = (float)(1.0 / sqrt(var + (double)gn.in_eps));
    }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
      const int gk = k / cgi;
      const float ak = gn.in_gamma[k] * s_mr[2 * gk + 1];
      coef_s[k] = make_float2(ak, gn.in_beta[k] - s_mr[2 * gk] * ak);
    }
    // (visible to every wave after the first barrier of the K loop)
  }
  const float *Xb = X + (size_t)bi * bsx;
  const float *X2b = gn.x2 ? gn.x2 + (size_t)bi * gn.bsx2 : nullptr;
  const int k1 = X2b ? gn.k1 : K;  // rows [k1, K) live in X2b
  float *Yb = Y + (size_t)bi * bsy;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int c = 0; c < NI; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // register-prefetched staging; loads carry neither a branch nor a select on the loaded value (clamped addresses; rows >= M and
  // columns >= N only feed outputs that are never stored, k >= K is zeroed on the way to LDS)
  constexpr int AI = (3 * GS * BM + 255) / 256, BI = (GS * BN) / 256;
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v ar[AI];
  float xr[BI][8];
  unsigned b_col[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) b_col[i] = (unsigned)min(n0 + (tid + i * 256) % BN, N - 1);
  auto load_stage = [&](int g0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = min(tid + i * 256, 3 * GS * BM - 1), row = e % BM, gs = e / BM, g = min(g0 + gs / 3, G - 1), sp = gs % 3;
      ar[i] = *reinterpret_cast<const u32x4v *>(&Wp[(unsigned)((g * 3 + sp) * M + min(m0 + row, M - 1))]);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int g = g0 + (tid + i * 256) / BN;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = min(g * 8 + j, K - 1);
        const float *rowp = kk < k1 ? Xb + (unsigned)kk * (unsigned)ldx : X2b + (unsigned)(kk - k1) * (unsigned)gn.ldx2;
        xr[i][j] = rowp[b_col[i]];
      }
    }
  };
  load_stage(0);
  for (int g0 = 0; g0 < G; g0 += GS) {
    __syncthreads();
    const u32x4v zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256, gs = e / BM;
      if (e < 3 * GS * BM) *reinterpret_cast<u32x4v *>(&As[e]) = (g0 + gs / 3 < G) ? ar[i] : zero;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e = tid + i * 256, c = e % BN, gl = e / BN, g = g0 + gl;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = g * 8 + j;
        float t = xr[i][j];
        if constexpr (FOLD) {
          float2 cf = coef_s[min(k, K - 1)];
          lds_settle(cf.x, cf.y);  // (common.h)
          t = swishf(cf.x * t + cf.y);
        }
        v[j] = k < K ? t : 0.f;
      }
      store_s3(reinterpret_cast<unsigned short *>(Bs + (size_t)gl * 3 * BN), (size_t)c, (size_t)BN, v);
    }
    __syncthreads();
    if (g0 + GS < G) load_stage(g0 + GS);
#pragma unroll
    for (int kk = 0; kk < GS / 2; ++kk) {
      bf16x8 a[MI][3], b[NI][3];
#pragma unroll
      for (int x = 0; x < MI; ++x)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) a[x][sp] = *reinterpret_cast<const bf16x8 *>(&As[((2 * kk + lh) * 3 + sp) * BM + x * 32 + li]);
#pragma unroll
      for (int y = 0; y < NI; ++y)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
          b[y][sp] = *reinterpret_cast<const bf16x8 *>(&Bs[((2 * kk + lh) * 3 + sp) * BN + (wave * NI + y) * 32 + li]);
      // smallest terms first; term-major so that consecutive MFMAs never depend on each other
#pragma unroll
      for (int term = 0; term < 6; ++term) {
        constexpr int ta[6] = {1, 2, 0, 1, 0, 0}, tb[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
        for (int x = 0; x < MI; ++x)
#pragma unroll
          for (int y = 0; y < NI; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[x][ta[term]], b[y][tb[term]], acc[x][y], 0, 0, 0);
      }
    }
  }
  // ---- epilogue (pw_gemm_kernel's): C/D map  row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
  float *Asf = reinterpret_cast<float *>(As);
  float bs[MI][4], bq[MI][4];
  float am[MI];
  constexpr int NBW = MI * NI * 2 * 4 * 2;  // per wave: [x][y][lh][j][stat]
  float *red = reinterpret_cast<float *>(Bs);
  if (gn.out_partial != nullptr || gn.amax != nullptr) __syncthreads();  // slower waves may still read operand fragments
#pragma unroll
  for (int x = 0; x < MI; ++x) {
    am[x] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { bs[x][j] = 0.f; bq[x][j] = 0.f; }
  }
#pragma unroll
  for (int x = 0; x < MI; ++x) {
    float badd[16], bb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) badd[r] = bb[r] = 0.f;
    if (bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) badd[r] = bias[min(m0 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1)];
    }
    if (bbias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bb[r] = bbias[(size_t)bi * ldbb + min(m0 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1)];
    }
#pragma unroll
    for (int y = 0; y < NI; ++y) {
      const int n = n0 + (wave * NI + y) * 32 + li;
      const int nn = min(n, N - 1);
      float rv[16];
      if (R) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[r] = R[(size_t)bi * bsr + (size_t)min(m0 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, M - 1) * ldr + nn];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = (acc[x][y][r] + badd[r]) + bb[r];
        if (act == 2) v = v > 0.f ? v : v * slope;
        else if (act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (R) v += rv[r];
        if (m < M && n < N) {
          Yb[(size_t)m * ldy + n] = v;
          bs[x][r >> 2] += v;
          bq[x][r >> 2] = __builtin_fmaf(v, v, bq[x][r >> 2]);  // explicitly fused: the same rounding in every tile variant
          am[x] = fmaxf(am[x], fabsf(v));
        }
      }
      if (gn.out_partial != nullptr) {  // canonical GroupNorm partials: one 32 x 32 sub-tile at a time (see pw_gemm_kernel)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) {
            bs[x][j] += __shfl_xor(bs[x][j], o, 64);
            bq[x][j] += __shfl_xor(bq[x][j], o, 64);
          }
          if (li == 0) {
            red[wave * NBW + (((x * NI + y) * 2 + lh) * 4 + j) * 2 + 0] = bs[x][j];
            red[wave * NBW + (((x * NI + y) * 2 + lh) * 4 + j) * 2 + 1] = bq[x][j];
          }
          bs[x][j] = 0.f;
          bq[x][j] = 0.f;
        }
      }
    }
  }
  if (gn.amax != nullptr) {
#pragma unroll
    for (int x = 0; x < MI; ++x) {
      const float mx = wave_max(am[x]);
      if (lane == 0) Asf[x * 4 + wave] = mx;
    }
    __syncthreads();
    if (tid < MI && m0 + tid * 32 < M) {
      const float mx = fmaxf(fmaxf(Asf[tid * 4], Asf[tid * 4 + 1]), fmaxf(Asf[tid * 4 + 2], Asf[tid * 4 + 3]));
      unsigned *slot = gn.amax + (size_t)bi * gn.amax_slots + (m0 + tid * 32) / gn.amax_rows;
      if (mx > 0.f && __float_as_uint(mx) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(slot, __float_as_uint(mx));
    }
  }
  if (gn.out_partial != nullptr) {
    __syncthreads();
    constexpr int NC = NI * MI * 2 * 4 * 2;  // [cb][x][lh][j][stat]
    if (tid < NC) {
      const int stat = tid & 1, j = (tid >> 1) & 3, lhh = (tid >> 3) & 1, x = (tid >> 4) % MI, cb = tid / (16 * MI);
      float sum = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int g = cb * 4 + st;
        const float v = red[(g / NI) * NBW + (((x * NI + (g % NI)) * 2 + lhh) * 4 + j) * 2 + stat];
        sum = st == 0 ? v : sum + v;
      }
      red[4 * NBW + tid] = sum;
    }
    __syncthreads();
    const int cg = gn.out_cg, Gn = M / cg;
    const int ngt = cg >= 32 ? 1 : 32 / cg;
    const int rt = cg >= 32 ? cg / 32 : 1;
    if (tid < NI * MI * ngt) {
      const int gi = tid % ngt, x = (tid / ngt) % MI, cb = tid / (ngt * MI);
      const int row0 = m0 + x * 32, colblock = blockIdx.x * NI + cb, ncb = (N + 127) / 128;
      if (row0 + gi * cg < M && colblock < ncb) {
        double a = 0.0, qq = 0.0;
        for (int hh = 0; hh < 2; ++hh)
          for (int j = 0; j < 4; ++j)
            if (cg >= 32 || ((8 * j + 4 * hh) >> (__ffs(cg) - 1)) == gi) {
              a += (double)red[4 * NBW + (((cb * MI + x) * 2 + hh) * 4 + j) * 2 + 0];
              qq += (double)red[4 * NBW + (((cb * MI + x) * 2 + hh) * 4 + j) * 2 + 1];
            }
        const int g = row0 / cg + gi, S = ncb * rt, sl = colblock * rt + (row0 / 32) % rt;
        double *dst = gn.out_partial + (((size_t)bi * Gn + g) * S + sl) * 2;
        dst[0] = a;
        dst[1] = qq;
      }
    }
  }
}

template <bool FOLD>
static void pw_s3_dispatch(int b, int m, int k, int n, const void *packed_w, const float *x, long long bs_x, int ld_x,
                           const float *bias, const float *batch_bias, int ld_bb, const float *residual, long long bs_r, int ld_r,
                           float *y, long long bs_y, int ld_y, int act, float slope, PwGn gn, hipStream_t s) {
#define S3_LAUNCH(MI, NI)                                                                                                        \
  hipLaunchKernelGGL((pw_s3_kernel<MI, NI, FOLD>), dim3(cdiv(n, 128 * NI), cdiv(m, 32 * MI), b), dim3(256), 0, s, m, k, n,      \
                     (const uint4 *)packed_w, x, bs_x, ld_x, bias, batch_bias, ld_bb, residual, bs_r, ld_r, y, bs_y, ld_y, act,  \
                     slope, gn)
  int mi, ni, bk;
  bdm_pw_tile(b, m, k, n, &mi, &ni, &bk);
  if (mi == 1 && ni == 1) S3_LAUNCH(1, 1);
  else if (mi == 1) S3_LAUNCH(1, 2);
  else if (ni == 2) S3_LAUNCH(2, 2);
  else S3_LAUNCH(2, 1);
#undef S3_LAUNCH
}

extern "C" int bdm_pointwise_conv_s3(int b, int m, int k, int n, const void *packed_w, const float *x, long long bs_x, int ld_x,
                                     const float *bias, const float *batch_bias, int ld_bb, const float *residual,
                                     long long bs_r, int ld_r, float *y, long long bs_y, int ld_y, int act, float slope,
                                     void *stream) {
  BDM_REQUIRE(b >= 0 && m >= 1 && k >= 1 && n >= 1 && packed_w != nullptr, "pointwise_conv_s3: bad sizes m=%d k=%d n=%d", m, k, n);
  BDM_REQUIRE(act == 0 || act == 2 || act == 3, "pointwise_conv_s3: act must be 0, 2 (LeakyReLU) or 3 (GELU)");
  BDM_REQUIRE((long long)k * ld_x + n < (1ll << 31), "pointwise_conv_s3: the operand spans more than 2^31 elements");
  if (b == 0) return BDM_OK;
  pw_s3_dispatch<false>(b, m, k, n, packed_w, x, bs_x, ld_x, bias, batch_bias, ld_bb, residual, bs_r, ld_r, y, bs_y, ld_y, act, slope,
                        PwGn{}, (hipStream_t)stream);
  return launch_status("pointwise_conv_s3");
}

extern "C" int bdm_pointwise_conv_gn_s3(int b, int m, int k, int n, const void *packed_w, const float *x, long long bs_x, int ld_x,
                                        const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y,
                                        long long bs_y, int ld_y, const void *in_partial, int in_slices, int in_groups,
                                        const float *in_gamma, const float *in_beta, float in_eps, int out_groups,
                                        void *out_partial, float *amax, int amax_rows, void *stream) {
  BDM_REQUIRE(b >= 0 && m >= 1 && k >= 1 && n >= 1 && packed_w != nullptr, "pointwise_conv_gn_s3: bad sizes m=%d k=%d n=%d", m, k, n);
  BDM_REQUIRE((long long)k * ld_x + n < (1ll << 31), "pointwise_conv_gn_s3: the operand spans more than 2^31 elements");
  PwGn gn{};
  if (x2 != nullptr) {
    BDM_REQUIRE(k1 >= 1 && k1 < k && (long long)(k - k1) * ld_x2 + n < (1ll << 31) && in_partial == nullptr,
                "pointwise_conv_gn_s3: second source needs 1 <= k1 < k and no input fold (k=%d k1=%d)", k, k1);
    gn.x2 = x2; gn.bsx2 = bs_x2; gn.ldx2 = ld_x2; gn.k1 = k1;
  }
  if (in_partial != nullptr) {
    BDM_REQUIRE(in_groups >= 1 && in_groups <= 8 && k % in_groups == 0 && k <= 1024 && in_slices >= 1 && in_gamma && in_beta,
                "pointwise_conv_gn_s3: input fold needs <= 8 groups dividing k <= 1024 (k=%d groups=%d)", k, in_groups);
    gn.in_partial = (const double *)in_partial;
    gn.in_S = in_slices; gn.in_G = in_groups; gn.in_gamma = in_gamma; gn.in_beta = in_beta; gn.in_eps = in_eps;
  }
  if (amax != nullptr) {
    BDM_REQUIRE(amax_rows >= 32 && amax_rows % 32 == 0, "pointwise_conv_gn_s3: amax_rows must be a multiple of 32 (got %d)", amax_rows);
    gn.amax = (unsigned *)amax; gn.amax_rows = amax_rows; gn.amax_slots = (m + amax_rows - 1) / amax_rows;
  }
  if (out_partial != nullptr) {
    const int cg = out_groups >= 1 && m % out_groups == 0 ? m / out_groups : 0;
    BDM_REQUIRE(cg >= 4 && (cg & (cg - 1)) == 0, "pointwise_conv_gn_s3: output statistics need a power-of-two >= 4 channels per group "
                "(m=%d groups=%d)", m, out_groups);
    gn.out_partial = (double *)out_partial; gn.out_cg = cg;
  }
  if (b == 0) return BDM_OK;
  if (in_partial != nullptr)
    pw_s3_dispatch<true>(b, m, k, n, packed_w, x, bs_x, ld_x, bias, nullptr, 0, nullptr, 0ll, 0, y, bs_y, ld_y, 0, 0.f, gn, (hipStream_t)stream);
  else
    pw_s3_dispatch<false>(b, m, k, n, packed_w, x, bs_x, ld_x, bias, nullptr, 0, nullptr, 0ll, 0, y, bs_y, ld_y, 0, 0.f, gn, (hipStream_t)stream);
  return launch_status("pointwise_conv_gn_s3");
}
