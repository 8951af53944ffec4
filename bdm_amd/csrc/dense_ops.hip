// dense_ops.hip -- gfx950 kernels for the dense per-point / per-voxel operators the reference
// runs as stock PyTorch modules around its plugin inside one denoiser forward:
//   Conv1d/Conv2d k=1 (SharedMLP, attention projections, zero-conv projections, classifier)
//       -> f32-input MFMA GEMM  (modules/shared_mlp.py:25-30, pvconv.py:21-31, pvcnn_fuse.py:111-123)
//   GroupNorm(8) [+ residual] [+ Swish]      (shared_mlp.py:27-29, pvconv.py:80-86,61)
//   max over neighbours                      (pointnet.py:86)
//   SE3d gate                                (se.py:8-19)
//   timestep embedding + embedf              (pvcnn_utils.py:171-185, pvcnn.py:72-76)
//   Voxelization.forward coordinate maths    (modules/voxelization.py:16-25)
//   trilinear devoxelise * SE gate + point branch  (pvconv.py:95-96)
//   attention cores                          (pvconv.py:40-63)
// All arithmetic is fp32; contractions use v_mfma_f32_32x32x2_f32 (exact fp32 products, k-ordered
// fp32 accumulation -- the 1e-3 end-to-end criterion after 1000 steps rules out bf16 here).
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"
#include "se_fc.h"

using namespace bdm;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// =====================================================================================
// Pointwise convolution = batched GEMM  Y[b] = W (M x K) * X[b] (K x N) + bias
// =====================================================================================
// Channel-first operands put the point index on the MFMA lane: B-operand rows are
// contiguous point runs (coalesced 16-B loads, conflict-free ds_read_b32), the weight tile is
// stored k-major in LDS so A-operand reads are conflict-free too.
// Tile: (32*MI) x (128*NI) per 256-thread workgroup, K chunk 16, 4 waves side by side along N.
// K chunk BK per barrier pair: 16 for the throughput shapes; 64 for the latency-bound ones (few workgroups, long K), where
// one chunk in flight per workgroup leaves the global-load latency exposed on every step.  Same k order: same bits.
// GroupNorm folding around a 1x1 convolution (SharedMLP = [conv -> GroupNorm(8) -> Swish]*): the GEMM can (a) leave the
// (sum, sum of squares) of the values it writes, per GroupNorm group, as slice partials -- the normalisation that follows then
// needs no statistics pass -- and (b) take its INPUT as the raw output of such a convolution plus that convolution's partials,
// applying normalise + Swish while the operand goes to LDS -- the normalised tensor is never written.  Deterministic: fixed
// summation orders everywhere (no float atomics).
#include "pointwise_common.h"

// amdgpu_waves_per_eu(2): with an occupancy target of two waves per SIMD the register allocator stops hoarding (2 x 2 tile with
// the folded GroupNorm: 220 -> 160 VGPRs, i.e. three workgroups per CU instead of two; the 32768-column SA layer 101 -> 79 us);
// a target of three spills the 32- and 64-deep K variants.
template <int MI, int NI, bool ATRANS = false, int BK = 16, bool FOLD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void pw_gemm_kernel(int M, int K, int N, const float *__restrict__ W, int ldw,
                                                      long long bsw, const int *__restrict__ m_count,
                                                      const float *__restrict__ X, long long bsx, int ldx,
                                                      const float *__restrict__ bias,
                                                      const float *__restrict__ bbias, int ldbb,
                                                      const float *__restrict__ R, long long bsr, int ldr,
                                                      float *__restrict__ Y, long long bsy, int ldy, int act,
                                                      float slope, PwGn gn) {
  constexpr int BM = 32 * MI, BN = 128 * NI, LDA = BM + 4;
  __shared__ float As[BK * LDA];
  __shared__ __align__(16) float Bs[BK * BN];
  __shared__ float2 coef_s[FOLD ? 1024 : 1];  // per input channel: swish(coef.x * x + coef.y) = Swish(GroupNorm(x))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, bi = blockIdx.z;
  if (m_count && m0 >= m_count[bi]) return;  // rows beyond this shape's live count (sparse convolution GEMM)
  if constexpr (FOLD) {
    // statistics of the input's groups from the producer's slice partials: 32 lanes per group (<= 8 groups), each lane adds
    // slices l, l + 32, ... in order, then a fixed butterfly
    __shared__ float s_mr[16];
    const int g = tid >> 5, l = tid & 31, cgi = K / gn.in_G;
    double a = 0.0, q = 0.0;
    if (g < gn.in_G) {
      const double *pp = gn.in_partial + ((size_t)bi * gn.in_G + g) * gn.in_S * 2;
      for (int sl = l; sl < gn.in_S; sl += 32) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    }
    a = half32_sum(a); q = half32_sum(q);   // (the 32 lanes of this group: DPP, bit-identical to the xor butterfly)
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
  // the tile's per-row terms (bias, per-shape bias) go to LDS now: fetched in the epilogue they were one more dependent global round trip
  // between a workgroup's last MFMA and its stores (visible to every wave after the first barrier of the K loop)
  __shared__ float s_rowterm[2][BM];
  if (tid < BM) {
    const int mm = min(m0 + tid, M - 1);
    s_rowterm[0][tid] = bias ? bias[mm] : 0.f;
    s_rowterm[1][tid] = bbias ? bbias[(size_t)bi * ldbb + mm] : 0.f;
  }
  W += (size_t)bi * bsw;
  const float *Xb = X + (size_t)bi * bsx;
  const float *X2b = gn.x2 ? gn.x2 + (size_t)bi * gn.bsx2 : nullptr;
  const int k1 = X2b ? gn.k1 : K;  // rows [k1, K) live in X2b
  float *Yb = Y + (size_t)bi * bsy;
  const bool vec_ok = ((ldx & 3) == 0) && ((((uintptr_t)Xb) & 15) == 0) &&
                      (!X2b || (((gn.ldx2 & 3) == 0) && ((((uintptr_t)X2b) & 15) == 0)));

  f32x16 acc[MI][NI];
#pragma unroll
  for (int a = 0; a < MI; ++a)
#pragma unroll
    for (int c = 0; c < NI; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // register-prefetched staging: the global loads of K-chunk c+1 are in flight during the MFMAs of chunk c.  The loads
  // carry no branch and no select: rows >= M and columns >= N are read from a clamped (valid) address -- they only feed
  // output elements the epilogue never stores -- and k >= K (last chunk) is read clamped and zeroed when the registers go to
  // LDS.  (A branch or a select on a freshly loaded value makes the wave wait for memory inside the load phase.)
  constexpr int AI = (BM * BK) / 256, BI = (BK * BN / 4) / 256;
  float ar[AI];
  float4 br[BI];
  const bool vec_all = vec_ok && (N & 3) == 0;  // every 4-column piece is entirely inside or outside the matrix
  unsigned a_off[AI], b_col[BI];  // loop-invariant part of each staged element's address
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int e = tid + i * 256, m = ATRANS ? e % BM : e / BK;
    const int mm = min(m0 + m, M - 1);
    a_off[i] = ATRANS ? (unsigned)mm : (unsigned)mm * (unsigned)ldw;
  }
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int e = tid + i * 256, c4 = (e % (BN / 4)) * 4;
    b_col[i] = (unsigned)(n0 + c4);
  }
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256, k = ATRANS ? e / BM : e % BK;
      const unsigned kk = (unsigned)min(k0 + k, K - 1);
      ar[i] = W[a_off[i] + (ATRANS ? kk * (unsigned)ldw : kk)];
    }
    if (vec_all) {
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int e = tid + i * 256, k = e / (BN / 4);
        const int kk = min(k0 + k, K - 1);
        const float *rowp = kk < k1 ? Xb + (unsigned)kk * (unsigned)ldx : X2b + (unsigned)(kk - k1) * (unsigned)gn.ldx2;
        br[i] = *reinterpret_cast<const float4 *>(rowp + min(b_col[i], (unsigned)(N - 4)));
      }
    } else {
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const int e = tid + i * 256, k = e / (BN / 4);
        const int kk = min(k0 + k, K - 1);
        const float *src = kk < k1 ? Xb + (unsigned)kk * (unsigned)ldx : X2b + (unsigned)(kk - k1) * (unsigned)gn.ldx2;
        const unsigned last = (unsigned)(N - 1);
        br[i] = make_float4(src[min(b_col[i], last)], src[min(b_col[i] + 1, last)], src[min(b_col[i] + 2, last)],
                            src[min(b_col[i] + 3, last)]);
      }
    }
  };
  // operand fragments of 8 MFMA steps (16 k) at a time, read from LDS one group AHEAD of the matrix work: the reads of
  // group s+1 are issued before the MFMAs of group s, so the dependent MFMA chain never waits for LDS
  constexpr int NS = BK / 16;
  float fa[2][8][MI], fb[2][8][NI];
  auto read_frag = [&](int s, int buf) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int x = 0; x < MI; ++x) fa[buf][kk][x] = As[(16 * s + 2 * kk + lh) * LDA + x * 32 + li];
#pragma unroll
      for (int y = 0; y < NI; ++y) fb[buf][kk][y] = Bs[(16 * s + 2 * kk + lh) * BN + (wave * NI + y) * 32 + li];
    }
  };
  load_chunk(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int e = tid + i * 256;
      const int m = ATRANS ? e % BM : e / BK, k = ATRANS ? e / BM : e % BK;
      As[k * LDA + m] = (k0 + k < K) ? ar[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int e = tid + i * 256, k = e / (BN / 4), c4 = (e % (BN / 4)) * 4;
      const bool ok = k0 + k < K;
      float4 v = br[i];
      if constexpr (FOLD) {
        float2 cf = coef_s[min(k0 + k, K - 1)];
        lds_settle(cf.x, cf.y);  // (common.h: the compiler's partial lgkmcnt wait is not enough next to an LDS-heavy neighbour)
        v = make_float4(swishf(cf.x * v.x + cf.y), swishf(cf.x * v.y + cf.y), swishf(cf.x * v.z + cf.y), swishf(cf.x * v.w + cf.y));
      }
      *reinterpret_cast<float4 *>(&Bs[k * BN + c4]) = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    __syncthreads();
    if (k0 + BK < K) load_chunk(k0 + BK);
    read_frag(0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) read_frag(s + 1, (s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the next group's LDS reads ABOVE this group's MFMAs
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int x = 0; x < MI; ++x)
#pragma unroll
          for (int y = 0; y < NI; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s & 1][kk][x], fb[s & 1][kk][y], acc[x][y], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: C/D map  row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31.  The per-row terms (bias, per-shape bias)
  // and the residual are fetched in batches under workgroup-uniform branches (clamped addresses), then applied.
  float bs[MI][4], bq[MI][4];  // GroupNorm partials of this lane's 4-row blocks j: rows x*32 + 8j + 4lh .. +3, ONE 32 x 32 sub-tile at a time
  float am[MI];                // max |y| of this lane's part of row block x
  // GroupNorm statistics are defined on a CANONICAL decomposition -- one slice per (32-row block, 128-column block), summed in the
  // order of the 1 x 1 tile: lane (4 rows), half-wave butterfly (32 columns), the block's four 32-column sub-tiles left to right,
  // then the 4-row blocks of a group in fp64 -- so the partials, and with them every normalised value downstream, do not depend
  // on the tile this launch happened to pick (the tile follows the batch size: a shape must not see its batch-mates).
  constexpr int NBW = MI * NI * 2 * 4 * 2;  // per wave: [x][y][lh][j][stat]
  float *red = Bs;                          // [4 waves][NBW], then [NI column blocks][MI][lh][j][stat]
  if (gn.out_partial != nullptr) __syncthreads();  // slower waves may still read operand fragments from Bs
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
    for (int r = 0; r < 16; ++r) {
      badd[r] = s_rowterm[0][x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      bb[r] = s_rowterm[1][x * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
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
        float v = (acc[x][y][r] + badd[r]) + bb[r];  // same order as y = (W x + b) + b_shape
        if (act == 2) v = v > 0.f ? v : v * slope;
        else if (act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // exact GELU (timm Mlp)
        if (R) v += rv[r];
        if (m < M && n < N) {
          Yb[(size_t)m * ldy + n] = v;
          bs[x][r >> 2] += v;
          bq[x][r >> 2] = __builtin_fmaf(v, v, bq[x][r >> 2]);  // explicitly fused: the same rounding in every tile variant
          am[x] = fmaxf(am[x], fabsf(v));
        }
      }
      if (gn.out_partial != nullptr) {  // this 32 x 32 sub-tile's sums: butterfly over its 32 columns, parked in LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bs[x][j] = half32_sum(bs[x][j]);   // DPP row sums + one readlane per row: bit-identical to the xor butterfly over the 32 columns,
          bq[x][j] = half32_sum(bq[x][j]);   // without its five dependent LDS-crossbar round trips per value (160 per wave in a 2 x 2 tile)
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
    // one atomic per (workgroup, row block) at most: waves combine through LDS, and a device-scope look at the slot first
    // skips the atomic once larger maxima have been published (thousands of same-address atomics cost this GEMM 60 us)
    __syncthreads();  // the operand tiles are dead: reuse As
#pragma unroll
    for (int x = 0; x < MI; ++x) {
      const float mx = wave_max(am[x]);
      if (lane == 0) As[x * 4 + wave] = mx;
    }
    __syncthreads();
    if (tid < MI && m0 + tid * 32 < M) {
      const float mx = fmaxf(fmaxf(As[tid * 4], As[tid * 4 + 1]), fmaxf(As[tid * 4 + 2], As[tid * 4 + 3]));
      unsigned *slot = gn.amax + (size_t)bi * gn.amax_slots + (m0 + tid * 32) / gn.amax_rows;
      if (mx > 0.f && __float_as_uint(mx) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(slot, __float_as_uint(mx));
    }
    __syncthreads();
  }
  if (gn.out_partial != nullptr) {
    __syncthreads();
    // the four sub-tiles of each canonical 128-column block, left to right: sub-tile g of the workgroup lives in wave g / NI, slot g % NI
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
    const int cg = gn.out_cg, G = M / cg;
    const int ngt = cg >= 32 ? 1 : 32 / cg;   // groups inside a 32-row block
    const int rt = cg >= 32 ? cg / 32 : 1;    // 32-row blocks per group
    if (tid < NI * MI * ngt) {
      const int gi = tid % ngt, x = (tid / ngt) % MI, cb = tid / (ngt * MI);
      const int row0 = m0 + x * 32, colblock = blockIdx.x * NI + cb, ncb = (N + 127) / 128;
      if (row0 + gi * cg < M && colblock < ncb) {
        double a = 0.0, qq = 0.0;
        for (int hh = 0; hh < 2; ++hh)
          for (int j = 0; j < 4; ++j)
            if (cg >= 32 || ((8 * j + 4 * hh) >> (__ffs(cg) - 1)) == gi) {  // cg is a power of two
              a += (double)red[4 * NBW + (((cb * MI + x) * 2 + hh) * 4 + j) * 2 + 0];
              qq += (double)red[4 * NBW + (((cb * MI + x) * 2 + hh) * 4 + j) * 2 + 1];
            }
        const int g = row0 / cg + gi, S = ncb * rt, sl = colblock * rt + (row0 / 32) % rt;
        double *dst = gn.out_partial + (((size_t)bi * G + g) * S + sl) * 2;
        dst[0] = a;
        dst[1] = qq;
      }
    }
  }
}

// ---- skinny GEMM: n <= 32 columns per shape (the 16-point level: attention projections 1536 x 512 . 16) ---------------------
// In pw_gemm_kernel such a problem keeps one wave of four busy and walks the whole K axis as a chain of barrier-separated chunks,
// each exposing a global-load latency (47 us for 0.4 GFLOP).  Here the workgroup owns a 32-row x 32-column tile and its four
// waves SPLIT K: each wave streams its quarter of W and x straight from global memory into MFMA operands (no LDS staging, no
// barriers in the loop, loads of several 8-deep blocks in flight), then the four partial tiles are added in wave order through
// LDS (deterministic) and the epilogue (bias, per-shape bias, activation, residual) is applied once.
// Operand map of v_mfma_f32_32x32x2: lane (li, lh) supplies A[row li][k] and B[k][column li] for ONE k per instruction; within an
// 8-deep block lane half lh takes k = kb + 4 lh + j at step j, so its four A values are one 16-byte load.
template <int NB, bool FOLD>  // NB column blocks of 32 (n <= 32 NB); FOLD: the operand is Swish(GroupNorm(x)), as in pw_gemm_kernel
__global__ __launch_bounds__(256) void pw_skinny_kernel(int M, int K, int N, const float *__restrict__ W, int ldw,
                                                        const float *__restrict__ X, long long bsx, int ldx,
                                                        const float *__restrict__ bias, const float *__restrict__ bbias, int ldbb,
                                                        const float *__restrict__ R, long long bsr, int ldr,
                                                        float *__restrict__ Y, long long bsy, int ldy, int act, float slope, PwGn gn) {
  __shared__ float red[4][NB * 16][64];
  __shared__ float2 coef_s[FOLD ? 1024 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 32, bi = blockIdx.y, n0 = blockIdx.z * (32 * NB);  // wider shapes: one workgroup per 32 NB columns
  if constexpr (FOLD) {  // same arithmetic as pw_gemm_kernel's prologue: the coefficients have the same bits
    __shared__ float s_mr[16];
    const int g = tid >> 5, l = tid & 31, cgi = K / gn.in_G;
    double a = 0.0, q = 0.0;
    if (g < gn.in_G) {
      const double *pp = gn.in_partial + ((size_t)bi * gn.in_G + g) * gn.in_S * 2;
      for (int sl = l; sl < gn.in_S; sl += 32) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    }
    a = half32_sum(a); q = half32_sum(q);   // (the 32 lanes of this group: DPP, bit-identical to the xor butterfly)
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
    __syncthreads();
  }
  const float *Xb = X + (size_t)bi * bsx;
  const float *X2b = gn.x2 ? gn.x2 + (size_t)bi * gn.bsx2 : nullptr;
  const int k1 = X2b ? gn.k1 : K;  // rows [k1, K) of the operand live in X2b (cat([x, x2], dim=1) read in place)
  auto xrow = [&](int k) { return k < k1 ? Xb + (unsigned)k * (unsigned)ldx : X2b + (unsigned)(k - k1) * (unsigned)gn.ldx2; };
  const int kq = ((K + 31) / 32) * 8;                  // K range of a wave: a multiple of 8
  const int k_lo = wave * kq, k_hi = min(K, k_lo + kq);
  const unsigned arow = (unsigned)min(m0 + li, M - 1) * (unsigned)ldw;
  unsigned bcol[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) bcol[c] = (unsigned)min(n0 + c * 32 + li, N - 1);
  const bool a_vec = ((ldw & 3) == 0) && ((((uintptr_t)W) & 15) == 0);
  f32x16 acc[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  // groups of four 8-deep blocks, two groups of operand registers: the loads of group g + 1 are issued before the MFMAs of
  // group g.  Blocks past the wave's range are skipped by wave-uniform branches (their A values stay zero).
  const bool fast = a_vec && (K & 7) == 0;
  float A0[4][4], B0[NB][4][4], A1[4][4], B1[NB][4][4];
  auto load_group = [&](int kg, float (&A)[4][4], float (&Bv)[NB][4][4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int kb = kg + 8 * g, k0 = kb + 4 * lh;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        A[g][j] = 0.f;
#pragma unroll
        for (int c = 0; c < NB; ++c) Bv[c][g][j] = 0.f;
      }
      if (kb < k_hi) {
        if (fast) {
          const float4 t = *reinterpret_cast<const float4 *>(W + arow + k0);
          A[g][0] = t.x; A[g][1] = t.y; A[g][2] = t.z; A[g][3] = t.w;
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < NB; ++c) Bv[c][g][j] = xrow(k0 + j)[bcol[c]];
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = min(k0 + j, K - 1);
            const float av = W[arow + k];
#pragma unroll
            for (int c = 0; c < NB; ++c) Bv[c][g][j] = xrow(k)[bcol[c]];
            A[g][j] = k0 + j < K ? av : 0.f;  // a zero A entry removes the clamped duplicate from the sum
          }
        }
      }
    }
  };
  auto mfma_group = [&](int kg, float (&A)[4][4], float (&Bv)[NB][4][4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float2 cf = make_float2(1.f, 0.f);
        if constexpr (FOLD) { cf = coef_s[min(kg + 8 * g + 4 * lh + j, K - 1)]; lds_settle(cf.x, cf.y); }  // (common.h)
#pragma unroll
        for (int c = 0; c < NB; ++c) {
          float bv = Bv[c][g][j];
          if constexpr (FOLD) bv = swishf(cf.x * bv + cf.y);
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[g][j], bv, acc[c], 0, 0, 0);
        }
      }
  };
  if (k_lo < k_hi) load_group(k_lo, A0, B0);
  for (int kg = k_lo; kg < k_hi; kg += 64) {
    if (kg + 32 < k_hi) load_group(kg + 32, A1, B1);
    mfma_group(kg, A0, B0);
    if (kg + 64 < k_hi) load_group(kg + 64, A0, B0);
    if (kg + 32 < k_hi) mfma_group(kg + 32, A1, B1);
  }
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][c * 16 + r][lane] = acc[c][r];
  __syncthreads();
  // NB * 1024 outputs, 256 threads: thread (wave, lane) finishes accumulator registers r = 4 wave .. 4 wave + 3 of lane `lane`,
  // i.e. the four consecutive rows 8 wave + 4 lh .. + 3 of column li of every column block
  float gs = 0.f, gq = 0.f, am = 0.f;
#pragma unroll
  for (int c = 0; c < NB; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = 4 * wave + q;
      const int m = m0 + q + 8 * wave + 4 * lh, n = n0 + c * 32 + li;
      if (m < M && n < N) {
        float v = ((red[0][c * 16 + r][lane] + red[1][c * 16 + r][lane]) + red[2][c * 16 + r][lane]) + red[3][c * 16 + r][lane];
        v = (v + (bias ? bias[m] : 0.f)) + (bbias ? bbias[(size_t)bi * ldbb + m] : 0.f);
        if (act == 2) v = v > 0.f ? v : v * slope;
        else if (act == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (R) v += R[(size_t)bi * bsr + (size_t)m * ldr + n];
        Y[(size_t)bi * bsy + (size_t)m * ldy + n] = v;
        gs += v;
        gq += v * v;
        am = fmaxf(am, fabsf(v));
      }
    }
  if (gn.amax != nullptr) {  // (amax_rows % 32 == 0: the tile's rows share a slot)
    const float mx = wave_max(am);
    if (lane == 0 && mx > 0.f) atomicMax(gn.amax + (size_t)bi * gn.amax_slots + m0 / gn.amax_rows, __float_as_uint(mx));
  }
  if (gn.out_partial != nullptr) {
    // GroupNorm statistics of the tile (layout of pw_gemm_kernel's partials: slices = column tiles x row tiles per group):
    // butterfly over the 32 columns, then the eight (wave, half) row blocks of 4 are added per group in fp64 in a fixed order
    gs = half32_sum(gs); gq = half32_sum(gq);   // (bit-identical to the xor butterfly)
    __syncthreads();  // red is dead
    float *blk = &red[0][0][0];  // [8 row blocks][2]
    if (li == 0) { blk[(2 * wave + lh) * 2] = gs; blk[(2 * wave + lh) * 2 + 1] = gq; }
    __syncthreads();
    const int cg = gn.out_cg, G = M / cg;
    const int ngt = cg >= 32 ? 1 : 32 / cg, rt = cg >= 32 ? cg / 32 : 1;
    if (tid < ngt && m0 + tid * cg < M) {
      double a = 0.0, qq = 0.0;
      for (int rb = 0; rb < 8; ++rb)
        if (cg >= 32 || ((4 * rb) >> (__ffs(cg) - 1)) == tid) { a += (double)blk[rb * 2]; qq += (double)blk[rb * 2 + 1]; }
      const int g = m0 / cg + tid, S = gridDim.z * rt, sl = blockIdx.z * rt + (int)(blockIdx.x % rt);
      double *dst = gn.out_partial + (((size_t)bi * G + g) * S + sl) * 2;
      dst[0] = a;
      dst[1] = qq;
    }
  }
}

// shapes of the GroupNorm-folded entry point that take the skinny kernel (the caller also needs x2 == NULL and amax == NULL; the
// slice count of the statistics must not depend on those, so two-source / amax calls of such shapes use the general kernel's
// layout only when this returns 0 -- see bdm_pointwise_conv_gn)
static int pw_skinny_max_n() { return 64; }  // widest shape (columns) that takes the skinny kernel (wider measured slower)
static bool pw_skinny_shape(int k, int n) { return n <= pw_skinny_max_n() && k >= 128; }

// workgroup count below which the long-K shapes take the 64-deep K chunk
static int pw_deep_limit() { return 1024; }

// biggest tile that still gives the 256 CUs two workgroups each; small problems (the 16..256-point levels) are latency-bound
// and prefer many small tiles over operand reuse, and take the 64-deep K chunk when K is long
// K chunk of the 256-column tiles: 32 halves the barrier pairs of the short-K layers but measured neutral to slightly slower on
// the forward (7.42 vs 7.46 ms), so 16 it is
static int pw_wide_bk() { return 16; }

static void pw_tile(int b, int m, int k, int n, int *mi, int *ni, int *bk) {
  auto blocks = [&](int a, int c) { return (long long)cdiv(n, 128 * c) * cdiv(m, 32 * a) * b; };
  const int deep_limit = pw_deep_limit();
  *bk = 16;
  if (m <= 32) {
    if (n <= 128 || blocks(1, 2) < 512) { *mi = 1; *ni = 1; if (k >= 128 && blocks(1, 1) < deep_limit) *bk = 64; }
    else { *mi = 1; *ni = 2; *bk = pw_wide_bk(); }
  } else if (n > 128 && blocks(2, 2) >= 512) {
    *mi = 2; *ni = 2; *bk = pw_wide_bk();
  } else if (blocks(2, 1) >= 512) {
    *mi = 2; *ni = 1; if (k >= 128 && blocks(2, 1) < deep_limit) *bk = 64;
  } else {
    *mi = 1; *ni = 1; if (k >= 128 && blocks(1, 1) < deep_limit) *bk = 64;
  }
}

void bdm_pw_tile(int b, int m, int k, int n, int *mi, int *ni, int *bk) { pw_tile(b, m, k, n, mi, ni, bk); }  // for pointwise_s3.hip

// tile choice + launch, shared by the plain and the GroupNorm-folded entry points.  *bm_out = rows of the chosen tile,
// *gx_out = column tiles (the GroupNorm slice count derives from both)
template <bool FOLD>
static void pw_dispatch(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x, int ld_x,
                        const float *bias, const float *batch_bias, int ld_bb, const float *residual, long long bs_r, int ld_r,
                        float *y, long long bs_y, int ld_y, int act, float slope, PwGn gn, hipStream_t s) {
#define PW_LAUNCH(MI, NI, BK)                                                                                             \
  hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, false, BK, FOLD>), dim3(cdiv(n, 128 * NI), cdiv(m, 32 * MI), b), dim3(256), 0, s, \
                     m, k, n, w, ldw, 0ll, (const int *)nullptr, x, bs_x, ld_x, bias, batch_bias, ld_bb, residual, bs_r, ld_r,   \
                     y, bs_y, ld_y, act, slope, gn)
  int mi, ni, bk;
  pw_tile(b, m, k, n, &mi, &ni, &bk);
  if (mi == 1 && ni == 1 && bk == 16) PW_LAUNCH(1, 1, 16);
  else if (mi == 1 && ni == 1) PW_LAUNCH(1, 1, 64);
  else if (mi == 1 && ni == 2 && bk == 16) PW_LAUNCH(1, 2, 16);
  else if (mi == 1 && ni == 2) PW_LAUNCH(1, 2, 32);
  else if (mi == 2 && ni == 2 && bk == 16) PW_LAUNCH(2, 2, 16);
  else if (mi == 2 && ni == 2) PW_LAUNCH(2, 2, 32);
  else if (bk == 16) PW_LAUNCH(2, 1, 16);
  else PW_LAUNCH(2, 1, 64);
#undef PW_LAUNCH
}

extern "C" int bdm_pointwise_conv(int b, int m, int k, int n, const float *w, int ldw, const float *x,
                                  long long bs_x, int ld_x, const float *bias, const float *batch_bias,
                                  int ld_bb, const float *residual, long long bs_r, int ld_r, float *y,
                                  long long bs_y, int ld_y, int act, float slope, void *stream) {
  BDM_REQUIRE(b >= 0 && m >= 1 && k >= 1 && n >= 0, "pointwise_conv: bad sizes m=%d k=%d n=%d", m, k, n);
  BDM_REQUIRE(act == 0 || act == 2 || act == 3, "pointwise_conv: act must be 0 (none), 2 (leaky relu) or 3 (gelu)");
  BDM_REQUIRE((long long)k * ld_x + n < (1ll << 31) && (long long)m * ldw + k < (1ll << 31),
              "pointwise_conv: one operand spans more than 2^31 elements");
  if (b == 0 || n == 0) return BDM_OK;
  if (pw_skinny_shape(k, n)) {
    if (n <= 32)
      hipLaunchKernelGGL((pw_skinny_kernel<1, false>), dim3(cdiv(m, 32), b, 1), dim3(256), 0, (hipStream_t)stream, m, k, n, w, ldw, x, bs_x,
                         ld_x, bias, batch_bias, ld_bb, residual, bs_r, ld_r, y, bs_y, ld_y, act, slope, PwGn{});
    else
      hipLaunchKernelGGL((pw_skinny_kernel<2, false>), dim3(cdiv(m, 32), b, cdiv(n, 64)), dim3(256), 0, (hipStream_t)stream, m, k, n, w, ldw, x, bs_x,
                         ld_x, bias, batch_bias, ld_bb, residual, bs_r, ld_r, y, bs_y, ld_y, act, slope, PwGn{});
    return launch_status("pointwise_conv");
  }
  pw_dispatch<false>(b, m, k, n, w, ldw, x, bs_x, ld_x, bias, batch_bias, ld_bb, residual, bs_r, ld_r, y, bs_y, ld_y, act, slope,
                     PwGn{}, (hipStream_t)stream);
  return launch_status("pointwise_conv");
}

// slices per (shape, group) the GroupNorm-folded convolution below writes for an (m x n) output in `groups` groups
extern "C" int bdm_pointwise_conv_gn_slices(int b, int m, int k, int n, int groups) {
  if (groups < 1 || m % groups) return 0;
  if (pw_skinny_shape(k, n)) return (n <= 32 ? 1 : cdiv(n, 64)) * ((m / groups) >= 32 ? (m / groups) / 32 : 1);  // 32-row tiles
  const int cg = m / groups;  // canonical decomposition (independent of the tile and of b): 128-column blocks x 32-row blocks
  return cdiv(n, 128) * (cg >= 32 ? cg / 32 : 1);
}

static int pointwise_conv_gn_impl(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x,
                                 int ld_x, const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y,
                                 long long bs_y, int ld_y, const void *in_partial, int in_slices, int in_groups,
                                 const float *in_gamma, const float *in_beta, float in_eps, int out_groups,
                                 void *out_partial, float *amax, int amax_rows, const float *add, long long bs_add, int ld_add,
                                 void *stream, const float *batch_bias = nullptr, int ld_bb = 0) {
  BDM_REQUIRE(b >= 0 && m >= 1 && k >= 1 && n >= 1, "pointwise_conv_gn: bad sizes m=%d k=%d n=%d", m, k, n);
  BDM_REQUIRE((long long)k * ld_x + n < (1ll << 31) && (long long)m * ldw + k < (1ll << 31),
              "pointwise_conv_gn: one operand spans more than 2^31 elements");
  PwGn gn{};
  if (x2 != nullptr) {
    BDM_REQUIRE(k1 >= 1 && k1 < k && (long long)(k - k1) * ld_x2 + n < (1ll << 31) && in_partial == nullptr,
                "pointwise_conv_gn: second source needs 1 <= k1 < k and no input fold (k=%d k1=%d)", k, k1);
    gn.x2 = x2; gn.bsx2 = bs_x2; gn.ldx2 = ld_x2; gn.k1 = k1;
  }
  if (in_partial != nullptr) {
    BDM_REQUIRE(in_groups >= 1 && in_groups <= 8 && k % in_groups == 0 && k <= 1024 && in_slices >= 1 && in_gamma && in_beta,
                "pointwise_conv_gn: input fold needs <= 8 groups dividing k <= 1024 (k=%d groups=%d)", k, in_groups);
    gn.in_partial = (const double *)in_partial;
    gn.in_S = in_slices; gn.in_G = in_groups; gn.in_gamma = in_gamma; gn.in_beta = in_beta; gn.in_eps = in_eps;
  }
  if (amax != nullptr) {
    BDM_REQUIRE(amax_rows >= 32 && amax_rows % 32 == 0, "pointwise_conv_gn: amax_rows must be a multiple of 32 (got %d)", amax_rows);
    gn.amax = (unsigned *)amax; gn.amax_rows = amax_rows; gn.amax_slots = (m + amax_rows - 1) / amax_rows;
  }
  if (out_partial != nullptr) {
    const int cg = out_groups >= 1 && m % out_groups == 0 ? m / out_groups : 0;
    BDM_REQUIRE(cg >= 4 && (cg & (cg - 1)) == 0, "pointwise_conv_gn: output statistics need a power-of-two >= 4 channels per group "
                "(m=%d groups=%d)", m, out_groups);
    gn.out_partial = (double *)out_partial; gn.out_cg = cg;
  }
  if (b == 0) return BDM_OK;
  BDM_REQUIRE(add == nullptr || !pw_skinny_shape(k, n), "pointwise_conv_gn_add: not for the skinny shapes (n <= 64, k >= 128)");
  if (pw_skinny_shape(k, n)) {
#define SK_LAUNCH(NB, FOLD)                                                                                                       \
    hipLaunchKernelGGL((pw_skinny_kernel<NB, FOLD>), dim3(cdiv(m, 32), b, cdiv(n, 32 * NB)), dim3(256), 0, (hipStream_t)stream, m, k, n, w, ldw, x, bs_x, \
                       ld_x, bias, batch_bias, ld_bb, (const float *)nullptr, 0ll, 0, y, bs_y, ld_y, 0, 0.f, gn)
    if (in_partial != nullptr) { if (n <= 32) SK_LAUNCH(1, true); else SK_LAUNCH(2, true); }
    else { if (n <= 32) SK_LAUNCH(1, false); else SK_LAUNCH(2, false); }
#undef SK_LAUNCH
    return launch_status("pointwise_conv_gn");
  }
  if (in_partial != nullptr)
    pw_dispatch<true>(b, m, k, n, w, ldw, x, bs_x, ld_x, bias, batch_bias, ld_bb, add, bs_add, ld_add, y, bs_y, ld_y, 0, 0.f, gn, (hipStream_t)stream);
  else
    pw_dispatch<false>(b, m, k, n, w, ldw, x, bs_x, ld_x, bias, batch_bias, ld_bb, add, bs_add, ld_add, y, bs_y, ld_y, 0, 0.f, gn, (hipStream_t)stream);
  return launch_status("pointwise_conv_gn");
}

// Sparse first-convolution GEMM (sparse_conv.hip): Y[b] (n_max x n27) = Xc[b]^T (n_max x cin) . Wt (cin x n27), rows
// >= n_occ[b] skipped.  Xc is channel-first (b, cin, n_max): the A operand arrives transposed.
extern "C" int bdm_pointwise_conv_gn(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x,
                                     int ld_x, const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y,
                                     long long bs_y, int ld_y, const void *in_partial, int in_slices, int in_groups,
                                     const float *in_gamma, const float *in_beta, float in_eps, int out_groups,
                                     void *out_partial, float *amax, int amax_rows, void *stream) {
  return pointwise_conv_gn_impl(b, m, k, n, w, ldw, x, bs_x, ld_x, x2, bs_x2, ld_x2, k1, bias, y, bs_y, ld_y, in_partial, in_slices,
                                in_groups, in_gamma, in_beta, in_eps, out_groups, out_partial, amax, amax_rows, nullptr, 0ll, 0, stream);
}

// The same with a per-element addend: y = W x' + bias + add, statistics / amax over y INCLUDING the addend (the hoisted share of a
// layer whose remaining input columns were applied to the conditioning image once per trajectory: sparse_conv.hip 2'').
extern "C" int bdm_pointwise_conv_gn_add(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x,
                                         int ld_x, const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y,
                                         long long bs_y, int ld_y, const void *in_partial, int in_slices, int in_groups,
                                         const float *in_gamma, const float *in_beta, float in_eps, int out_groups,
                                         void *out_partial, float *amax, int amax_rows, const float *add, long long bs_add,
                                         int ld_add, void *stream) {
  BDM_REQUIRE(add != nullptr, "pointwise_conv_gn_add: add is NULL");
  return pointwise_conv_gn_impl(b, m, k, n, w, ldw, x, bs_x, ld_x, x2, bs_x2, ld_x2, k1, bias, y, bs_y, ld_y, in_partial, in_slices,
                                in_groups, in_gamma, in_beta, in_eps, out_groups, out_partial, amax, amax_rows, add, bs_add, ld_add, stream);
}

// The same with a per-SHAPE bias batch_bias (b, ld_bb >= m): y = (W x' + bias) + batch_bias[shape] (+ add), statistics / amax over that y.
// The share of a layer whose remaining input columns are constant along the points of a shape (the time embedding concatenated to the
// features: W . [x ; t 1^T] = W_x . x + (W_t . t) 1^T, pvcnn.py:88 / pointnet.py:104-112).  add may be NULL.
extern "C" int bdm_pointwise_conv_gn_bb(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x,
                                        int ld_x, const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y,
                                        long long bs_y, int ld_y, const void *in_partial, int in_slices, int in_groups,
                                        const float *in_gamma, const float *in_beta, float in_eps, int out_groups,
                                        void *out_partial, float *amax, int amax_rows, const float *batch_bias, int ld_bb,
                                        const float *add, long long bs_add, int ld_add, void *stream) {
  BDM_REQUIRE(batch_bias != nullptr && ld_bb >= m, "pointwise_conv_gn_bb: batch_bias is NULL or its row stride is below m");
  return pointwise_conv_gn_impl(b, m, k, n, w, ldw, x, bs_x, ld_x, x2, bs_x2, ld_x2, k1, bias, y, bs_y, ld_y, in_partial, in_slices,
                                in_groups, in_gamma, in_beta, in_eps, out_groups, out_partial, amax, amax_rows, add, bs_add, ld_add, stream,
                                batch_bias, ld_bb);
}

extern "C" int bdm_sparse_conv_gemm(int b, int n_max, int cin, int n27, const float *xc, const float *wt,
                                    const int *n_occ, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && n_max >= 1 && cin >= 1 && n27 >= 1, "sparse_conv_gemm: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL((pw_gemm_kernel<2, 2, true>), dim3(cdiv(n27, 256), cdiv(n_max, 64), b), dim3(256), 0,
                     (hipStream_t)stream, n_max, cin, n27, xc, n_max, (long long)cin * n_max, n_occ, wt, 0ll, n27,
                     (const float *)nullptr, (const float *)nullptr, 0, (const float *)nullptr, 0ll, 0, y,
                     (long long)n_max * n27, n27, 0, 0.f, PwGn{});
  return launch_status("sparse_conv_gemm");
}

// =====================================================================================
// GroupNorm(G) [+ residual] [+ Swish] over (B, C, L) with row stride ld
// =====================================================================================
// Two launches: slice-partial (sum, sumsq) in fp64, then normalise; the second kernel re-reduces
// the S partials in a fixed order, so results are run-to-run deterministic.
#define GN_MAX_SLICES 64

__global__ void gn_stats_kernel(int cg, int L, const float *__restrict__ x, long long bs, int ld,
                                const float *__restrict__ res, long long bs_r, int ld_r, int G,
                                double *__restrict__ partial) {
  const int S = gridDim.x, s = blockIdx.x, bg = blockIdx.y, bi = bg / G, g = bg % G;
  const float *xb = x + (size_t)bi * bs + (size_t)g * cg * ld;
  const float *rb = res ? res + (size_t)bi * bs_r + (size_t)g * cg * ld_r : nullptr;
  const long long total = (long long)cg * L;
  const long long per = (total + S - 1) / S;
  const long long lo = (long long)s * per, hi = lo + per < total ? lo + per : total;
  double sum = 0.0, sq = 0.0;
  for (long long e = lo + threadIdx.x; e < hi; e += blockDim.x) {
    const int row = (int)(e / L), col = (int)(e % L);
    float v = xb[(size_t)row * ld + col];
    if (rb) v += rb[(size_t)row * ld_r + col];
    sum += v;
    sq += (double)v * v;
  }
  __shared__ double sh[2][16];
  sum = wave_sum(sum);
  sq = wave_sum(sq);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[0][wave] = sum; sh[1][wave] = sq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, q = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += sh[0][w]; q += sh[1][w]; }
    partial[((size_t)bg * S + s) * 2 + 0] = a;
    partial[((size_t)bg * S + s) * 2 + 1] = q;
  }
}

__global__ void gn_apply_kernel(int cg, int L, const float *__restrict__ x, long long bs, int ld,
                                const float *__restrict__ res, long long bs_r, int ld_r, int G, int S,
                                const double *__restrict__ partial, const float *__restrict__ gamma,
                                const float *__restrict__ beta, float eps, int act, float *__restrict__ y,
                                long long bs_y, int ld_y) {
  const int bg = blockIdx.y, bi = bg / G, g = bg % G;
  double a = 0.0, q = 0.0;
  for (int s = 0; s < S; ++s) { a += partial[((size_t)bg * S + s) * 2]; q += partial[((size_t)bg * S + s) * 2 + 1]; }
  const double cnt = (double)cg * L;
  const double mean_d = a / cnt;
  double var = q / cnt - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float *xb = x + (size_t)bi * bs + (size_t)g * cg * ld;
  const float *rb = res ? res + (size_t)bi * bs_r + (size_t)g * cg * ld_r : nullptr;
  float *yb = y + (size_t)bi * bs_y + (size_t)g * cg * ld_y;
  const long long total = (long long)cg * L;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(e / L), col = (int)(e % L);
    float v = xb[(size_t)row * ld + col];
    if (rb) v += rb[(size_t)row * ld_r + col];
    const int ch = g * cg + row;
    v = (v - mean) * rstd * gamma[ch] + beta[ch];
    if (act == 1) v = swishf(v);
    yb[(size_t)row * ld_y + col] = v;
  }
}

// ---- fast paths for densely packed chunks (row stride == L, L % 4 == 0, 16-B aligned): one (shape, group) chunk is a
// contiguous run of cg*L floats.
//   * gn_onepass_kernel: chunk <= 64 Ki floats -> ONE workgroup keeps it in registers: single HBM read, mean then
//     centred variance (two reductions on registers), normalise, Swish, store.  One launch instead of two.
//   * gn_stats_vec / gn_apply_vec: float4 versions of the two-pass kernels for the big voxel / grouped tensors.
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double a = 0.0;
  for (int w = 0; w < nw; ++w) a += sh[w];
  return a;
}

#define GN1_MAXV 16  // float4 per thread
__global__ __launch_bounds__(1024) void gn_onepass_kernel(int cg, int L, const float *__restrict__ x, long long bs,
                                                          const float *__restrict__ res, long long bs_r, int G,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          float eps, int act, float *__restrict__ y, long long bs_y) {
  __shared__ double sh[16];
  const int bg = blockIdx.x, bi = bg / G, g = bg % G;
  const int n4 = cg * L / 4;
  const float4 *xb = reinterpret_cast<const float4 *>(x + (size_t)bi * bs + (size_t)g * cg * L);
  const float4 *rb = res ? reinterpret_cast<const float4 *>(res + (size_t)bi * bs_r + (size_t)g * cg * L) : nullptr;
  float4 *yb = reinterpret_cast<float4 *>(y + (size_t)bi * bs_y + (size_t)g * cg * L);
  float4 v[GN1_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < GN1_MAXV; ++i) {
    const int e = threadIdx.x + i * 1024;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < n4) {
      v[i] = xb[e];
      if (rb) { const float4 r = rb[e]; v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w; }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const double cnt = (double)cg * L;
  const float mean = (float)(block_sum((double)s, sh) / cnt);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < GN1_MAXV; ++i) {
    const int e = threadIdx.x + i * 1024;
    if (e < n4) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = (float)(1.0 / sqrt(block_sum((double)q, sh) / cnt + (double)eps));
  const int L4 = L / 4;
  const float inv_L4 = 1.0f / (float)L4;  // e / L4 as a float product: exact for e < 2^20 (an integer division is ~40 instructions)
#pragma unroll
  for (int i = 0; i < GN1_MAXV; ++i) {
    const int e = threadIdx.x + i * 1024;
    if (e < n4) {
      const int ch = g * cg + (int)(((float)e + 0.5f) * inv_L4);
      const float ga = gamma[ch] * rstd, be = beta[ch] - mean * ga;
      float4 o;
      o.x = v[i].x * ga + be; o.y = v[i].y * ga + be; o.z = v[i].z * ga + be; o.w = v[i].w * ga + be;
      if (act == 1) { o.x = swishf(o.x); o.y = swishf(o.y); o.z = swishf(o.z); o.w = swishf(o.w); }
      yb[e] = o;
    }
  }
}

__global__ void gn_stats_vec_kernel(int cg, int L, const float *__restrict__ x, long long bs,
                                    const float *__restrict__ res, long long bs_r, int G, double *__restrict__ partial) {
  __shared__ double sh[16];
  const int S = gridDim.x, s = blockIdx.x, bg = blockIdx.y, bi = bg / G, g = bg % G;
  const int n4 = cg * L / 4;
  const float4 *xb = reinterpret_cast<const float4 *>(x + (size_t)bi * bs + (size_t)g * cg * L);
  const float4 *rb = res ? reinterpret_cast<const float4 *>(res + (size_t)bi * bs_r + (size_t)g * cg * L) : nullptr;
  const int per = (n4 + S - 1) / S, lo = s * per, hi = min(lo + per, n4);
  float sum = 0.f, sq = 0.f;  // <= 2048 addends per thread; combined in fp64 below
  for (int e = lo + threadIdx.x; e < hi; e += blockDim.x) {
    float4 v = xb[e];
    if (rb) { const float4 r = rb[e]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    sum += (v.x + v.y) + (v.z + v.w);
    sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const double a = block_sum((double)sum, sh), q = block_sum((double)sq, sh);
  if (threadIdx.x == 0) {
    partial[((size_t)bg * S + s) * 2 + 0] = a;
    partial[((size_t)bg * S + s) * 2 + 1] = q;
  }
}

__global__ void gn_apply_vec_kernel(int cg, int L, const float *__restrict__ x, long long bs,
                                    const float *__restrict__ res, long long bs_r, int G, int S,
                                    const double *__restrict__ partial, const float *__restrict__ gamma,
                                    const float *__restrict__ beta, float eps, int act, float *__restrict__ y,
                                    long long bs_y) {
  const int bg = blockIdx.y, bi = bg / G, g = bg % G;
  double a = 0.0, q = 0.0;
  for (int s = 0; s < S; ++s) { a += partial[((size_t)bg * S + s) * 2]; q += partial[((size_t)bg * S + s) * 2 + 1]; }
  const double cnt = (double)cg * L;
  const double mean_d = a / cnt;
  double var = q / cnt - mean_d * mean_d;
  if (var < 0) var = 0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  const int n4 = cg * L / 4, L4 = L / 4;
  const float4 *xb = reinterpret_cast<const float4 *>(x + (size_t)bi * bs + (size_t)g * cg * L);
  const float4 *rb = res ? reinterpret_cast<const float4 *>(res + (size_t)bi * bs_r + (size_t)g * cg * L) : nullptr;
  float4 *yb = reinterpret_cast<float4 *>(y + (size_t)bi * bs_y + (size_t)g * cg * L);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += gridDim.x * blockDim.x) {
    float4 v = xb[e];
    if (rb) { const float4 r = rb[e]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    const int ch = g * cg + e / L4;
    const float ga = gamma[ch], be = beta[ch];
    float4 o;
    o.x = (v.x - mean) * rstd * ga + be; o.y = (v.y - mean) * rstd * ga + be;
    o.z = (v.z - mean) * rstd * ga + be; o.w = (v.w - mean) * rstd * ga + be;
    if (act == 1) { o.x = swishf(o.x); o.y = swishf(o.y); o.z = swishf(o.z); o.w = swishf(o.w); }
    yb[e] = o;
  }
}

// (sum, sumsq) slice partials of a contiguous (b, c, l) tensor; *slices_out = number of slices per (shape, group)
extern "C" int bdm_group_norm_stats(int b, int c, int l, int groups, const float *x, long long bs_x, void *workspace,
                                    int *slices_out, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && l >= 1 && groups >= 1 && c % groups == 0 && workspace != nullptr, "group_norm_stats: bad arguments");
  const int cg = c / groups;
  const long long total = (long long)cg * l;
  int S = (int)((total + 16383) / 16384);
  if (S < 1) S = 1;
  if (S > GN_MAX_SLICES) S = GN_MAX_SLICES;
  *slices_out = S;
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  const bool packed = (l % 4 == 0) && ((((uintptr_t)x) & 15) == 0) && (bs_x % 4 == 0) && total < (1ll << 30);
  if (packed)
    hipLaunchKernelGGL(gn_stats_vec_kernel, dim3(S, b * groups), dim3(256), 0, s, cg, l, x, bs_x, (const float *)nullptr,
                       0ll, groups, (double *)workspace);
  else
    hipLaunchKernelGGL(gn_stats_kernel, dim3(S, b * groups), dim3(256), 0, s, cg, l, x, bs_x, l, (const float *)nullptr,
                       0ll, 0, groups, (double *)workspace);
  return launch_status("group_norm_stats");
}

extern "C" size_t bdm_group_norm_workspace_bytes(int b, int groups) {
  return sizeof(double) * 2 * (size_t)b * groups * GN_MAX_SLICES;
}

extern "C" int bdm_group_norm(int b, int c, int l, int groups, const float *x, long long bs_x, int ld_x,
                              const float *residual, long long bs_r, int ld_r, const float *gamma,
                              const float *beta, float eps, int act, float *y, long long bs_y, int ld_y,
                              void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && l >= 1 && groups >= 1 && c % groups == 0, "group_norm: bad sizes c=%d groups=%d", c, groups);
  BDM_REQUIRE(workspace != nullptr, "group_norm: workspace is NULL");
  BDM_REQUIRE(act == 0 || act == 1, "group_norm: act must be 0 (none) or 1 (swish)");
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int cg = c / groups;
  const long long total = (long long)cg * l;
  double *partial = (double *)workspace;
  auto al16 = [](const void *p) { return (((uintptr_t)p) & 15) == 0; };
  const bool packed = ld_x == l && ld_y == l && (!residual || ld_r == l) && (l % 4 == 0) && al16(x) && al16(y) &&
                      (!residual || al16(residual)) && (bs_x % 4 == 0) && (bs_y % 4 == 0) && (!residual || bs_r % 4 == 0) &&
                      total < (1ll << 30);
  if (packed && total <= 4ll * 1024 * GN1_MAXV) {
    hipLaunchKernelGGL(gn_onepass_kernel, dim3(b * groups), dim3(1024), 0, s, cg, l, x, bs_x, residual, bs_r, groups, gamma,
                       beta, eps, act, y, bs_y);
    return launch_status("gn_onepass");
  }
  int S = (int)((total + 16383) / 16384);
  if (S < 1) S = 1;
  if (S > GN_MAX_SLICES) S = GN_MAX_SLICES;
  if (packed) {
    hipLaunchKernelGGL(gn_stats_vec_kernel, dim3(S, b * groups), dim3(256), 0, s, cg, l, x, bs_x, residual, bs_r, groups,
                       partial);
    int gx = (int)((total / 4 + 1023) / 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(gn_apply_vec_kernel, dim3(gx, b * groups), dim3(256), 0, s, cg, l, x, bs_x, residual, bs_r, groups, S,
                       partial, gamma, beta, eps, act, y, bs_y);
    return launch_status("gn_vec");
  }
  hipLaunchKernelGGL(gn_stats_kernel, dim3(S, b * groups), dim3(256), 0, s, cg, l, x, bs_x, ld_x, residual, bs_r,
                     ld_r, groups, partial);
  int rc = launch_status("gn_stats");
  if (rc) return rc;
  int gx = (int)((total + 1023) / 1024);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, b * groups), dim3(256), 0, s, cg, l, x, bs_x, ld_x, residual, bs_r,
                     ld_r, groups, S, partial, gamma, beta, eps, act, y, bs_y, ld_y);
  return launch_status("gn_apply");
}

// =====================================================================================
// max over the neighbour axis: (B, C, M, U) -> (B, C, M)
// =====================================================================================
__global__ void max_u_kernel(int c, int m, int u, const float *__restrict__ x, float *__restrict__ y,
                             long long bs_y, int ld_y) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= m) return;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) {
    const float *row = x + (((size_t)bi * c + ci) * m + j) * u;
    float v = row[0];
    if ((u & 3) == 0) {
      const float4 *r4 = reinterpret_cast<const float4 *>(row);
      for (int q = 0; q < u / 4; ++q) {
        const float4 t = r4[q];
        v = fmaxf(v, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
      }
    } else {
      for (int q = 1; q < u; ++q) v = fmaxf(v, row[q]);
    }
    y[(size_t)bi * bs_y + (size_t)ci * ld_y + j] = v;
  }
}

extern "C" int bdm_max_over_neighbors(int b, int c, int m, int u, const float *x, float *y, long long bs_y,
                                      int ld_y, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && m >= 1 && u >= 1, "max_over_neighbors: bad sizes");
  if (b == 0) return BDM_OK;
  dim3 grid(cdiv(m, 64), c < 256 ? c : 256, b);
  hipLaunchKernelGGL(max_u_kernel, grid, dim3(64), 0, (hipStream_t)stream, c, m, u, x, y, bs_y, ld_y);
  return launch_status("max_over_neighbors");
}

// The same reduction over Swish(GroupNorm(x)) with x the RAW output of the SA MLP's last convolution and the GroupNorm
// statistics taken from that convolution's slice partials (bdm_pointwise_conv_gn): the normalised (B, C, M, U) tensor is
// never written.  Swish is not monotonic, so every element is transformed before the maximum.
__global__ void max_u_gn_kernel(int c, int m, int u, int lpr, int G, int S, const float *__restrict__ x,
                                const double *__restrict__ partial, const float *__restrict__ gamma,
                                const float *__restrict__ beta, float eps, float *__restrict__ y, long long bs_y, int ld_y) {
  // u / 4 lanes share a row (one 16-byte piece each: the wave reads whole 128-byte lines), then a butterfly maximum; rows that
  // are not a multiple of 4 long (or longer than 256) take one lane per row (lpr = lanes per row, a power of two)
  const int lane = threadIdx.x;
  const int rpw = 64 / lpr, sub = lane % lpr;  // a workgroup (one wave) covers 64 rows in lpr passes of rpw rows
  const int bi = blockIdx.z, cg = c / G;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) {
    const int g = ci / cg;
    const double *pp = partial + ((size_t)bi * G + g) * S * 2;
    double a = 0.0, q = 0.0;
    for (int sl = lane; sl < S; sl += 64) { a += pp[2 * sl]; q += pp[2 * sl + 1]; }
    a = wave_sum_bfly(a); q = wave_sum_bfly(q);   // (DPP + readlanes, bit-identical to the 64-lane xor butterfly)
    const double cnt = (double)cg * m * u, mean = a / cnt;
    double var = q / cnt - mean * mean;
    if (var < 0) var = 0;
    const float ca = gamma[ci] * (float)(1.0 / sqrt(var + (double)eps)), cb = beta[ci] - (float)mean * ca;
    const float *xb = x + ((size_t)bi * c + ci) * m * u;
    if (lpr > 1) {
      for (int it = 0; it < lpr; ++it) {
        const int j = blockIdx.x * 64 + it * rpw + lane / lpr;
        const float4 t = reinterpret_cast<const float4 *>(xb + (size_t)min(j, m - 1) * u)[sub];
        float v = fmaxf(fmaxf(swishf(ca * t.x + cb), swishf(ca * t.y + cb)), fmaxf(swishf(ca * t.z + cb), swishf(ca * t.w + cb)));
        for (int o = 1; o < lpr; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        if (j < m && sub == 0) y[(size_t)bi * bs_y + (size_t)ci * ld_y + j] = v;
      }
    } else {
      const int j = blockIdx.x * 64 + lane;
      const float *row = xb + (size_t)min(j, m - 1) * u;
      float v = swishf(ca * row[0] + cb);
      for (int qd = 1; qd < u; ++qd) v = fmaxf(v, swishf(ca * row[qd] + cb));
      if (j < m) y[(size_t)bi * bs_y + (size_t)ci * ld_y + j] = v;
    }
  }
}

extern "C" int bdm_max_over_neighbors_gn(int b, int c, int m, int u, const float *x, const void *in_partial, int in_slices,
                                         int groups, const float *gamma, const float *beta, float eps, float *y, long long bs_y,
                                         int ld_y, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && m >= 1 && u >= 1, "max_over_neighbors_gn: bad sizes");
  BDM_REQUIRE(in_partial != nullptr && in_slices >= 1 && groups >= 1 && c % groups == 0 && gamma && beta,
              "max_over_neighbors_gn: bad GroupNorm arguments");
  if (b == 0) return BDM_OK;
  const int lpr = ((u & 3) == 0 && u <= 256 && ((u / 4) & (u / 4 - 1)) == 0 && ((reinterpret_cast<size_t>(x) & 15) == 0)) ? u / 4 : 1;
  BDM_REQUIRE(lpr == 1 || (u & 3) == 0, "max_over_neighbors_gn: internal");
  dim3 grid(cdiv(m, 64), c < 256 ? c : 256, b);
  hipLaunchKernelGGL(max_u_gn_kernel, grid, dim3(64), 0, (hipStream_t)stream, c, m, u, lpr, groups, in_slices, x,
                     (const double *)in_partial, gamma, beta, eps, y, bs_y, ld_y);
  return launch_status("max_over_neighbors_gn");
}

// =====================================================================================
// Grouped SA input: out[b] = cat[ group(coords) - centre , group(features) ]   (ball_query.py:16-30)
// =====================================================================================
__global__ void sa_group_kernel(int c, int n, int m, int u, const float *__restrict__ coords,
                                const float *__restrict__ centers, const float *__restrict__ feat, long long bs_f,
                                int ld_f, const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z, mu = m * u;
  if (e >= mu) return;
  const int src = idx[(size_t)bi * mu + e];
  float *ob = out + (size_t)bi * (c + 3) * mu;
  if (blockIdx.y == 0) {
    const int j = e / u;
    const float *pc = coords + (size_t)bi * 3 * n;
    const float *cc = centers + (size_t)bi * 3 * m;
#pragma unroll
    for (int d = 0; d < 3; ++d) ob[(size_t)d * mu + e] = pc[(size_t)d * n + src] - cc[(size_t)d * m + j];
  }
  const float *fb = feat + (size_t)bi * bs_f;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) ob[(size_t)(3 + ci) * mu + e] = fb[(size_t)ci * ld_f + src];
}

// --- point-major path ---------------------------------------------------------------------------
// The direct kernel above issues one scattered 4-byte load per (element, channel): a wave's 64 lanes hit 64
// different lines, so the texture-address unit, not HBM, sets its rate (~4 B / clk / CU).  With a workspace the
// op first repacks [coords ; features] point-major, (B, N, P) with P = 4*ceil((3+C)/4), so that a neighbour's
// channels are one contiguous row: the gather is then 16-byte loads (4x fewer address lookups per byte), each
// lane keeps its element's channels in registers and the stores out[ch][e] are coalesced across lanes as before.
__global__ __launch_bounds__(256) void sa_pack_points_kernel(int c, int n, int p, const float *__restrict__ coords,
                                                             const float *__restrict__ feat, long long bs_f, int ld_f,
                                                             float *__restrict__ ws) {
  // thread = (point, 3 four-channel columns): twelve coalesced row reads in flight, three 16-byte stores into the row
  const int pt = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.z;
  if (pt >= n) return;
  float v[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int ch = blockIdx.y * 12 + i;
    v[i] = ch < 3 ? coords[((size_t)bi * 3 + ch) * n + pt]
                  : (ch < 3 + c ? feat[(size_t)bi * bs_f + (size_t)(ch - 3) * ld_f + pt] : 0.f);
  }
  float4 *dst = reinterpret_cast<float4 *>(ws + ((size_t)bi * n + pt) * p) + blockIdx.y * 3;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if ((blockIdx.y * 3 + i) * 4 < p) dst[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
}

constexpr int SAQ = 12;  // float4 columns of a point-major row per thread (48 channels in registers)
__global__ __launch_bounds__(256) void sa_group_pm_kernel(int c, int n, int m, int u, int p,
                                                          const float *__restrict__ ws, const float *__restrict__ centers,
                                                          const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z, mu = m * u;
  if (e >= mu) return;
  const int src = idx[(size_t)bi * mu + e];
  const int q0 = blockIdx.y * SAQ, q1 = min(q0 + SAQ, p >> 2);  // this block's float4 columns of the row
  const float4 *row = reinterpret_cast<const float4 *>(ws + ((size_t)bi * n + src) * p);
  float4 v[SAQ];
#pragma unroll
  for (int q = 0; q < SAQ; ++q)
    if (q0 + q < q1) v[q] = row[q0 + q];
  if (blockIdx.y == 0) {
    const int j = e / u;
    const float *cc = centers + (size_t)bi * 3 * m;
    v[0].x -= cc[j]; v[0].y -= cc[m + j]; v[0].z -= cc[2 * m + j];
  }
  float *ob = out + (size_t)bi * (c + 3) * mu + e;
  const int rows = c + 3;
#pragma unroll
  for (int q = 0; q < SAQ; ++q) {
    const int ch = (q0 + q) * 4;
    if (q0 + q < q1) {
      if (ch + 0 < rows) ob[(size_t)(ch + 0) * mu] = v[q].x;
      if (ch + 1 < rows) ob[(size_t)(ch + 1) * mu] = v[q].y;
      if (ch + 2 < rows) ob[(size_t)(ch + 2) * mu] = v[q].z;
      if (ch + 3 < rows) ob[(size_t)(ch + 3) * mu] = v[q].w;
    }
  }
}

// --- LDS-staged path (round 6) ---------------------------------------------------------------------
// The levels that materialise the grouped tensor are small (1024 / 256 / 64 points): a channel row of ALL the shape's points is 4 KB or
// less.  A workgroup owns SAG_R channel rows of one shape: it brings them into LDS with LDS-DMA (global_load_lds_dwordx4: a
// wave-instruction moves 64 x 16 bytes from per-lane addresses straight into 1 KB of LDS, no VGPR in between; all of a wave's pieces in
// flight, one wait), then every (centre, neighbour) element of its block reads its SAG_R values from LDS -- 64 random dwords in two
// cycles, where a scattered 4-byte load from global memory costs the address unit a line per lane (~4 B / clk / CU: what bounds
// sa_group_kernel) -- and the stores out[ch][e] are coalesced across lanes as before.  17.3 -> 11.7 us at level 1 of the denoisers, B = 16
// (35 MB written at 3 TB/s: the write stream is what is left; profiles/r06_g1_query_and_gather.txt).
constexpr int SAG_R = 3;        // channel rows per workgroup (the three coordinate rows are workgroup 0's)
constexpr int SAG_EB = 4096;    // (centre, neighbour) elements per workgroup
__global__ __launch_bounds__(256) void sa_group_lds_kernel(int c, int n, int m, int u, int dma, const float *__restrict__ coords,
                                                           const float *__restrict__ centers, const float *__restrict__ feat,
                                                           long long bs_f, int ld_f, const int *__restrict__ idx, float *__restrict__ out) {
  extern __shared__ __align__(16) float sag_tile[];   // [SAG_R][n] (+ 1 KB of slack)
  const int bi = blockIdx.z, ch0 = blockIdx.x * SAG_R, rows = c + 3, nr = min(SAG_R, rows - ch0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t mu = (size_t)m * u;
  auto row_of = [&](int ch) { return ch < 3 ? coords + ((size_t)bi * 3 + ch) * n : feat + (size_t)bi * bs_f + (size_t)(ch - 3) * ld_f; };
  if (dma) {                                          // n % 256 == 0, rows 16-byte aligned: whole 1-KB pieces
    const int ppr = n >> 8;                           // pieces per row
    for (int pc = wave; pc < nr * ppr; pc += 4) {
      const int r = pc / ppr, p = pc - r * ppr;
      const float *g = row_of(ch0 + r) + p * 256 + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                       (__attribute__((address_space(3))) void *)(sag_tile + (size_t)r * n + p * 256), 16, 0, 0);
    }
  } else {
    for (int r = 0; r < nr; ++r) {
      const float *row = row_of(ch0 + r);
      for (int k = threadIdx.x; k < n; k += 256) sag_tile[(size_t)r * n + k] = row[k];
    }
  }
  const int e0 = blockIdx.y * SAG_EB, e1 = (int)min((size_t)e0 + SAG_EB, mu);
  const int *ib = idx + (size_t)bi * mu;
  int src[SAG_EB / 256];
#pragma unroll
  for (int k = 0; k < SAG_EB / 256; ++k) {            // the indices are in flight with the rows
    const int e = e0 + threadIdx.x + k * 256;
    src[k] = e < e1 ? ib[e] : 0;
  }
  __syncthreads();                                    // (drains the DMA: vmcnt(0))
  float *ob = out + ((size_t)bi * rows + ch0) * mu;
  const float *cc = centers + (size_t)bi * 3 * m;
#pragma unroll
  for (int k = 0; k < SAG_EB / 256; ++k) {
    const int e = e0 + threadIdx.x + k * 256;
    if (e < e1) {
      float v[SAG_R];
#pragma unroll
      for (int r = 0; r < SAG_R; ++r) v[r] = r < nr ? sag_tile[(size_t)r * n + src[k]] : 0.f;
      if (ch0 == 0) {                                 // group(coords) - centre (ball_query.py:27); SAG_R >= 3: all three rows are here
        const int j = e / u;
#pragma unroll
        for (int r = 0; r < 3; ++r) v[r] = v[r] - cc[(size_t)r * m + j];
      }
#pragma unroll
      for (int r = 0; r < SAG_R; ++r)
        if (r < nr) ob[(size_t)r * mu + e] = v[r];
    }
  }
}

static inline int sa_row_floats(int c) { return ((c + 3 + 3) / 4) * 4; }

extern "C" size_t bdm_sa_group_workspace_bytes(int b, int c, int n) {
  return (size_t)b * n * sa_row_floats(c) * sizeof(float);
}

extern "C" int bdm_sa_group(int b, int c, int n, int m, int u, const float *coords, const float *centers,
                            const float *features, long long bs_f, int ld_f, const int *indices, float *out,
                            void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 1 && u >= 1, "sa_group: bad sizes");
  if (b == 0) return BDM_OK;
  if (workspace != nullptr) {
    const int p = sa_row_floats(c);
    BDM_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "sa_group: workspace must be 16-byte aligned");
    hipLaunchKernelGGL(sa_pack_points_kernel, dim3(cdiv(n, 256), cdiv(p / 4, 3), b), dim3(256), 0, (hipStream_t)stream, c, n, p,
                       coords, features, bs_f, ld_f, (float *)workspace);
    hipLaunchKernelGGL(sa_group_pm_kernel, dim3(cdiv(m * u, 256), cdiv(p / 4, SAQ), b), dim3(256), 0, (hipStream_t)stream, c,
                       n, m, u, p, (const float *)workspace, centers, indices, out);
    return launch_status("sa_group");
  }
  // SAG_R rows of the shape's points fit LDS (96 KB at 8192 points); below 256 points a row is a fraction of a DMA piece and the whole
  // level is launch-bound either way (measured at B = 16, kernel trace: 1024 points 17.3 -> 11.7 us, 256 points 8.9 -> 7.8, 64 points 6.0 -> 6.5)
  if (n >= 256 && n <= 8192 && (long long)m * u <= (1ll << 24)) {
    const int dma = (n % 256 == 0) && (ld_f % 4 == 0) && (bs_f % 4 == 0) && ((reinterpret_cast<size_t>(coords) & 15) == 0) &&
                    (c == 0 || (reinterpret_cast<size_t>(features) & 15) == 0);
    const size_t smem = sizeof(float) * SAG_R * (size_t)n + 1024;
    BDM_ALLOW_LDS(sa_group_lds_kernel, smem);
    hipLaunchKernelGGL(sa_group_lds_kernel, dim3(cdiv(c + 3, SAG_R), cdiv(m * u, SAG_EB), b), dim3(256), smem, (hipStream_t)stream, c, n, m, u,
                       dma, coords, centers, features, bs_f, ld_f, indices, out);
    return launch_status("sa_group");
  }
  int gy = c < 32 ? (c < 1 ? 1 : c) : 32;
  hipLaunchKernelGGL(sa_group_kernel, dim3(cdiv(m * u, 256), gy, b), dim3(256), 0, (hipStream_t)stream, c, n, m, u,
                     coords, centers, features, bs_f, ld_f, indices, out);
  return launch_status("sa_group");
}

// =====================================================================================
// Small helpers: broadcast a per-shape vector along points; strided copy; axpby-free add
// =====================================================================================
__global__ void bcast_rows_kernel(int c, int l, const float *__restrict__ v, int ld_v, float *__restrict__ y,
                                  long long bs_y, int ld_y) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (col >= l) return;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y)
    y[(size_t)bi * bs_y + (size_t)ci * ld_y + col] = v[(size_t)bi * ld_v + ci];
}
extern "C" int bdm_broadcast_rows(int b, int c, int l, const float *v, int ld_v, float *y, long long bs_y,
                                  int ld_y, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && l >= 1, "broadcast_rows: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(bcast_rows_kernel, dim3(cdiv(l, 256), c < 64 ? c : 64, b), dim3(256), 0, (hipStream_t)stream, c,
                     l, v, ld_v, y, bs_y, ld_y);
  return launch_status("broadcast_rows");
}

__global__ void copy_rows_kernel(int c, int l, const float *__restrict__ x, long long bs_x, int ld_x,
                                 float *__restrict__ y, long long bs_y, int ld_y) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (col >= l) return;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y)
    y[(size_t)bi * bs_y + (size_t)ci * ld_y + col] = x[(size_t)bi * bs_x + (size_t)ci * ld_x + col];
}
extern "C" int bdm_copy_rows(int b, int c, int l, const float *x, long long bs_x, int ld_x, float *y,
                             long long bs_y, int ld_y, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && l >= 1, "copy_rows: bad sizes");
  if (b == 0 || c == 0) return BDM_OK;
  hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(l, 256), c < 64 ? c : 64, b), dim3(256), 0, (hipStream_t)stream, c,
                     l, x, bs_x, ld_x, y, bs_y, ld_y);
  return launch_status("copy_rows");
}

// torch.cat([a, b], dim=1) in ONE launch; each part is either a (c, l) row block (ld > 0) or a point-invariant column
// (ld == 0: element (shape, channel) at x[shape * bs + channel], broadcast along l) -- pvcnn.py:100-108 cat([features, temb])
__global__ void concat2_rows_kernel(int l, int c0, const float *__restrict__ x0, long long bs0, int ld0, int c1,
                                    const float *__restrict__ x1, long long bs1, int ld1, float *__restrict__ y, long long bs_y,
                                    int ld_y) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (col >= l) return;
  for (int ci = blockIdx.y; ci < c0 + c1; ci += gridDim.y) {
    float v;
    if (ci < c0) v = ld0 ? x0[(size_t)bi * bs0 + (size_t)ci * ld0 + col] : x0[(size_t)bi * bs0 + ci];
    else v = ld1 ? x1[(size_t)bi * bs1 + (size_t)(ci - c0) * ld1 + col] : x1[(size_t)bi * bs1 + (ci - c0)];
    y[(size_t)bi * bs_y + (size_t)ci * ld_y + col] = v;
  }
}
extern "C" int bdm_concat2_rows(int b, int l, int c0, const float *x0, long long bs_0, int ld_0, int c1, const float *x1,
                                long long bs_1, int ld_1, float *y, long long bs_y, int ld_y, void *stream) {
  BDM_REQUIRE(b >= 0 && l >= 1 && c0 >= 0 && c1 >= 0 && c0 + c1 >= 1, "concat2_rows: bad sizes");
  if (b == 0) return BDM_OK;
  const int c = c0 + c1;
  hipLaunchKernelGGL(concat2_rows_kernel, dim3(cdiv(l, 256), c < 64 ? c : 64, b), dim3(256), 0, (hipStream_t)stream, l, c0, x0,
                     bs_0, ld_0, c1, x1, bs_1, ld_1, y, bs_y, ld_y);
  return launch_status("concat2_rows");
}

// (B, N, C) point-major  ->  (B, C, N) channel-first  (point_cloud_model.py:65 `inputs.transpose(1, 2)`)
__global__ void transpose_kernel(int rows, int cols, const float *__restrict__ x, float *__restrict__ y) {
  __shared__ float tile[32][33];
  const int bi = blockIdx.z;
  const float *xb = x + (size_t)bi * rows * cols;
  float *yb = y + (size_t)bi * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = xb[(size_t)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) yb[(size_t)c * rows + r] = tile[threadIdx.x][i];
  }
}
extern "C" int bdm_transpose(int b, int rows, int cols, const float *x, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && rows >= 1 && cols >= 1, "transpose: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32), b), dim3(32, 8), 0, (hipStream_t)stream,
                     rows, cols, x, y);
  return launch_status("transpose");
}

// =====================================================================================
// Timestep embedding + embedf   (pvcnn_utils.py:171-185, pvcnn.py:72-76,87-88)
// =====================================================================================
__global__ void time_embed_kernel(int dim, const float *__restrict__ t, const long long *__restrict__ t64, const float *__restrict__ w0,
                                  const float *__restrict__ b0, const float *__restrict__ w2,
                                  const float *__restrict__ b2, float *__restrict__ out) {
  extern __shared__ float sh[];  // emb[dim], hid[dim]
  float *emb = sh, *hid = sh + dim;
  const int bi = blockIdx.x, i = threadIdx.x, half = dim / 2;
  if (i < dim) {
    const int fi = i < half ? i : i - half;
    // numpy float64 exp, then .float()
    const float freq = (float)exp(-(double)fi * (log(10000.0) / (double)(half - 1)));
    const float arg = (t64 ? (float)t64[bi] : t[bi]) * freq;   // (t.float(): pvcnn_utils.py:178)
    emb[i] = i < half ? sinf(arg) : cosf(arg);
  }
  __syncthreads();
  if (i < dim) {
    float a = b0[i];
    for (int k = 0; k < dim; ++k) a += w0[i * dim + k] * emb[k];
    hid[i] = a > 0.f ? a : 0.1f * a;
  }
  __syncthreads();
  if (i < dim) {
    float a = b2[i];
    for (int k = 0; k < dim; ++k) a += w2[i * dim + k] * hid[k];
    out[(size_t)bi * dim + i] = a;
  }
}
extern "C" int bdm_time_embedding(int b, int dim, const float *t, const float *w0, const float *b0,
                                  const float *w2, const float *b2, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && dim >= 4 && dim % 2 == 0 && dim <= 1024, "time_embedding: bad dim %d", dim);
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(time_embed_kernel, dim3(b), dim3((dim + 63) / 64 * 64), 2 * dim * sizeof(float),
                     (hipStream_t)stream, dim, t, (const long long *)nullptr, w0, b0, w2, b2, out);
  return launch_status("time_embedding");
}
// The same from the int64 timesteps the schedulers hand over (the `.float()` of pvcnn_utils.py:178 happens in the kernel: no cast
// launch inside a recorded reverse step).
extern "C" int bdm_time_embedding_i64(int b, int dim, const long long *t, const float *w0, const float *b0,
                                      const float *w2, const float *b2, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && dim >= 4 && dim % 2 == 0 && dim <= 1024 && t != nullptr, "time_embedding_i64: bad arguments (dim %d)", dim);
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(time_embed_kernel, dim3(b), dim3((dim + 63) / 64 * 64), 2 * dim * sizeof(float),
                     (hipStream_t)stream, dim, (const float *)nullptr, t, w0, b0, w2, b2, out);
  return launch_status("time_embedding_i64");
}

// =====================================================================================
// Voxelization.forward coordinate maths  (modules/voxelization.py:16-25, normalize=True)
// =====================================================================================
// One workgroup per shape: mean over points, max point norm, then
//   nc = clamp(((p - mean) / (2*maxnorm + eps) + 0.5) * r, 0, r-1);  vox = round-half-even(nc).
__global__ void voxel_coords_kernel(int n, int r, float eps, const float *__restrict__ coords,
                                    float *__restrict__ norm_coords, int *__restrict__ vox_coords) {
  __shared__ double shd[3][16];
  __shared__ float shf[16];
  __shared__ float s_mean[3], s_max;
  const int bi = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
  const float *p = coords + (size_t)bi * 3 * n;
  double s[3] = {0.0, 0.0, 0.0};
  // the thread's first four points stay in registers for all three passes (n <= 4 T: every level of the denoisers), loaded with all
  // twelve reads in flight; the passes were 3 x (n / T) dependent round trips of one load each
  float px[4], py[4], pz[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid + u * T, ic = i < n ? i : 0;
    px[u] = p[ic]; py[u] = p[n + ic]; pz[u] = p[2 * n + ic];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (tid + u * T < n) { s[0] += px[u]; s[1] += py[u]; s[2] += pz[u]; }
  for (int i = tid + 4 * T; i < n; i += T) { s[0] += p[i]; s[1] += p[n + i]; s[2] += p[2 * n + i]; }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    s[d] = wave_sum(s[d]);
    if (lane == 0) shd[d][wave] = s[d];
  }
  __syncthreads();
  if (tid < 3) {
    double a = 0.0;
    for (int w = 0; w < nw; ++w) a += shd[tid][w];
    s_mean[tid] = (float)(a / (double)n);
  }
  __syncthreads();
  const float mx = s_mean[0], my = s_mean[1], mz = s_mean[2];
  float best = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (tid + u * T < n) {
      const float x = px[u] - mx, y = py[u] - my, z = pz[u] - mz;
      best = fmaxf(best, sqrtf(x * x + y * y + z * z));
    }
  for (int i = tid + 4 * T; i < n; i += T) {
    const float x = p[i] - mx, y = p[n + i] - my, z = p[2 * n + i] - mz;
    best = fmaxf(best, sqrtf(x * x + y * y + z * z));
  }
  best = wave_max(best);
  if (lane == 0) shf[wave] = best;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int w = 0; w < nw; ++w) a = fmaxf(a, shf[w]);
    s_max = a;
  }
  __syncthreads();
  const float denom = s_max * 2.0f + eps;
  float *nc = norm_coords + (size_t)bi * 3 * n;
  int *vc = vox_coords + (size_t)bi * 3 * n;
  const float hi = (float)(r - 1);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = tid + u * T;
    if (i < n) {
      const float pv[3] = {px[u], py[u], pz[u]};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float v = (pv[d] - s_mean[d]) / denom + 0.5f;
        v = fminf(fmaxf(v * (float)r, 0.f), hi);
        nc[(size_t)d * n + i] = v;
        vc[(size_t)d * n + i] = (int)rintf(v);
      }
    }
  }
  for (int i = tid + 4 * T; i < n; i += T) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float v = (p[(size_t)d * n + i] - s_mean[d]) / denom + 0.5f;
      v = fminf(fmaxf(v * (float)r, 0.f), hi);
      nc[(size_t)d * n + i] = v;
      vc[(size_t)d * n + i] = (int)rintf(v);
    }
  }
}
extern "C" int bdm_voxel_coords(int b, int n, int r, float eps, const float *coords, float *norm_coords,
                                int *vox_coords, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && r >= 1, "voxel_coords: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(voxel_coords_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, n, r, eps, coords,
                     norm_coords, vox_coords);
  return launch_status("voxel_coords");
}

// =====================================================================================
// SE3d gate: s = sigmoid(W2 relu(W1 mean_grid(v)))   (se.py:8-19, with_se_relu=True)
// =====================================================================================
__global__ void row_mean_kernel(int l, const float *__restrict__ x, float *__restrict__ mean) {
  const int row = blockIdx.x;
  const float *xr = x + (size_t)row * l;
  double s = 0.0;
  for (int i = threadIdx.x; i < l; i += blockDim.x) s += xr[i];
  __shared__ double sh[16];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sh[w];
    mean[row] = (float)(a / (double)l);
  }
}
__global__ void se_fc_kernel(int c, int h, const float *__restrict__ mean, const float *__restrict__ w1,
                             const float *__restrict__ w2, float *__restrict__ gate) {
  extern __shared__ float sh[];  // s[c], hid[h]
  float *s = sh, *hid = sh + c;
  const int bi = blockIdx.x;
  for (int i = threadIdx.x; i < c; i += blockDim.x) s[i] = mean[(size_t)bi * c + i];
  __syncthreads();
  se_hidden_layer(c, h, w1, s, hid);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) gate[(size_t)bi * c + i] = se_gate_of(i, h, w2, hid);
}
// One launch: every workgroup reduces one (shape, channel) row; the LAST workgroup of a shape to finish (device-scope
// counter behind a release fence, acquire fence before reading the means) evaluates the two small FC layers.  The sums
// run in the same order as in the two-kernel form, so the gate is bit-identical; no workgroup ever waits for another.
// NOT the default: on the 8-XCD part the device-scope release fence (L2 write-back) per workgroup costs far more than the
// launch it saves (measured +0.9 ms per denoiser forward for 14 gates).
__global__ __launch_bounds__(256) void se_gate_fused_kernel(int c, int h, int l, const float *__restrict__ x,
                                                            const float *__restrict__ w1, const float *__restrict__ w2,
                                                            float *mean, int *counters, float *__restrict__ gate) {
  extern __shared__ float shf[];  // s[c], hid[h] (last workgroup only)
  __shared__ double sh[16];
  __shared__ int s_last;
  const int row = blockIdx.x, bi = row / c;
  const float *xr = x + (size_t)row * l;
  double acc = 0.0;
  for (int i = threadIdx.x; i < l; i += blockDim.x) acc += xr[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sh[w];
    mean[row] = (float)(a / (double)l);
    __threadfence();  // release: the mean is visible device-wide before the counter moves
    s_last = atomicAdd(&counters[bi], 1) == c - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire: see the other workgroups' means
  float *sv = shf, *hid = shf + c;
  const volatile float *mv = mean + (size_t)bi * c;
  for (int i = threadIdx.x; i < c; i += blockDim.x) sv[i] = mv[i];
  __syncthreads();
  se_hidden_layer(c, h, w1, sv, hid);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) gate[(size_t)bi * c + i] = se_gate_of(i, h, w2, hid);
  if (threadIdx.x == 0) counters[bi] = 0;  // ready for the next call (ordered by the kernel boundary)
}

extern "C" int bdm_se_gate(int b, int c, int hidden, int l, const float *x, const float *w1, const float *w2,
                           float *mean_ws, float *gate, int *counters, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && hidden >= 1 && l >= 1, "se_gate: bad sizes");
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (counters != nullptr) {
    hipLaunchKernelGGL(se_gate_fused_kernel, dim3(b * c), dim3(256), (c + hidden) * sizeof(float), s, c, hidden, l, x, w1, w2,
                       mean_ws, counters, gate);
    return launch_status("se_gate");
  }
  hipLaunchKernelGGL(row_mean_kernel, dim3(b * c), dim3(256), 0, s, l, x, mean_ws);
  int rc = launch_status("se_row_mean");
  if (rc) return rc;
  hipLaunchKernelGGL(se_fc_kernel, dim3(b), dim3(256), (c + hidden) * sizeof(float), s, c, hidden, mean_ws, w1, w2,
                     gate);
  return launch_status("se_fc");
}

// statistics of a second tensor (the PVConv's point branch, raw output of its 1x1 convolution) to turn into affine forms
struct GnFold {
  const double *partial;  // (b, G, S, 2) or NULL
  int S, G, l;            // slices, groups, row length (points)
  const float *gamma, *beta;
  float eps;
};

// ---- GroupNorm-folded form of the PVConv tail (pvconv.py:84-96 without an attention block) ---------------------------
// The second convolution leaves its output RAW plus the GroupNorm slice partials (bdm_conv3d_3x3x3_h2_gn).  Instead of a
// normalise + Swish pass that rewrites the grid, the two consumers apply it on the fly:
//   row_mean_gn_kernel   per (shape, channel) row: finalise the group's mean / rstd from the partials, store the row's
//                        affine form (a, b) = (gamma rstd, beta - mean gamma rstd), and reduce mean_l swish(a x + b) for SE;
//   devox_gn_fused_kernel the trilinear gather evaluates swish(a g + b) at each of the 8 corners.
// One read of the grid by each consumer, no write: the grid is never rewritten.
__global__ void row_mean_gn_kernel(int c, int l, int G, int S, const float *__restrict__ x, const double *__restrict__ partial,
                                   const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                   float *__restrict__ mean, float2 *__restrict__ coef, GnFold pf, float2 *__restrict__ pf_coef) {
  const int row = blockIdx.x, bi = row / c, ch = row % c, cg = c / G, g = ch / cg;
  __shared__ float s_ab[2];
  __shared__ double sh[16];
  if (threadIdx.x == 64 && pf.partial != nullptr) {
    // the point branch's GroupNorm (same channel count): its affine form for this row, applied by the devoxelisation kernel
    const int cgp = c / pf.G, gp = ch / cgp;
    double a = 0.0, q = 0.0;
    const double *p = pf.partial + ((size_t)bi * pf.G + gp) * pf.S * 2;
    for (int s = 0; s < pf.S; ++s) { a += p[2 * s]; q += p[2 * s + 1]; }
    const double cnt = (double)cgp * pf.l, mu = a / cnt;
    double var = q / cnt - mu * mu;
    if (var < 0) var = 0;
    const float ga = pf.gamma[ch] * (float)(1.0 / sqrt(var + (double)pf.eps));
    pf_coef[row] = make_float2(ga, pf.beta[ch] - (float)mu * ga);
  }
  if (threadIdx.x == 0) {
    double a = 0.0, q = 0.0;
    const double *p = partial + ((size_t)bi * G + g) * S * 2;
    for (int s = 0; s < S; ++s) { a += p[2 * s]; q += p[2 * s + 1]; }
    const double cnt = (double)cg * l, mu = a / cnt;
    double var = q / cnt - mu * mu;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float ga = gamma[ch] * rstd, be = beta[ch] - (float)mu * ga;
    s_ab[0] = ga; s_ab[1] = be;
    coef[row] = make_float2(ga, be);
  }
  __syncthreads();
  const float ga = s_ab[0], be = s_ab[1];
  const float *xr = x + (size_t)row * l;
  double acc = 0.0;
  if ((l & 3) == 0 && ((reinterpret_cast<uintptr_t>(xr) & 15) == 0)) {  // four independent chains per thread, 16-byte loads
    const float4 *x4 = reinterpret_cast<const float4 *>(xr);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = threadIdx.x; i < l / 4; i += blockDim.x) {
      const float4 v = x4[i];
      s0 += swishf(v.x * ga + be); s1 += swishf(v.y * ga + be); s2 += swishf(v.z * ga + be); s3 += swishf(v.w * ga + be);
    }
    acc = (double)((s0 + s1) + (s2 + s3));  // <= 128 addends per chain at l = 32768; combined in fp64 below
  } else {
    for (int i = threadIdx.x; i < l; i += blockDim.x) acc += (double)swishf(xr[i] * ga + be);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sh[w];
    mean[row] = (float)(a / (double)l);
  }
}

static int se_gate_gn_impl(int b, int c, int hidden, int l, int groups, const float *x, const void *gn_workspace, int slices,
                           const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *mean_ws,
                           float *coef, float *gate, GnFold pf, float *pf_coef, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && hidden >= 1 && l >= 1 && groups >= 1 && c % groups == 0 && slices >= 1 && slices <= GN_MAX_SLICES &&
              gn_workspace != nullptr && coef != nullptr, "se_gate_gn: bad arguments");
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(row_mean_gn_kernel, dim3(b * c), dim3(256), 0, s, c, l, groups, slices, x, (const double *)gn_workspace,
                     gamma, beta, eps, mean_ws, (float2 *)coef, pf, (float2 *)pf_coef);
  int rc = launch_status("se_row_mean_gn");
  if (rc) return rc;
  if (w1 == nullptr) return BDM_OK;  // no SE block: only the affine forms are wanted
  hipLaunchKernelGGL(se_fc_kernel, dim3(b), dim3(256), (c + hidden) * sizeof(float), s, c, hidden, mean_ws, w1, w2, gate);
  return launch_status("se_fc");
}
extern "C" int bdm_se_gate_gn(int b, int c, int hidden, int l, int groups, const float *x, const void *gn_workspace,
                              int slices, const float *gamma, const float *beta, float eps, const float *w1,
                              const float *w2, float *mean_ws, float *coef, float *gate, void *stream) {
  return se_gate_gn_impl(b, c, hidden, l, groups, x, gn_workspace, slices, gamma, beta, eps, w1, w2, mean_ws, coef, gate, GnFold{},
                         nullptr, stream);
}
// The same, and also the affine forms pf_coef (b, c, 2) of the POINT BRANCH's GroupNorm(pf_groups) from the slice partials its
// 1x1 convolution left (bdm_pointwise_conv_gn; n points per row): bdm_devoxelize_gn_gate_add_pf then normalises + Swishes the
// raw point-branch rows while adding them, and the branch needs no GroupNorm launch of its own.
extern "C" int bdm_se_gate_gn_pf(int b, int c, int hidden, int l, int groups, const float *x, const void *gn_workspace,
                                 int slices, const float *gamma, const float *beta, float eps, const float *w1,
                                 const float *w2, float *mean_ws, float *coef, float *gate, const void *pf_partial,
                                 int pf_slices, int pf_groups, int pf_n, const float *pf_gamma, const float *pf_beta,
                                 float pf_eps, float *pf_coef, void *stream) {
  BDM_REQUIRE(pf_partial != nullptr && pf_slices >= 1 && pf_groups >= 1 && c % pf_groups == 0 && pf_n >= 1 && pf_gamma && pf_beta &&
              pf_coef, "se_gate_gn_pf: bad point-branch arguments");
  GnFold pf{(const double *)pf_partial, pf_slices, pf_groups, pf_n, pf_gamma, pf_beta, pf_eps};
  return se_gate_gn_impl(b, c, hidden, l, groups, x, gn_workspace, slices, gamma, beta, eps, w1, w2, mean_ws, coef, gate, pf, pf_coef,
                         stream);
}

__device__ __forceinline__ float se_gate_from_hidden(int ci, int hidden, const float *__restrict__ w2, const float *s_hid) {
  return se_gate_of(ci, hidden, w2, s_hid);  // (se_fc.h: the one definition of the gate)
}

// se_mean != NULL: the SE block's two small FC layers (se.py:8-19) are evaluated HERE from the per-channel means -- every
// workgroup recomputes the hidden vector (c/8 dot products of length c: a few thousand MACs) instead of a separate launch;
// same summation order as se_fc_kernel, so the gate is bit-identical to the two-launch form.
__global__ void devox_gn_fused_kernel(int b, int cslots, int pblocks, int c, int n, int r, const float *__restrict__ coords,
                                      const float *__restrict__ grid, const float2 *__restrict__ coef,
                                      const float *__restrict__ gate, const float *__restrict__ se_mean, int hidden,
                                      const float *__restrict__ w1, const float *__restrict__ w2,
                                      const float *__restrict__ add, long long bs_a, int ld_a,
                                      const float2 *__restrict__ add_coef, float *__restrict__ out, long long bs_o, int ld_o) {
  __shared__ float s_hid[64];
  __shared__ float s_mv[1024];
  const int span = 8 * pblocks, wg = blockIdx.x;
  const int unit = (wg / span) * 8 + (wg % span) % 8, pb = (wg % span) / 8;
  if (unit >= b * cslots) return;
  const int bi = unit / cslots, c_first = unit % cslots;
  if (se_mean != nullptr) {  // block-uniform
    for (int k = threadIdx.x; k < c; k += blockDim.x) s_mv[k] = se_mean[(size_t)bi * c + k];
    __syncthreads();
    se_hidden_layer(c, hidden, w1, s_mv, s_hid);
    __syncthreads();
  }
  {
#pragma clang fp contract(off)
  const int i = pb * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r2 = r * r, r3 = r2 * r;
  const float *pc = coords + (size_t)bi * 3 * n;
  const float x = pc[i], y = pc[n + i], z = pc[2 * n + i];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float x1 = x - xl, y1 = y - yl, z1 = z - zl;
  const float x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
  const float w000 = x0 * y0 * z0, w001 = x0 * y0 * z1, w010 = x0 * y1 * z0, w011 = x0 * y1 * z1,
              w100 = x1 * y0 * z0, w101 = x1 * y0 * z1, w110 = x1 * y1 * z0, w111 = x1 * y1 * z1;
  const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
  const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
  const int i001 = i000 + sz, i010 = i000 + sy, i011 = i010 + sz;
  const int i100 = i000 + sx, i101 = i100 + sz, i110 = i100 + sy, i111 = i110 + sz;
  for (int ci = c_first; ci < c; ci += cslots) {
    const float *g = grid + ((size_t)bi * c + ci) * r3;
    const float2 ab = coef[(size_t)bi * c + ci];
    float s = gate ? gate[(size_t)bi * c + ci] : 1.0f;
    if (se_mean != nullptr) s = se_gate_from_hidden(ci, hidden, w2, s_hid);
    // corners with weight 0 (frac == 0 on an axis) alias an in-grid cell: their value is finite and multiplies 0
    float acc = w000 * (swishf(g[i000] * ab.x + ab.y) * s);
    acc += w001 * (swishf(g[i001] * ab.x + ab.y) * s);
    acc += w010 * (swishf(g[i010] * ab.x + ab.y) * s);
    acc += w011 * (swishf(g[i011] * ab.x + ab.y) * s);
    acc += w100 * (swishf(g[i100] * ab.x + ab.y) * s);
    acc += w101 * (swishf(g[i101] * ab.x + ab.y) * s);
    acc += w110 * (swishf(g[i110] * ab.x + ab.y) * s);
    acc += w111 * (swishf(g[i111] * ab.x + ab.y) * s);
    if (add) {
      float av = add[(size_t)bi * bs_a + (size_t)ci * ld_a + i];
      if (add_coef) {  // raw point-branch convolution output: its GroupNorm + Swish applied here
        const float2 pc2 = add_coef[(size_t)bi * c + ci];
        av = swishf(av * pc2.x + pc2.y);
      }
      acc += av;
    }
    out[(size_t)bi * bs_o + (size_t)ci * ld_o + i] = acc;
  }
}
}
// LDS-cached form: one workgroup per (shape, `cpw` channels) first evaluates val[v] = swish(a grid[v] + b) * gate for the whole
// channel grid into LDS (coalesced 16-byte reads, the transcendental once per CELL instead of once per (point, corner)), then
// every point gathers its 8 corners from LDS.  The global-memory form above issues 8 scattered 4-byte loads per (point, channel)
// and is bound by the texture-address unit (PMC: 57 % of its wave cycles are issue stalls).  Same expressions in the same
// order: bit-identical output.
__global__ __launch_bounds__(1024) void devox_gn_lds_kernel(int c, int n, int r, int cpw, const float *__restrict__ coords,
                                                            const float *__restrict__ grid, const float2 *__restrict__ coef,
                                                            const float *__restrict__ gate, const float *__restrict__ add,
                                                            long long bs_a, int ld_a, const float2 *__restrict__ add_coef,
                                                            float *__restrict__ out, long long bs_o, int ld_o) {
  extern __shared__ __align__(16) float vals[];  // [cpw][r3]
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x, T = blockDim.x;
  const int groups = (c + cpw - 1) / cpw, bi = blockIdx.x / groups, c0 = (blockIdx.x % groups) * cpw;
  const int nc = min(cpw, c - c0);
  {
#pragma clang fp contract(off)
    for (int cl = 0; cl < nc; ++cl) {
      const int ci = c0 + cl;
      const float2 ab = coef[(size_t)bi * c + ci];
      const float s = gate ? gate[(size_t)bi * c + ci] : 1.0f;
      const float4 *g4 = reinterpret_cast<const float4 *>(grid + ((size_t)bi * c + ci) * r3);
      float4 *v4 = reinterpret_cast<float4 *>(vals + (size_t)cl * r3);
      for (int e = tid; e < r3 / 4; e += T) {
        const float4 g = g4[e];
        v4[e] = make_float4(swishf(g.x * ab.x + ab.y) * s, swishf(g.y * ab.x + ab.y) * s, swishf(g.z * ab.x + ab.y) * s,
                            swishf(g.w * ab.x + ab.y) * s);
      }
    }
  }
  __syncthreads();
  {
#pragma clang fp contract(off)
    const float *pc = coords + (size_t)bi * 3 * n;
    for (int i = tid; i < n; i += T) {
      const float x = pc[i], y = pc[n + i], z = pc[2 * n + i];
      const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
      const float x1 = x - xl, y1 = y - yl, z1 = z - zl;
      const float x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
      const float w000 = x0 * y0 * z0, w001 = x0 * y0 * z1, w010 = x0 * y1 * z0, w011 = x0 * y1 * z1,
                  w100 = x1 * y0 * z0, w101 = x1 * y0 * z1, w110 = x1 * y1 * z0, w111 = x1 * y1 * z1;
      const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
      const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
      const int i001 = i000 + sz, i010 = i000 + sy, i011 = i010 + sz;
      const int i100 = i000 + sx, i101 = i100 + sz, i110 = i100 + sy, i111 = i110 + sz;
      for (int cl = 0; cl < nc; ++cl) {
        const int ci = c0 + cl;
        const float *v = vals + (size_t)cl * r3;
        // corners with weight 0 (frac == 0 on an axis) alias an in-grid cell: their value is finite and multiplies 0
        float cv[8] = {v[i000], v[i001], v[i010], v[i011], v[i100], v[i101], v[i110], v[i111]};
        lds_settle8(cv);  // all eight LDS reads have landed before the first use (common.h)
        float acc = w000 * cv[0];
        acc += w001 * cv[1];
        acc += w010 * cv[2];
        acc += w011 * cv[3];
        acc += w100 * cv[4];
        acc += w101 * cv[5];
        acc += w110 * cv[6];
        acc += w111 * cv[7];
        if (add) {
          float av = add[(size_t)bi * bs_a + (size_t)ci * ld_a + i];
          if (add_coef) {
            const float2 pc2 = add_coef[(size_t)bi * c + ci];
            av = swishf(av * pc2.x + pc2.y);
          }
          acc += av;
        }
        out[(size_t)bi * bs_o + (size_t)ci * ld_o + i] = acc;
      }
    }
  }
}

static int devox_gn_launch(int b, int c, int n, int r, const float *coords, const float *grid, const float *coef,
                           const float *gate, const float *se_mean, int hidden, const float *w1, const float *w2,
                           const float *add, long long bs_a, int ld_a, const float *add_coef, float *out, long long bs_o, int ld_o,
                           void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && n >= 1 && r >= 1 && coef != nullptr, "devoxelize_gn_gate_add: bad arguments");
  BDM_REQUIRE(se_mean == nullptr || (hidden >= 1 && hidden <= 64 && c <= 1024 && w1 != nullptr && w2 != nullptr), "devoxelize_gn_se_add: bad SE arguments");
  if (b == 0) return BDM_OK;
  const int lsel = bdm_staging_choice();  // BDM_STAGING=0 keeps the global-memory gather, =1 forces the LDS form (common.h)
  const int r3 = r * r * r;
  // measured per shape at B = 16 (tools/forward_rows.py): the LDS form wins at 16^3 (29.5 -> 21.9 us) and at 32^3 with >= 64
  // channels (70.5 -> 64.8 us); at 8^3 and for the 32-channel 32^3 layers it has too few workgroups and loses.
  const bool lds_fits = se_mean == nullptr && (r3 & 3) == 0 && r3 <= 32768 && ((reinterpret_cast<size_t>(grid) & 15) == 0);
  const bool lds_pays = r == 16 || (r == 32 && (long long)b * c >= 1024);
  if (lds_fits && lsel != 0 && (lds_pays || lsel == 1)) {
    int cpw = 8192 / r3;  // channels per workgroup: 32 KB of LDS for the small grids, one channel (128 KB) at 32^3
    cpw = cpw < 1 ? 1 : (cpw > c ? c : cpw);
    const size_t smem = sizeof(float) * (size_t)cpw * r3;
    BDM_ALLOW_LDS(devox_gn_lds_kernel, smem);
    const int T = n >= 2048 ? 1024 : (n >= 512 ? 512 : 256);
    hipLaunchKernelGGL(devox_gn_lds_kernel, dim3(b * cdiv(c, cpw)), dim3(T), smem, (hipStream_t)stream, c, n, r, cpw, coords, grid,
                       (const float2 *)coef, gate, add, bs_a, ld_a, (const float2 *)add_coef, out, bs_o, ld_o);
    return launch_status("devoxelize_gn_gate_add");
  }
  const int cslots = c < 64 ? c : 64, pblocks = cdiv(n, 256);
  hipLaunchKernelGGL(devox_gn_fused_kernel, dim3(cdiv(b * cslots, 8) * 8 * pblocks), dim3(256), 0, (hipStream_t)stream, b, cslots,
                     pblocks, c, n, r, coords, grid, (const float2 *)coef, gate, se_mean, hidden, w1, w2, add, bs_a, ld_a,
                     (const float2 *)add_coef, out, bs_o, ld_o);
  return launch_status("devoxelize_gn_gate_add");
}
extern "C" int bdm_devoxelize_gn_gate_add(int b, int c, int n, int r, const float *coords, const float *grid,
                                          const float *coef, const float *gate, const float *add, long long bs_a, int ld_a,
                                          float *out, long long bs_o, int ld_o, void *stream) {
  return devox_gn_launch(b, c, n, r, coords, grid, coef, gate, nullptr, 0, nullptr, nullptr, add, bs_a, ld_a, nullptr, out, bs_o, ld_o,
                         stream);
}
// add = RAW output of the point branch's 1x1 convolution, add_coef (b, c, 2) = its GroupNorm's affine forms (bdm_se_gate_gn_pf)
extern "C" int bdm_devoxelize_gn_gate_add_pf(int b, int c, int n, int r, const float *coords, const float *grid,
                                             const float *coef, const float *gate, const float *add, long long bs_a, int ld_a,
                                             const float *add_coef, float *out, long long bs_o, int ld_o, void *stream) {
  BDM_REQUIRE(add != nullptr && add_coef != nullptr, "devoxelize_gn_gate_add_pf: add / add_coef is NULL");
  return devox_gn_launch(b, c, n, r, coords, grid, coef, gate, nullptr, 0, nullptr, nullptr, add, bs_a, ld_a, add_coef, out, bs_o, ld_o,
                         stream);
}
// The same with the SE block's FC layers evaluated inside the kernel from the channel means of bdm_se_gate_gn(w1 = NULL).
extern "C" int bdm_devoxelize_gn_se_add(int b, int c, int n, int r, const float *coords, const float *grid, const float *coef,
                                        const float *se_mean, int hidden, const float *w1, const float *w2, const float *add,
                                        long long bs_a, int ld_a, float *out, long long bs_o, int ld_o, void *stream) {
  BDM_REQUIRE(se_mean != nullptr, "devoxelize_gn_se_add: se_mean is NULL");
  return devox_gn_launch(b, c, n, r, coords, grid, coef, nullptr, se_mean, hidden, w1, w2, add, bs_a, ld_a, nullptr, out, bs_o, ld_o,
                         stream);
}

// =====================================================================================
// PVConv tail: out = trilinear_devoxelize(grid * gate) + point_branch   (pvconv.py:95-96)
// =====================================================================================
__global__ void devox_fused_kernel(int b, int cslots, int pblocks, int c, int n, int r, const float *__restrict__ coords,
                                   const float *__restrict__ grid, const float *__restrict__ gate,
                                   const float *__restrict__ add, long long bs_a, int ld_a, float *__restrict__ out,
                                   long long bs_o, int ld_o) {
#pragma clang fp contract(off)  // same unfused order as the stand-alone operator and the oracle
  // XCD-aware 1-D launch: unit = (shape, channel slot); all point blocks of a unit get the same workgroup id mod 8, so the
  // unit's voxel rows (r^3 floats per channel) are pulled into ONE XCD's L2 instead of all eight
  const int span = 8 * pblocks, wg = blockIdx.x;
  const int unit = (wg / span) * 8 + (wg % span) % 8, pb = (wg % span) / 8;
  if (unit >= b * cslots) return;
  const int bi = unit / cslots, c_first = unit % cslots;
  const int i = pb * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r2 = r * r, r3 = r2 * r;
  const float *pc = coords + (size_t)bi * 3 * n;
  const float x = pc[i], y = pc[n + i], z = pc[2 * n + i];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float x1 = x - xl, y1 = y - yl, z1 = z - zl;
  const float x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
  const float w000 = x0 * y0 * z0, w001 = x0 * y0 * z1, w010 = x0 * y1 * z0, w011 = x0 * y1 * z1,
              w100 = x1 * y0 * z0, w101 = x1 * y0 * z1, w110 = x1 * y1 * z0, w111 = x1 * y1 * z1;
  const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
  const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
  const int i001 = i000 + sz, i010 = i000 + sy, i011 = i010 + sz;
  const int i100 = i000 + sx, i101 = i100 + sz, i110 = i100 + sy, i111 = i110 + sz;
  for (int ci = c_first; ci < c; ci += cslots) {
    const float *g = grid + ((size_t)bi * c + ci) * r3;
    const float s = gate ? gate[(size_t)bi * c + ci] : 1.0f;
    float acc = w000 * (g[i000] * s);
    acc += w001 * (g[i001] * s);
    acc += w010 * (g[i010] * s);
    acc += w011 * (g[i011] * s);
    acc += w100 * (g[i100] * s);
    acc += w101 * (g[i101] * s);
    acc += w110 * (g[i110] * s);
    acc += w111 * (g[i111] * s);
    if (add) acc += add[(size_t)bi * bs_a + (size_t)ci * ld_a + i];
    out[(size_t)bi * bs_o + (size_t)ci * ld_o + i] = acc;
  }
}
extern "C" int bdm_devoxelize_gate_add(int b, int c, int n, int r, const float *coords, const float *grid,
                                       const float *gate, const float *add, long long bs_a, int ld_a,
                                       float *out, long long bs_o, int ld_o, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && n >= 1 && r >= 1, "devoxelize_gate_add: bad sizes");
  if (b == 0) return BDM_OK;
  const int cslots = c < 64 ? c : 64, pblocks = cdiv(n, 256);
  hipLaunchKernelGGL(devox_fused_kernel, dim3(cdiv(b * cslots, 8) * 8 * pblocks), dim3(256), 0, (hipStream_t)stream, b, cslots,
                     pblocks, c, n, r, coords, grid, gate, add, bs_a, ld_a, out, bs_o, ld_o);
  return launch_status("devoxelize_gate_add");
}

// =====================================================================================
// Attention cores  (pvconv.py:40-63):  h = v * softmax(q^T k)^T, no 1/sqrt(C) scale
// =====================================================================================
// (a) the 4096-token voxel attention: fp16x3 (attention_h2.hip, the default) or bf16x6 (attention_s3.hip); the fp32-input MFMA
//     flash kernel they superseded lives in experimental/attention_fp32.hip (`make EXPERIMENTAL=1`).
// (b) small-sequence kernel (global attention: 16 tokens x 512 channels): one workgroup per shape.  q, k, v are staged
//     in LDS with coalesced reads first; the C-long dot products are split over T / L^2 channel slices whose partials
//     are reduced in a fixed order.
__global__ __launch_bounds__(1024) void attn_small_kernel(int C, int L, int nsl, const float *__restrict__ q,
                                                          const float *__restrict__ k, const float *__restrict__ v,
                                                          long long bs, int ld, float *__restrict__ out, long long bs_o,
                                                          int ld_o) {
  extern __shared__ float sm[];
  const int LL = L * L, T = blockDim.x, tid = threadIdx.x;
  float *S = sm, *Sp = S + LL, *Qs = Sp + nsl * LL, *Ks = Qs + C * L, *Vs = Ks + C * L;
  const int bi = blockIdx.x;
  const float *qb = q + (size_t)bi * bs, *kb = k + (size_t)bi * bs, *vb = v + (size_t)bi * bs;
  // four elements of each operand per step, their twelve loads in flight together (one element per step was C L / T = 8 dependent
  // round trips for the global attention's 16 tokens x 512 channels: a third of this single-workgroup-per-shape kernel)
  for (int e0 = tid; e0 < C * L; e0 += 4 * T) {
    float qv[4], kv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + u * T, C * L - 1), c = e / L, i = e - c * L;
      qv[u] = qb[(size_t)c * ld + i]; kv[u] = kb[(size_t)c * ld + i]; vv[u] = vb[(size_t)c * ld + i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e0 + u * T < C * L) { Qs[e0 + u * T] = qv[u]; Ks[e0 + u * T] = kv[u]; Vs[e0 + u * T] = vv[u]; }
  }
  __syncthreads();
  if (nsl > 1) {  // L^2 <= T / 2: thread = (slice, pair)
    const int sl = tid / LL, pr = tid % LL;
    if (sl < nsl) {
      const int i = pr / L, j = pr % L;
      float a = 0.f;
      for (int c = sl; c < C; c += nsl) a += Qs[c * L + i] * Ks[c * L + j];
      Sp[sl * LL + pr] = a;
    }
    __syncthreads();
    if (tid < LL) {
      float a = 0.f;
      for (int s2 = 0; s2 < nsl; ++s2) a += Sp[s2 * LL + tid];
      S[tid] = a;
    }
  } else {
    for (int e = tid; e < LL; e += T) {
      const int i = e / L, j = e % L;
      float a = 0.f;
      for (int c = 0; c < C; ++c) a += Qs[c * L + i] * Ks[c * L + j];
      S[e] = a;
    }
  }
  __syncthreads();
  for (int i = tid; i < L; i += T) {
    float mx = -INFINITY;
    for (int j = 0; j < L; ++j) mx = fmaxf(mx, S[i * L + j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) { const float e = expf(S[i * L + j] - mx); S[i * L + j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < L; ++j) S[i * L + j] *= inv;
  }
  __syncthreads();
  float *ob = out + (size_t)bi * bs_o;
  for (int e = tid; e < C * L; e += T) {
    const int c = e / L, i = e % L;
    float a = 0.f;
    for (int j = 0; j < L; ++j) a += Vs[c * L + j] * S[i * L + j];
    ob[(size_t)c * ld_o + i] = a;
  }
}

#ifdef BDM_EXPERIMENTAL
int attention_flash_fp32(int b, int c, int l, const float *q, const float *k, const float *v, long long bs_qkv, int ld_qkv,
                         float *out, long long bs_o, int ld_o, hipStream_t s);  // experimental/attention_fp32.hip
#endif
int attention_flash_s3(int b, int c, int l, const float *q, const float *k, const float *v, long long bs_qkv, int ld_qkv,
                       float *out, long long bs_o, int ld_o, void *workspace, hipStream_t s);  // attention_s3.hip

extern "C" int bdm_attention_core(int b, int c, int l, const float *q, const float *k, const float *v,
                                  long long bs_qkv, int ld_qkv, float *out, long long bs_o, int ld_o,
                                  void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && l >= 1, "attention_core: bad sizes");
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (workspace != nullptr && l > 64 && c <= 64)
    return attention_flash_s3(b, c, l, q, k, v, bs_qkv, ld_qkv, out, bs_o, ld_o, workspace, s);
  if (l <= 64) {
    const int nsl = 1024 / (l * l) >= 2 ? 1024 / (l * l) : 1;
    const size_t smem = sizeof(float) * ((size_t)(1 + nsl) * l * l + 3 * (size_t)c * l);
    BDM_REQUIRE(smem <= 160 * 1024, "attention_core: %d channels x %d tokens do not fit the small-sequence kernel", c, l);
    BDM_ALLOW_LDS(attn_small_kernel, smem);
    hipLaunchKernelGGL(attn_small_kernel, dim3(b), dim3(1024), smem, s, c, l, nsl, q, k, v, bs_qkv, ld_qkv, out, bs_o, ld_o);
    return launch_status("attention_small");
  }
  BDM_REQUIRE(c <= 64, "attention_core: the flash kernels support at most 64 channels at %d tokens (got %d)", l, c);
#ifdef BDM_EXPERIMENTAL
  return attention_flash_fp32(b, c, l, q, k, v, bs_qkv, ld_qkv, out, bs_o, ld_o, s);
#else
  set_error("attention_core: %d tokens without a workspace would take the fp32-input flash kernel, which is only built with "
            "`make EXPERIMENTAL=1`; pass a workspace of bdm_attention_workspace_bytes (bf16x6) or use bdm_attention_core_h2", l);
  return BDM_ERR_UNSUPPORTED;
#endif
}
