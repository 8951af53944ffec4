// pvcnn_ops.hip -- gfx950 kernels for the seven forward operators of the reference plugin
// `_pvcnn_backend` (reference: experiments/model/pvcnn/modules/functional/src/**).
//
// The reference launches grid = B blocks for every one of these (e.g. ball_query.cu:55),
// i.e. < 7 % of a 256-CU part at B = 16.  Here every operator is tiled over
// (batch x centres/points/voxels x channel chunks) so that a B = 16 call fills the chip, the
// neighbour scans are wave-cooperative (64 candidates per step, ballot-ordered appends), and
// the iterative sampler keeps its state in registers with one barrier per round.
//
// Index-producing arithmetic is written with explicit unfused IEEE ops (common.h: sqdist3)
// so that indices are bit-identical to the CPU oracle on identical inputs.
#include <stdarg.h>
#include <stdio.h>

#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

// The whole file is compiled without multiply-add contraction (also -ffp-contract=off in the
// Makefile): __fmul_rn/__fadd_rn are plain '*' and '+' in ROCm's headers and would otherwise
// be fused, breaking bit-equality of indices with the oracle.
#pragma clang fp contract(off)

namespace bdm {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bdm

extern "C" const char *bdm_last_error(void) { return bdm::g_err; }
extern "C" int bdm_abi_version(void) { return 3; }

using namespace bdm;

// =====================================================================================
// Furthest point sampling
// =====================================================================================
// One workgroup per shape (the M-1 rounds are strictly sequential).  Each thread owns PPT
// points (coordinates + running min-distance in registers).  A round is: distance update,
// per-thread best, 64-lane shuffle reduction, one LDS slot per wave, ONE barrier (slots are
// double-buffered by round parity), redundant final reduction in every wave.
//
// Ordering key: (distance bits, ~rank) with rank(k) = (k mod 512) * Q + k / 512,
// Q = ceil(n / 512): the lexicographic (k mod 512, k) preference of the reference's
// 512-thread scan + tree (sampling.cu:120-160) as a single u64 max.
// 64-lane max of a u32 in ~10 instructions: rotate-and-max inside each row of 16 lanes with DPP (no LDS crossbar
// round trips), then one readlane per row.  The result is wave-uniform (SGPR).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  unsigned t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
  v = v > t ? v : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);  // row_ror:4
  v = v > t ? v : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);  // row_ror:2
  v = v > t ? v : t;
  t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);  // row_ror:1
  v = v > t ? v : t;
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}

// Round of the register-resident sampler, T threads x PPT points (T * PPT >= n):
//   * ownership is laid out so that a thread's points come in INCREASING rank: the per-thread best is "max distance bits,
//     first such i" (max3 + one compare/select per point) instead of a lexicographic (distance, rank) compare per point;
//   * distances two points at a time (float2 arithmetic -> v_pk_add_f32 / v_pk_mul_f32, no contraction: bit-equal to the scalar
//     (dx*dx + dy*dy) + dz*dz of the oracle);
//   * 64-lane max of the distance bits, then the rank of the winner by BALLOT: one lane holds the maximum in all but
//     degenerate rounds (equal distances), so a readlane replaces the second 64-lane reduction; ties take the reduction;
//   * <= 4 waves: every thread reads all wave slots and folds them in a pairwise tree of 64-bit keys (no second DPP reduction);
//   * the centre of round j-1 is written by thread 0 at the top of round j, from the coordinates every thread has just read.
#ifdef FPS_TIMING   // per-phase shader-clock sums of rounds 64 .. 191, wave 0 (a debug build for tools/fps_phase_probe.py only)
__device__ unsigned long long *g_fps_ts = nullptr;
extern "C" int bdm_debug_fps_timestamps(unsigned long long *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_fps_ts), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
#define FPS_STAMP(i) do { if (timed) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); acc_ts[i] += t_ - last_ts; last_ts = t_; } } while (0)
#else
#define FPS_STAMP(i)
#endif

template <int T, int PPT, bool use_lds>
__global__ __launch_bounds__(T) void fps_kernel(int n, int m, const float *__restrict__ coords, int *__restrict__ indices,
                                                float *__restrict__ centers_out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int NW = T / 64;
  constexpr int H = T >= 512 ? 1 : 512 / T;   // T < 512: a thread owns (k mod 512) = tid + T*h, h < H
  constexpr int J = PPT / H;                  // ... and k div 512 = j (T >= 512: k = tid + T*i)
  static_assert(PPT % H == 0 && PPT % 2 == 0, "points per thread: a multiple of 2 and of 512 / T");
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *cx = coords + (size_t)blockIdx.x * 3 * n, *cy = cx + n, *cz = cy + n;
  int *out = indices + (size_t)blockIdx.x * m;
  float *cen = centers_out ? centers_out + (size_t)blockIdx.x * 3 * m : nullptr;

  uint2 *slots = reinterpret_cast<uint2 *>(smem_raw);  // [2][16] (distance bits, ~rank)
  float *sx = reinterpret_cast<float *>(smem_raw + 2 * 16 * sizeof(uint2));
  float *sy = sx + n, *sz = sy + n;

  const unsigned Q = (unsigned)((n + 511) / 512);
  const bool q_pow2 = (Q & (Q - 1u)) == 0u;
  const unsigned q_shift = (unsigned)__ffs((int)Q) - 1u;
  auto point_of = [&](int i) -> int { return T >= 512 ? tid + T * i : tid + T * (i / J) + 512 * (i % J); };
  f2 px[PPT / 2], py[PPT / 2], pz[PPT / 2];
  unsigned dist[PPT];  // running minimum as BITS: squared distances are >= +0, so their bits order like unsigned integers
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = point_of(i);
    float x = 0.f, y = 0.f, z = 0.f;
    dist[i] = 0u;  // a slot beyond n: distance 0 for ever, and point 0 (rank 0) wins every all-zero round
    if (k < n) {
      x = cx[k]; y = cy[k]; z = cz[k];
      if (use_lds) { sx[k] = x; sy[k] = y; sz[k] = z; }
      dist[i] = __float_as_uint(1e38f);  // sampling.cpp:53-54
    }
    px[i >> 1][i & 1] = x; py[i >> 1][i & 1] = y; pz[i >> 1][i & 1] = z;
  }
  if (tid == 0) out[0] = 0;
  __syncthreads();

  int cur = 0;
#ifdef FPS_TIMING
  unsigned long long acc_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ts = 0;
#endif
  for (int j = 1; j <= m; ++j) {
#ifdef FPS_TIMING
    const bool timed = j >= 64 && j < 192 && g_fps_ts != nullptr;
    if (timed) { __builtin_amdgcn_s_waitcnt(0); last_ts = __builtin_readcyclecounter(); }
#endif
    float x1, y1, z1;
    if (use_lds) { x1 = sx[cur]; y1 = sy[cur]; z1 = sz[cur]; }
    else { x1 = cx[cur]; y1 = cy[cur]; z1 = cz[cur]; }
    if (tid == 0 && cen) { cen[j - 1] = x1; cen[m + j - 1] = y1; cen[2 * m + j - 1] = z1; }
    if (j == m) break;
    FPS_STAMP(0);   // centre fetched
    const f2 qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
    unsigned bd = 0u;
#pragma unroll
    for (int i = 0; i < PPT / 2; ++i) {
      const f2 dx = px[i] - qx, dy = py[i] - qy, dz = pz[i] - qz;
      const f2 d = (dx * dx + dy * dy) + dz * dz;
      const unsigned a = min(__float_as_uint(d[0]), dist[2 * i]), b = min(__float_as_uint(d[1]), dist[2 * i + 1]);
      dist[2 * i] = a;
      dist[2 * i + 1] = b;
      const unsigned ab = a > b ? a : b;
      bd = bd > ab ? bd : ab;
    }
    FPS_STAMP(1);   // distances + per-thread maximum
    int bi = PPT - 1;
#pragma unroll
    for (int i = PPT - 2; i >= 0; --i) bi = dist[i] == bd ? i : bi;  // first (= best-ranked) point at the maximum
    const unsigned k = (unsigned)point_of(bi);
    const unsigned br = 0xFFFFFFFFu - ((k & 511u) * Q + (k >> 9));
    FPS_STAMP(2);   // per-thread argmax + rank key
    unsigned wd = wave_max_u32(bd), wr;
    {
      const unsigned long long hit = __ballot(bd == wd);
      if ((hit & (hit - 1ull)) == 0ull) wr = (unsigned)__builtin_amdgcn_readlane((int)br, (int)__builtin_ctzll(hit));
      else wr = wave_max_u32(bd == wd ? br : 0u);
    }
    FPS_STAMP(3);   // wave maximum + winner's rank
    if (NW > 1) {
      uint2 *slot = slots + (j & 1) * 16;
      if (lane == 0) slot[wave] = make_uint2(wd, wr);
      __syncthreads();
      FPS_STAMP(4); // slot write + barrier
      if (NW <= 4) {
        // every thread folds all wave slots: (distance bits, ~rank) as one 64-bit key, pairwise tree (independent compares)
        unsigned long long key[NW];
#pragma unroll
        for (int w = 0; w < NW; w += 2) {
          const uint4 v = *reinterpret_cast<const uint4 *>(slot + w);
          key[w] = ((unsigned long long)v.x << 32) | v.y;
          key[w + 1] = ((unsigned long long)v.z << 32) | v.w;
        }
#pragma unroll
        for (int span = 1; span < NW; span *= 2)
#pragma unroll
          for (int w = 0; w + span < NW; w += 2 * span) key[w] = key[w] > key[w + span] ? key[w] : key[w + span];
        wd = (unsigned)(key[0] >> 32);
        wr = (unsigned)key[0];
      } else {
        const uint2 sv = lane < NW ? slot[lane] : make_uint2(0u, 0u);
        wd = wave_max_u32(sv.x);
        const unsigned long long hit = __ballot(sv.x == wd && lane < NW);
        if ((hit & (hit - 1ull)) == 0ull) wr = (unsigned)__builtin_amdgcn_readlane((int)sv.y, (int)__builtin_ctzll(hit));
        else wr = wave_max_u32((sv.x == wd && lane < NW) ? sv.y : 0u);
      }
    }
    FPS_STAMP(5);   // slots read + folded
    const unsigned rank = 0xFFFFFFFFu - wr;
    // rank -> point index; Q is a power of two for the usual sizes: shifts instead of two integer divisions per round
    cur = q_pow2 ? (int)(((rank & (Q - 1u)) << 9) + (rank >> q_shift)) : (int)((rank % Q) * 512u + rank / Q);
    cur = __builtin_amdgcn_readfirstlane(cur);
    if (tid == 0) out[j] = cur;
    FPS_STAMP(6);   // rank -> index, output
  }
#ifdef FPS_TIMING
  if (g_fps_ts != nullptr && lane == 0)
    for (int i = 0; i < 8; ++i) g_fps_ts[((size_t)blockIdx.x * 16 + wave) * 8 + i] = acc_ts[i];
#endif
}

// Threads per shape, measured at B = 16 on MI355X (us per call, round 3): n=4096,m=1024: 128 thr 981 | 256: 711 | 512: 705 | 1024: 780
// (round 2's 1024 x 4 kernel: 1130); n=1024,m=256: 64: 143 | 128: 128 | 256: 124 | 512: 158 (round 2: 197); n=8192: 512: 973 | 1024: 988.
static int fps_threads(int n) {
  if (n <= 128) return 64;
  if (n <= 2048) return 256;
  if (n <= 8192) return 512;
  return 1024;
}

extern "C" int bdm_furthest_point_sampling(int b, int n, int m, const float *coords, int *indices,
                                           float *centers_out, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && m >= 0, "fps: bad sizes b=%d n=%d m=%d", b, n, m);
  BDM_REQUIRE(n <= 16384, "fps: n=%d exceeds the 16384-point limit of the register-resident sampler", n);
  if (b == 0 || m == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  int T = fps_threads(n);
  int ppt = T >= 512 ? 2 : 2 * (512 / T);  // smallest legal count: even, a multiple of 512 / T
  while ((long long)T * ppt < n) ppt *= 2;
  const bool use_lds = n <= 12288;
  const size_t smem = 2 * 16 * sizeof(uint2) + (use_lds ? (size_t)3 * n * sizeof(float) : 0);
#define FPS_LAUNCH_(TT, P, LDS)                                                                          \
  if (T == TT && ppt == P && use_lds == LDS) {                                                          \
    BDM_ALLOW_LDS((fps_kernel<TT, P, LDS>), smem);                                                      \
    hipLaunchKernelGGL((fps_kernel<TT, P, LDS>), dim3(b), dim3(TT), smem, s, n, m, coords, indices,     \
                       centers_out);                                                                    \
    return launch_status("fps");                                                                        \
  }
#define FPS_LAUNCH(TT, P) FPS_LAUNCH_(TT, P, true)
  FPS_LAUNCH(64, 16) FPS_LAUNCH(64, 32)
  FPS_LAUNCH(128, 8) FPS_LAUNCH(128, 16) FPS_LAUNCH(128, 32)
  FPS_LAUNCH(256, 4) FPS_LAUNCH(256, 8) FPS_LAUNCH(256, 16) FPS_LAUNCH(256, 32)
  FPS_LAUNCH(512, 2) FPS_LAUNCH(512, 4) FPS_LAUNCH(512, 8) FPS_LAUNCH(512, 16) FPS_LAUNCH(512, 32)
  FPS_LAUNCH(1024, 2) FPS_LAUNCH(1024, 4) FPS_LAUNCH(1024, 8) FPS_LAUNCH(1024, 16) FPS_LAUNCH_(1024, 16, false)
#undef FPS_LAUNCH
#undef FPS_LAUNCH_
  set_error("fps: no kernel for %d threads x %d points per thread (n=%d)", T, ppt, n);
  return BDM_ERR_UNSUPPORTED;
}

// =====================================================================================
// Gather
// =====================================================================================
__global__ void gather_kernel(int c, int n, int m, const float *__restrict__ feat,
                              const int *__restrict__ idx, float *__restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= m) return;
  const int src = idx[(size_t)bi * m + j];
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y)
    out[((size_t)bi * c + ci) * m + j] = feat[((size_t)bi * c + ci) * n + src];
}

extern "C" int bdm_gather_features_forward(int b, int c, int n, int m, const float *features,
                                           const int *indices, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0, "gather: bad sizes");
  if (b == 0 || c == 0 || m == 0) return BDM_OK;
  dim3 grid(cdiv(m, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, m, features, indices, out);
  return launch_status("gather");
}

// =====================================================================================
// Ball query: one wave per centre, 64 candidates per step, ballot-ordered append
// =====================================================================================
// Two phases per tile of up to 4096 candidates, both branch-free in the part that scales with N:
//  scan   - the tile is staged in LDS so that lane l owns the S = len/64 CONSECUTIVE candidates [l*S, l*S+S) and
//           step s reads them at consecutive LDS addresses (row pitch 65: conflict-free both ways).  One step costs
//           three LDS reads shared by the CW centres of the wave, packed-fp32 distance arithmetic for two centres
//           per instruction (v_pk_add/mul_f32: the same IEEE single operations as the scalar form, so indices stay
//           bit-identical to the oracle) and "bits = 2*bits + hit" per centre -- no ballots, no appends.
//  emit   - per centre, a wave prefix sum of the per-lane hit counts gives every hit its rank in candidate order
//           (lane-major = index order); the first u ranks are stored.  Runs once per tile, not once per step.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int BQ_TILE = 4096, BQ_PITCH = 65;

// inclusive prefix sum over the 64 lanes on the DPP network (no LDS round trips): shifts within rows of 16, then
// row broadcasts of lane 15 / 31 to the following rows (the scan sequence of LLVM's AMDGPU atomic optimizer)
__device__ __forceinline__ int wave_inclusive_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}

// bits = 2 * bits + (d < r2): the compare's carry goes straight into an add-with-carry
__device__ __forceinline__ void push_hit(unsigned &bits, float d, float r2) {
  asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(d), "s"(r2) : "vcc");
}

template <int CW, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ball_query_kernel(int n, int m, float r2, int u, const float *__restrict__ centers,
                                                         const float *__restrict__ points, int *__restrict__ neighbors) {
  static_assert(CW % 2 == 0, "CW even");
  constexpr int CP = CW / 2;
  __shared__ float sx[64 * BQ_PITCH], sy[64 * BQ_PITCH], sz[64 * BQ_PITCH];
  const int lane = threadIdx.x & 63;
  const int j0 = (blockIdx.x * WAVES + (threadIdx.x >> 6)) * CW;
  const int bi = blockIdx.y;
  const float *px = points + (size_t)bi * 3 * n, *py = px + n, *pz = py + n;
  const float *cb = centers + (size_t)bi * 3 * m;
  f32x2 cx[CP], cy[CP], cz[CP];
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    const int j = min(j0 + c, m - 1);  // slots past the last centre repeat it; nothing is stored for them
    cx[c >> 1][c & 1] = cb[j]; cy[c >> 1][c & 1] = cb[m + j]; cz[c >> 1][c & 1] = cb[2 * m + j];
  }
  int cnt[CW], first[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) { cnt[c] = (j0 + c < m) ? 0 : u; first[c] = 0; }  // a full list takes no more hits
  bool open = j0 < m;

  for (int t0 = 0; t0 < n; t0 += BQ_TILE) {
    const int len = min(BQ_TILE, n - t0);
    int lg = 0;  // S = 2^lg candidates per lane, S * 64 >= len
    while ((64 << lg) < len) ++lg;
    const int S = 1 << lg;
    if (t0) __syncthreads();
    for (int k = threadIdx.x; k < (64 << lg); k += WAVES * 64) {
      const bool in = k < len;
      const int a = (k & (S - 1)) * BQ_PITCH + (k >> lg);
      // beyond the end: +inf coordinates -> distance inf or NaN -> never a hit
      sx[a] = in ? px[t0 + k] : INFINITY; sy[a] = in ? py[t0 + k] : INFINITY; sz[a] = in ? pz[t0 + k] : INFINITY;
    }
    __syncthreads();
    if (!open) continue;
    unsigned wa[CW], wb[CW];  // hit bits of steps [0,32) and [32,64), newest in bit 0
#pragma unroll
    for (int c = 0; c < CW; ++c) { wa[c] = 0u; wb[c] = 0u; }
    const int na = min(S, 32), nbits = S - na;
    // four steps per batch: their 12 LDS reads are issued together, ahead of the arithmetic
    auto scan = [&](unsigned (&W)[CW], int s0, int s1) {
      const float *qx = sx + lane, *qy = sy + lane, *qz = sz + lane;
      int st = s0;
      for (; st + 4 <= s1; st += 4) {
        float xs[4], ys[4], zs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xs[i] = qx[(st + i) * BQ_PITCH]; ys[i] = qy[(st + i) * BQ_PITCH]; zs[i] = qz[(st + i) * BQ_PITCH];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int cp = 0; cp < CP; ++cp) {
            const f32x2 dx = cx[cp] - xs[i], dy = cy[cp] - ys[i], dz = cz[cp] - zs[i];
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
            push_hit(W[2 * cp], d[0], r2);
            push_hit(W[2 * cp + 1], d[1], r2);
          }
        }
      }
      for (; st < s1; ++st) {
        const float xk = qx[st * BQ_PITCH], yk = qy[st * BQ_PITCH], zk = qz[st * BQ_PITCH];
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const f32x2 dx = cx[cp] - xk, dy = cy[cp] - yk, dz = cz[cp] - zk;
          const f32x2 d = (dx * dx + dy * dy) + dz * dz;
          push_hit(W[2 * cp], d[0], r2);
          push_hit(W[2 * cp + 1], d[1], r2);
        }
      }
    };
    scan(wa, 0, na);
    scan(wb, 32, S);
    bool any_open = false;
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      unsigned long long w = ((unsigned long long)wa[c] << nbits) | wb[c];  // step s at bit S-1-s
      if (cnt[c] < u && __ballot(w != 0ull)) {
        const int mine = __popcll(w);
        const int incl = wave_inclusive_sum(mine);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        const int kbase = t0 + (lane << lg) + S - 1;  // candidate of bit b: kbase - b
        if (cnt[c] == 0) {
          const int fl = __ffsll((long long)__ballot(mine > 0)) - 1;
          first[c] = __shfl(kbase - (63 - __clzll((long long)w)), fl, 64);
        }
        int pos = cnt[c] + incl - mine;
        int *nb = neighbors + ((size_t)bi * m + j0 + c) * u;
        while (w && pos < u) {
          const int bit = 63 - __clzll((long long)w);
          nb[pos++] = kbase - bit;
          w ^= 1ull << bit;
        }
        cnt[c] += total;
      }
      any_open |= cnt[c] < u;
    }
    open = any_open;
  }
  // slots never reached: first hit (ball_query.cu:40-44), or 0 when there was none
  // (the reference's output tensor is torch::zeros, ball_query.cpp:20-22)
#pragma unroll
  for (int c = 0; c < CW; ++c) {
    if (j0 + c >= m) break;
    int *nb = neighbors + ((size_t)bi * m + j0 + c) * u;
    const int fill = cnt[c] > 0 ? first[c] : 0;
    for (int s = lane; s < u; s += 64)
      if (s >= cnt[c]) nb[s] = fill;
  }
}

extern "C" int bdm_ball_query(int b, int n, int m, float radius, int u, const float *centers,
                              const float *points, int *neighbors, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && m >= 0 && u >= 1, "ball_query: bad sizes");
  if (b == 0 || m == 0) return BDM_OK;
  const float r2 = radius * radius;  // ball_query.cpp:24
  // centres per wave / waves per workgroup: a workgroup stages one 4096-candidate tile (50 KB of LDS) for all its
  // centres, so big launches use 8 waves x 4 centres (2 workgroups = 16 waves per CU), small ones spread out
  const long long centres = (long long)b * m;
  hipStream_t st = (hipStream_t)stream;
  if (centres >= 16384) {
    hipLaunchKernelGGL((ball_query_kernel<4, 8>), dim3(cdiv(m, 32), b), dim3(512), 0, st, n, m, r2, u, centers, points, neighbors);
  } else if (centres >= 4096) {
    hipLaunchKernelGGL((ball_query_kernel<4, 4>), dim3(cdiv(m, 16), b), dim3(256), 0, st, n, m, r2, u, centers, points, neighbors);
  } else {
    hipLaunchKernelGGL((ball_query_kernel<2, 4>), dim3(cdiv(m, 8), b), dim3(256), 0, st, n, m, r2, u, centers, points, neighbors);
  }
  return launch_status("ball_query");
}

// =====================================================================================
// Grouping
// =====================================================================================
__global__ void grouping_kernel(int c, int n, int mu, const float *__restrict__ feat,
                                const int *__restrict__ idx, float *__restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (e >= mu) return;
  const int src = idx[(size_t)bi * mu + e];
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y)
    out[((size_t)bi * c + ci) * mu + e] = feat[((size_t)bi * c + ci) * n + src];
}

extern "C" int bdm_grouping_forward(int b, int c, int n, int m, int u, const float *features,
                                    const int *indices, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0 && u >= 0, "grouping: bad sizes");
  if (b == 0 || c == 0 || m * u == 0) return BDM_OK;
  dim3 grid(cdiv(m * u, 256), c < 32 ? c : 32, b);
  hipLaunchKernelGGL(grouping_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, m * u, features,
                     indices, out);
  return launch_status("grouping");
}

// =====================================================================================
// Three nearest neighbours + inverse-squared-distance interpolation
// =====================================================================================
// One workgroup = 64 points; its four waves each scan a quarter of the staged centres (4x shorter serial chain, 4 waves
// per SIMD) in batches of four with one test of the batch minimum against the current third-best, then the per-wave
// top-3 lists are merged by (distance, index): the same triple as the reference's sequential strict-< scan.
__device__ __forceinline__ void top3_insert(float d, int idx, float &b0, float &b1, float &b2, int &i0, int &i1, int &i2) {
  if (d < b2) {
    b2 = d; i2 = idx;
    if (d < b1) {
      b2 = b1; i2 = i1; b1 = d; i1 = idx;
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = idx; }
    }
  }
}
__device__ __forceinline__ bool lex_less(float d, int i, float e, int j) { return d < e || (d == e && i < j); }

__global__ __launch_bounds__(256) void three_nn_search_kernel(int m, int n, const float *__restrict__ points,
                                                              const float *__restrict__ centers, int *__restrict__ indices,
                                                              float *__restrict__ weights) {
  constexpr int CH = 1024;
  __shared__ __align__(16) float sc[3 * CH];  // centres, staged in chunks; padded with +inf (never selected)
  __shared__ float md[3][3][64];
  __shared__ int mi[3][3][64];
  const int bi = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const float *cb = centers + (size_t)bi * 3 * m;
  const float *pb = points + (size_t)bi * 3 * n;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) { ux = pb[j]; uy = pb[n + j]; uz = pb[2 * n + j]; }
  // running bests hold float values (the reference keeps them in doubles, :37)
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int base = 0; base < m; base += CH) {
    const int len = min(CH, m - base), lenp = (len + 15) & ~15;
    __syncthreads();
    for (int t = threadIdx.x; t < lenp; t += 256) {
      const bool in = t < len;
      sc[t] = in ? cb[base + t] : INFINITY; sc[CH + t] = in ? cb[m + base + t] : INFINITY;
      sc[2 * CH + t] = in ? cb[2 * m + base + t] : INFINITY;
    }
    __syncthreads();
    const int q = lenp >> 2, k0 = wave * q;
    for (int k = k0; k < k0 + q; k += 4) {
      const float4 cx = *reinterpret_cast<const float4 *>(sc + k), cy = *reinterpret_cast<const float4 *>(sc + CH + k),
                   cz = *reinterpret_cast<const float4 *>(sc + 2 * CH + k);
      const float d0 = sqdist3(ux, uy, uz, cx.x, cy.x, cz.x), d1 = sqdist3(ux, uy, uz, cx.y, cy.y, cz.y),
                  d2 = sqdist3(ux, uy, uz, cx.z, cy.z, cz.z), d3 = sqdist3(ux, uy, uz, cx.w, cy.w, cz.w);
      if (fminf(fminf(d0, d1), fminf(d2, d3)) < b2) {
        top3_insert(d0, base + k, b0, b1, b2, i0, i1, i2);
        top3_insert(d1, base + k + 1, b0, b1, b2, i0, i1, i2);
        top3_insert(d2, base + k + 2, b0, b1, b2, i0, i1, i2);
        top3_insert(d3, base + k + 3, b0, b1, b2, i0, i1, i2);
      }
    }
  }
  if (wave > 0) {
    md[wave - 1][0][lane] = b0; md[wave - 1][1][lane] = b1; md[wave - 1][2][lane] = b2;
    mi[wave - 1][0][lane] = i0; mi[wave - 1][1][lane] = i1; mi[wave - 1][2][lane] = i2;
  }
  __syncthreads();
  if (wave > 0 || j >= n) return;
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const float d = md[w][s][lane];
      const int idx = mi[w][s][lane];
      if (d == INFINITY) continue;  // unfilled slot
      if (lex_less(d, idx, b2, i2)) {
        b2 = d; i2 = idx;
        if (lex_less(d, idx, b1, i1)) {
          b2 = b1; i2 = i1; b1 = d; i1 = idx;
          if (lex_less(d, idx, b0, i0)) { b1 = b0; i1 = i0; b0 = d; i0 = idx; }
        }
      }
    }
  b0 = fmaxf(fminf(1e10f, b0), 1e-10f);
  b1 = fmaxf(fminf(1e10f, b1), 1e-10f);
  b2 = fmaxf(fminf(1e10f, b2), 1e-10f);
  const float d0d1 = __fmul_rn(b0, b1), d0d2 = __fmul_rn(b0, b2), d1d2 = __fmul_rn(b1, b2);
  const float inv = __fdiv_rn(1.0f, __fadd_rn(__fadd_rn(d0d1, d0d2), d1d2));
  float *w = weights + (size_t)bi * 3 * n;
  int *id = indices + (size_t)bi * 3 * n;
  w[j] = __fmul_rn(d1d2, inv);         id[j] = i0;
  w[n + j] = __fmul_rn(d0d2, inv);     id[n + j] = i1;
  w[2 * n + j] = __fmul_rn(d0d1, inv); id[2 * n + j] = i2;
}

__global__ void three_nn_apply_kernel(int c, int m, int n, const float *__restrict__ feat, long long bs_f,
                                      int ld_f, const int *__restrict__ indices,
                                      const float *__restrict__ weights, float *__restrict__ out,
                                      long long bs_o, int ld_o) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= n) return;
  const int *id = indices + (size_t)bi * 3 * n;
  const float *w = weights + (size_t)bi * 3 * n;
  const int i0 = id[j], i1 = id[n + j], i2 = id[2 * n + j];
  const float w0 = w[j], w1 = w[n + j], w2 = w[2 * n + j];
  const float *f = feat + (size_t)bi * bs_f;
  float *o = out + (size_t)bi * bs_o;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) {
    const float *fr = f + (size_t)ci * ld_f;
    o[(size_t)ci * ld_o + j] =
        __fadd_rn(__fadd_rn(__fmul_rn(fr[i0], w0), __fmul_rn(fr[i1], w1)), __fmul_rn(fr[i2], w2));
  }
}

extern "C" int bdm_three_nn_search(int b, int m, int n, const float *points, const float *centers,
                                   int *indices, float *weights, void *stream) {
  BDM_REQUIRE(b >= 0 && m >= 1 && n >= 0, "three_nn_search: bad sizes");
  if (b == 0 || n == 0) return BDM_OK;
  dim3 grid(cdiv(n, 64), b);
  hipLaunchKernelGGL(three_nn_search_kernel, grid, dim3(256), 0, (hipStream_t)stream, m, n, points, centers, indices, weights);
  return launch_status("three_nn_search");
}

extern "C" int bdm_three_nn_apply(int b, int c, int m, int n, const float *features, long long bs_f, int ld_f,
                                  const int *indices, const float *weights, float *out, long long bs_o,
                                  int ld_o, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && m >= 1 && n >= 0, "three_nn_apply: bad sizes");
  if (b == 0 || c == 0 || n == 0) return BDM_OK;
  dim3 grid(cdiv(n, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(three_nn_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, m, n, features, bs_f,
                     ld_f, indices, weights, out, bs_o, ld_o);
  return launch_status("three_nn_apply");
}

// PointNetFPModule input assembly in ONE launch (pointnet.py:104-112): out0 = cat[interpolate(centers_features), skip],
// out1 = interpolate(centers_temb), all three with the same (indices, weights).  Channel index space of the grid's y axis:
// [0, c_a) interpolated features, [c_a, c_a + c_s) skip rows (plain copy), [c_a + c_s, c_a + c_s + c_t) interpolated t_emb.
__global__ void fp_assemble_kernel(int n, int c_a, const float *__restrict__ fa, long long bs_a, int ld_a, int c_s,
                                   const float *__restrict__ fs, long long bs_s, int ld_s, int c_t,
                                   const float *__restrict__ ft, long long bs_t, int ld_t, const int *__restrict__ indices,
                                   const float *__restrict__ weights, float *__restrict__ out0, long long bs_0, int ld_0,
                                   float *__restrict__ out1, long long bs_1, int ld_1) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (j >= n) return;
  const int *id = indices + (size_t)bi * 3 * n;
  const float *w = weights + (size_t)bi * 3 * n;
  const int i0 = id[j], i1 = id[n + j], i2 = id[2 * n + j];
  const float w0 = w[j], w1 = w[n + j], w2 = w[2 * n + j];
#pragma unroll 4
  for (int ci = blockIdx.y; ci < c_a + c_s + c_t; ci += gridDim.y) {
    if (ci < c_a) {
      const float *fr = fa + (size_t)bi * bs_a + (size_t)ci * ld_a;
      out0[(size_t)bi * bs_0 + (size_t)ci * ld_0 + j] =
          __fadd_rn(__fadd_rn(__fmul_rn(fr[i0], w0), __fmul_rn(fr[i1], w1)), __fmul_rn(fr[i2], w2));
    } else if (ci < c_a + c_s) {
      out0[(size_t)bi * bs_0 + (size_t)ci * ld_0 + j] = fs[(size_t)bi * bs_s + (size_t)(ci - c_a) * ld_s + j];
    } else {
      const int ct = ci - c_a - c_s;
      const float *fr = ft + (size_t)bi * bs_t + (size_t)ct * ld_t;
      out1[(size_t)bi * bs_1 + (size_t)ct * ld_1 + j] =
          __fadd_rn(__fadd_rn(__fmul_rn(fr[i0], w0), __fmul_rn(fr[i1], w1)), __fmul_rn(fr[i2], w2));
    }
  }
}

extern "C" int bdm_fp_assemble(int b, int m, int n, const int *indices, const float *weights, int c_a, const float *fa,
                               long long bs_a, int ld_a, int c_s, const float *fs, long long bs_s, int ld_s, int c_t,
                               const float *ft, long long bs_t, int ld_t, float *out0, long long bs_0, int ld_0, float *out1,
                               long long bs_1, int ld_1, void *stream) {
  BDM_REQUIRE(b >= 0 && m >= 1 && n >= 0 && c_a >= 0 && c_s >= 0 && c_t >= 0, "fp_assemble: bad sizes");
  if (b == 0 || n == 0 || c_a + c_s + c_t == 0) return BDM_OK;
  const int c = c_a + c_s + c_t;
  dim3 grid(cdiv(n, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(fp_assemble_kernel, grid, dim3(256), 0, (hipStream_t)stream, n, c_a, fa, bs_a, ld_a, c_s, fs, bs_s, ld_s, c_t,
                     ft, bs_t, ld_t, indices, weights, out0, bs_0, ld_0, out1, bs_1, ld_1);
  return launch_status("fp_assemble");
}

extern "C" int bdm_three_nn_interpolate_forward(int b, int c, int m, int n, const float *points,
                                                const float *centers, const float *features, float *out,
                                                int *indices, float *weights, void *stream) {
  int rc = bdm_three_nn_search(b, m, n, points, centers, indices, weights, stream);
  if (rc) return rc;
  return bdm_three_nn_apply(b, c, m, n, features, (long long)c * m, m, indices, weights, out,
                            (long long)c * n, n, stream);
}

// =====================================================================================
// Average voxelisation, deterministic
// =====================================================================================
// Plan kernel (one workgroup per shape, LDS histogram of r^3 counters):
//   ind[i], cnt[v]  ->  start[v] (exclusive scan)  ->  per-voxel point lists, each list in
//   ascending point index (unordered atomic fill, then rank-within-list placement).
// Reduce kernel: out[c][v] = sum over list(v), in list order, of feat[c][p] * (1/cnt[v]).
extern "C" size_t bdm_voxelize_workspace_bytes(int b, int n, int r) {
  return sizeof(int) * ((size_t)b * r * r * r + 2 * (size_t)b * n);
}

__global__ void vox_plan_kernel(int n, int r, const int *__restrict__ coords, int *__restrict__ ind,
                                int *__restrict__ cnt, int *__restrict__ start, int *__restrict__ tmp,
                                int *__restrict__ sorted, int n_max, int *__restrict__ occ_index,
                                int *__restrict__ occ_list, int *__restrict__ n_occ, unsigned char *__restrict__ rowocc) {
  extern __shared__ int lcnt[];  // [r3] counters, then cursors
  __shared__ int wave_tot[16];
  const int r2 = r * r, r3 = r2 * r;
  const int bi = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int *vx = coords + (size_t)bi * 3 * n, *vy = vx + n, *vz = vy + n;
  int *id = ind + (size_t)bi * n;
  int *gc = cnt + (size_t)bi * r3;
  int *gs = start + (size_t)bi * r3;
  int *tp = tmp + (size_t)bi * n;
  int *so = sorted + (size_t)bi * n;

  for (int v = tid; v < r3; v += T) lcnt[v] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    const int v = vx[i] * r2 + vy[i] * r + vz[i];
    id[i] = v;
    atomicAdd(&lcnt[v], 1);
  }
  __syncthreads();
  // exclusive scan over r3 counters: each thread owns a contiguous run
  const int per = (r3 + T - 1) / T;
  const int lo = min(tid * per, r3), hi = min(lo + per, r3);
  int local = 0;
  for (int v = lo; v < hi; ++v) { const int cv = lcnt[v]; gc[v] = cv; local += cv; }
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wave; ++w) wave_off += wave_tot[w];
  int run = wave_off + incl - local;
  for (int v = lo; v < hi; ++v) { const int cv = lcnt[v]; gs[v] = run; lcnt[v] = run; run += cv; }
  __syncthreads();
  for (int i = tid; i < n; i += T) tp[atomicAdd(&lcnt[id[i]], 1)] = i;
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    const int v = id[i];
    const int s = gs[v], cv = gc[v];
    int rank = 0;
    for (int q = 0; q < cv; ++q) rank += tp[s + q] < i;
    so[s + rank] = i;
  }
  if (occ_index == nullptr) return;
  // ---- optional tail (bdm_voxelize_plan_full): occupied-cell compaction and row occupancy of the same shape, in the
  // launch that already owns it (what bdm_voxel_compact + bdm_voxel_row_occupancy do in two more launches)
  __syncthreads();
  int *oi = occ_index + (size_t)bi * r3;
  int *ol = occ_list + (size_t)bi * n_max;
  int occ_local = 0;
  for (int v = lo; v < hi; ++v) occ_local += gc[v] > 0;
  int oincl = occ_local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(oincl, o, 64);
    if (lane >= o) oincl += t;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = oincl;
  __syncthreads();
  int off = 0, total = 0;
  for (int w = 0; w < (T >> 6); ++w) { if (w < wave) off += wave_tot[w]; total += wave_tot[w]; }
  int orun = off + oincl - occ_local;
  for (int v = lo; v < hi; ++v) {
    if (gc[v] > 0) { oi[v] = orun; ol[orun] = v; ++orun; }
    else oi[v] = -1;
  }
  if (tid == 0) n_occ[bi] = total;
  for (int row = tid; row < r2; row += T) {
    int any = 0;
    for (int z = 0; z < r; ++z) any |= gc[row * r + z];
    rowocc[(size_t)bi * r2 + row] = any ? 1 : 0;
  }
}

// The same plan for the 32^3 grids with EIGHT workgroups per shape instead of one (the single-workgroup form walks 32 768
// counters with 1 024 threads, 32 serial cells per thread and several dependent passes: ~140 us on the critical path of every
// forward).  Workgroup s owns the cells [s * CS, (s + 1) * CS); it needs two numbers from the other slabs -- the points and
// the occupied cells that precede its range -- and derives both by itself from the coordinates (every workgroup reads all n
// points: 48 KB) and a whole-grid occupancy bitset in LDS, so the slabs never communicate.  Same outputs, bit for bit.
template <int CS>  // cells per slab
__global__ __launch_bounds__(1024) void vox_plan_slab_kernel(int n, int r, int slabs, const int *__restrict__ coords,
                                                             int *__restrict__ ind, int *__restrict__ cnt,
                                                             int *__restrict__ start, int *__restrict__ sorted, int n_max,
                                                             int *__restrict__ occ_index, int *__restrict__ occ_list,
                                                             int *__restrict__ n_occ, unsigned char *__restrict__ rowocc) {
  extern __shared__ int slab_smem[];
  constexpr int T = 1024, PER = CS / T;  // cells per thread (contiguous run)
  static_assert(CS % T == 0 && PER >= 1, "slab size must be a multiple of the workgroup size");
  const int r2 = r * r, r3 = r2 * r;
  int *lcnt = slab_smem;                                   // [CS] counters, then cursors
  unsigned *bits = reinterpret_cast<unsigned *>(lcnt + CS);  // [r3 / 32] occupancy of the whole grid
  int *lbeg = reinterpret_cast<int *>(bits + r3 / 32);     // [CS] slab-local start of each cell's list
  int *lst = lbeg + CS;                                    // [n] the slab's points grouped by cell (unordered inside a cell)
  __shared__ int red[16], red2[16];
  const int bi = blockIdx.x / slabs, sl = blockIdx.x % slabs, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lo_cell = sl * CS;
  const int *vx = coords + (size_t)bi * 3 * n, *vy = vx + n, *vz = vy + n;
  int *id = ind + (size_t)bi * n;
  int *gc = cnt + (size_t)bi * r3, *gs = start + (size_t)bi * r3, *so = sorted + (size_t)bi * n;

  for (int v = tid; v < CS; v += T) lcnt[v] = 0;
  for (int w = tid; w < r3 / 32; w += T) bits[w] = 0u;
  __syncthreads();
  int before = 0;  // points whose cell precedes this slab
  for (int i = tid; i < n; i += T) {
    const int v = vx[i] * r2 + vy[i] * r + vz[i];
    if (sl == 0) id[i] = v;
    atomicOr(&bits[v >> 5], 1u << (v & 31));
    before += v < lo_cell;
    if (v >= lo_cell && v < lo_cell + CS) atomicAdd(&lcnt[v - lo_cell], 1);
  }
  // block sums: `before`, and (after the barrier) the occupied cells preceding the slab / in the whole grid
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
  if (lane == 0) red[wave] = before;
  __syncthreads();
  int occ_before = 0, occ_total = 0;
  for (int w = tid; w < r3 / 32; w += T) {
    const int pc = __popc(bits[w]);
    occ_total += pc;
    occ_before += (w < lo_cell / 32) ? pc : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { occ_before += __shfl_xor(occ_before, o, 64); occ_total += __shfl_xor(occ_total, o, 64); }
  __shared__ int red3[16];
  if (lane == 0) { red2[wave] = occ_before; red3[wave] = occ_total; }
  __syncthreads();
  int point_base = 0, occ_base = 0, occ_all = 0;
  for (int w = 0; w < T / 64; ++w) { point_base += red[w]; occ_base += red2[w]; occ_all += red3[w]; }
  if (sl == 0 && tid == 0 && n_occ != nullptr) n_occ[bi] = occ_all;
  __syncthreads();  // red / red2 are reused below

  // exclusive scans over the slab's cells (points and occupied cells): each thread owns PER contiguous cells
  int cv[PER], local = 0, olocal = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) { cv[j] = lcnt[tid * PER + j]; local += cv[j]; olocal += cv[j] > 0; }
  int incl = local, oincl = olocal;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t1 = __shfl_up(incl, o, 64), t2 = __shfl_up(oincl, o, 64);
    if (lane >= o) { incl += t1; oincl += t2; }
  }
  if (lane == 63) { red[wave] = incl; red2[wave] = oincl; }
  __syncthreads();
  int woff = 0, owoff = 0;
  for (int w = 0; w < wave; ++w) { woff += red[w]; owoff += red2[w]; }
  int run = woff + incl - local;            // slab-local start of this thread's first cell
  int orun = occ_base + owoff + oincl - olocal;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int v = lo_cell + tid * PER + j;
    gc[v] = cv[j];
    gs[v] = point_base + run;
    lcnt[tid * PER + j] = run;  // cursor for the fill below
    lbeg[tid * PER + j] = run;
    if (occ_index != nullptr) {
      if (cv[j] > 0) { occ_index[(size_t)bi * r3 + v] = orun; occ_list[(size_t)bi * n_max + orun] = v; ++orun; }
      else occ_index[(size_t)bi * r3 + v] = -1;
    }
    run += cv[j];
  }
  if (rowocc != nullptr) {  // a grid row = r consecutive cells; r / PER threads share one row
    const int any = local > 0;
    // rows of the slab: one ballot per wave covers 64 * PER cells = 64 * PER / r rows
    const unsigned long long m = __ballot(any);
    constexpr int TPR_MAX = 64;
    const int tpr = r / PER;  // threads per row (8 for r = 32, PER = 4)
    if (tpr <= TPR_MAX && (lane % tpr) == 0) {
      const unsigned long long rowmask = ((tpr == 64 ? ~0ull : ((1ull << tpr) - 1ull)) << lane);
      const int row = (lo_cell + tid * PER) / r;
      rowocc[(size_t)bi * r2 + row] = (m & rowmask) ? 1 : 0;
    }
  }
  __syncthreads();
  // unordered fill of the slab's per-cell lists, then rank-within-list placement (ascending point index inside a cell)
  for (int i = tid; i < n; i += T) {
    const int v = vx[i] * r2 + vy[i] * r + vz[i];
    if (v >= lo_cell && v < lo_cell + CS) lst[atomicAdd(&lcnt[v - lo_cell], 1)] = i;
  }
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    const int v = vx[i] * r2 + vy[i] * r + vz[i];
    if (v >= lo_cell && v < lo_cell + CS) {
      const int s0 = lbeg[v - lo_cell], cvv = lcnt[v - lo_cell] - s0;  // cursor after the fill = end of the cell's list
      int rank = 0;
      for (int q = 0; q < cvv; ++q) rank += lst[s0 + q] < i;
      so[point_base + s0 + rank] = i;
    }
  }
}

__global__ void vox_reduce_kernel(int c, int n, int r3, const float *__restrict__ feat,
                                  const int *__restrict__ cnt, const int *__restrict__ start,
                                  const int *__restrict__ sorted, float *__restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (v >= r3) return;
  const int cv = cnt[(size_t)bi * r3 + v];
  const int s = start[(size_t)bi * r3 + v];
  const int *so = sorted + (size_t)bi * n + s;
  const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;  // vox.cu:66
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) {
    const float *f = feat + ((size_t)bi * c + ci) * n;
    float acc = 0.f;
    for (int q = 0; q < cv; ++q) acc = __fadd_rn(acc, __fmul_rn(f[so[q]], inv));
    out[((size_t)bi * c + ci) * r3 + v] = acc;
  }
}

// BDM_STAGING=0 keeps the one-workgroup-per-shape plan kernel at 32^3 (common.h)
static bool vox_plan_slabs() { return bdm_staging_choice() != 0; }

extern "C" int bdm_voxelize_plan(int b, int n, int r, const int *coords, int *ind, int *cnt, void *workspace,
                                 void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && r >= 1 && r <= 32, "voxelize_plan: bad sizes (r<=32 supported)");
  BDM_REQUIRE(workspace != nullptr, "voxelize_plan: workspace is NULL");
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(workspace, b, n, r3);
  const size_t smem = (size_t)r3 * sizeof(int);
  BDM_ALLOW_LDS(vox_plan_kernel, smem);
  hipLaunchKernelGGL(vox_plan_kernel, dim3(b), dim3(1024), smem, (hipStream_t)stream, n, r, coords, ind, cnt, w.start,
                     w.tmp, w.sorted, 0, (int *)nullptr, (int *)nullptr, (int *)nullptr, (unsigned char *)nullptr);
  return launch_status("vox_plan");
}

extern "C" int bdm_voxelize_plan_full(int b, int n, int r, int n_max, const int *coords, int *ind, int *cnt, void *workspace,
                                      int *occ_index, int *occ_list, int *n_occ, unsigned char *rowocc, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && r >= 1 && r <= 32 && n_max >= 1, "voxelize_plan_full: bad sizes (r<=32 supported)");
  BDM_REQUIRE(workspace && occ_index && occ_list && n_occ && rowocc, "voxelize_plan_full: NULL buffer");
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  VoxWs w = vox_ws(workspace, b, n, r3);
  if (r == 32 && n <= 24576 && vox_plan_slabs()) {  // eight workgroups per shape (LDS: 2 x 16 KB counters / starts + 4 KB bitset + 4 n bytes)
    const size_t sm = sizeof(int) * (2 * 4096 + (size_t)r3 / 32 + (size_t)n);
    BDM_ALLOW_LDS(vox_plan_slab_kernel<4096>, sm);
    hipLaunchKernelGGL(vox_plan_slab_kernel<4096>, dim3(b * 8), dim3(1024), sm, (hipStream_t)stream, n, r, 8, coords, ind, cnt,
                       w.start, w.sorted, n_max, occ_index, occ_list, n_occ, rowocc);
    return launch_status("vox_plan_full");
  }
  const size_t smem = (size_t)r3 * sizeof(int);
  BDM_ALLOW_LDS(vox_plan_kernel, smem);
  hipLaunchKernelGGL(vox_plan_kernel, dim3(b), dim3(1024), smem, (hipStream_t)stream, n, r, coords, ind, cnt, w.start,
                     w.tmp, w.sorted, n_max, occ_index, occ_list, n_occ, rowocc);
  return launch_status("vox_plan_full");
}

extern "C" int bdm_avg_voxelize_forward(int b, int c, int n, int r, const float *features, const int *coords,
                                        float *out, int *ind, int *cnt, void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 1 && r >= 1 && r <= 32, "avg_voxelize: bad sizes (r<=32 supported)");
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r;
  int rc = bdm_voxelize_plan(b, n, r, coords, ind, cnt, workspace, stream);
  VoxWs w = vox_ws(workspace, b, n, r3);
  if (rc || c == 0) return rc;
  dim3 grid(cdiv(r3, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(vox_reduce_kernel, grid, dim3(256), 0, s, c, n, r3, features, cnt, w.start, w.sorted, out);
  return launch_status("vox_reduce");
}

// =====================================================================================
// Trilinear devoxelisation (inference form)
// =====================================================================================
__global__ void devox_kernel(int c, int n, int r, const float *__restrict__ coords,
                             const float *__restrict__ grid, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int bi = blockIdx.z;
  if (i >= n) return;
  const int r2 = r * r, r3 = r2 * r;
  const float *pc = coords + (size_t)bi * 3 * n;
  const float x = pc[i], y = pc[n + i], z = pc[2 * n + i];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float x1 = x - xl, y1 = y - yl, z1 = z - zl;
  const float x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
  const float x0y0 = __fmul_rn(x0, y0), x0y1 = __fmul_rn(x0, y1), x1y0 = __fmul_rn(x1, y0), x1y1 = __fmul_rn(x1, y1);
  const float w000 = __fmul_rn(x0y0, z0), w001 = __fmul_rn(x0y0, z1), w010 = __fmul_rn(x0y1, z0),
              w011 = __fmul_rn(x0y1, z1), w100 = __fmul_rn(x1y0, z0), w101 = __fmul_rn(x1y0, z1),
              w110 = __fmul_rn(x1y1, z0), w111 = __fmul_rn(x1y1, z1);
  const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
  const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
  const int i001 = i000 + sz, i010 = i000 + sy, i011 = i010 + sz;
  const int i100 = i000 + sx, i101 = i100 + sz, i110 = i100 + sy, i111 = i110 + sz;
  for (int ci = blockIdx.y; ci < c; ci += gridDim.y) {
    const float *g = grid + ((size_t)bi * c + ci) * r3;
    float acc = __fmul_rn(w000, g[i000]);
    acc = __fadd_rn(acc, __fmul_rn(w001, g[i001]));
    acc = __fadd_rn(acc, __fmul_rn(w010, g[i010]));
    acc = __fadd_rn(acc, __fmul_rn(w011, g[i011]));
    acc = __fadd_rn(acc, __fmul_rn(w100, g[i100]));
    acc = __fadd_rn(acc, __fmul_rn(w101, g[i101]));
    acc = __fadd_rn(acc, __fmul_rn(w110, g[i110]));
    acc = __fadd_rn(acc, __fmul_rn(w111, g[i111]));
    out[((size_t)bi * c + ci) * n + i] = acc;
  }
}

extern "C" int bdm_trilinear_devoxelize_forward(int b, int c, int n, int r, const float *coords,
                                                const float *grid, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && r >= 1, "devoxelize: bad sizes");
  if (b == 0 || c == 0 || n == 0) return BDM_OK;
  dim3 g(cdiv(n, 256), c < 64 ? c : 64, b);
  hipLaunchKernelGGL(devox_kernel, g, dim3(256), 0, (hipStream_t)stream, c, n, r, coords, grid, out);
  return launch_status("devoxelize");
}
