// sparse_conv_fused.hip -- the FIRST 3x3x3 convolution of a PVConv in ONE kernel, without the 27x-expanded intermediate.
//
// sparse_conv.hip evaluates  out[:, v] = bias + sum_tap W_tap . vox[:, v + tap]  on the occupied cells as a batched GEMM
// Y (n_occ x 27*Cout) followed by an output-stationary gather; Y is written and read back once (27 * Cout * 4 bytes per
// occupied cell: 0.66 GB per 64 -> 64 layer at B = 16), which is what that pair of kernels is bound by
// (profiles/r01_pmc_hbm_traffic.csv).  Here the same products are formed by the matrix cores and accumulated straight into
// the OUTPUT tile, held in LDS:
//
//   workgroup = (shape, slab of SX output x-planes, 32 output channels); LDS holds the slab's SX*R*R x 32 accumulators
//   (initialised with the bias).  The occupied cells are sorted by voxel index (x major), so the cells that can reach the
//   slab through kernel column kx are ONE contiguous range of the compact list (planes x0+kx-1 .. x0+SX+kx-2).
//   for kx in 0..2:  for rounds of 4 chunks (one chunk of 32 occupied cells per wave):
//       K loop:   acc[ky,kz] (32 cells x 32 channels) += X_chunk (32 x Cin) . W[kx,ky,kz] (Cin x 32)     9 accumulators
//                 (v_mfma_f32_32x32x16_f16, fp16x3 split operands, see below)
//       scatter:  for each (ky,kz): row m of acc[ky,kz] is added to the LDS accumulator of cell
//                 (ux-kx+1, uy-ky+1, uz-kz+1) -- for ONE tap the map cell -> output cell is injective, so within a
//                 phase no two lanes (of any wave) touch the same address: plain ds_add, no ordering ambiguity.  Phases
//                 are separated by workgroup barriers, so every output cell receives its contributions in a fixed order
//                 (kx, round, ky, kz): the result is bit-reproducible run to run (no float atomics racing).
//   finally the slab is written channel-first (coalesced runs along the voxel index).
// Every (occupied cell, tap) product is computed exactly once, by the workgroup that owns its output plane.
//
// Arithmetic: fp16x3, as conv3d_h2.hip -- an fp32 operand times a power of two is stored as hi + lo (two fp16 terms,
// 22 signed bits), a product is lo.hi + hi.lo + hi.hi accumulated in fp32.  Weights carry a per-output-channel scale
// (pack time); the activations (voxel-mean point features, range unknown a priori) are scaled by ONE power of two per
// call derived on the device from max |x| (bdm_sparse_voxel_features_f32 accumulates it with an integer atomic max --
// order independent, hence deterministic) and split when the operand is loaded.  fp32-grade: <= 3e-7 relative L2 vs
// fp64 in tests/test_hip_dense.py.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// LDS-cached feature gather shared with sparse_conv.hip (defined there)
bool bdm_sparse_features_lds_launch(int out_kind, int b, int c, int n, int r3, int n_max, const float *features, long long bs_f,
                                    int ld_f, const int *cnt, const int *start, const int *sorted, const int *occ_list,
                                    const int *n_occ, void *out, unsigned *amax, hipStream_t stream, int *rc);

namespace {

__device__ __forceinline__ void split2s(float v, unsigned short &h, unsigned short &l) {
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  h = __builtin_bit_cast(unsigned short, hi);
  l = __builtin_bit_cast(unsigned short, lo);
}

__device__ __forceinline__ void split_record(const float4 &p, const float4 &q, float s, f16x8 &hi, f16x8 &lo) {
  const float v[8] = {p.x * s, p.y * s, p.z * s, p.w * s, q.x * s, q.y * s, q.z * s, q.w * s};
  unsigned short h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split2s(v[j], h[j], l[j]);
  uint4 ph, pl;
  ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
  pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
  hi = *reinterpret_cast<const f16x8 *>(&ph);
  lo = *reinterpret_cast<const f16x8 *>(&pl);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (vmcnt(0)), i.e.
// it waits for the global loads issued as PREFETCH for later steps and so exposes their whole latency at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// power of two s with amax * s in [2^14, 2^15)  (1 when amax is 0 / not finite)
__device__ __forceinline__ float act_scale_from_max(float amax) {
  if (!(amax > 0.f) || !(amax < INFINITY)) return 1.f;
  int ex;
  (void)frexpf(amax, &ex);  // amax = f * 2^ex, f in [0.5, 1)
  return ldexpf(1.f, 15 - ex);
}

}  // namespace

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
// weights (Cout, Cin, 3,3,3) fp32 -> [G][27][2][Cout] records of 8 fp16 (hi / lo of w * 2^e[co]); inv_scale[co] = 2^-e[co]
// ---------------------------------------------------------------------------------------------------
__global__ void sparse_fused_weight_scale_kernel(int cout, int cin, const float *__restrict__ w, float *__restrict__ scale,
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
    const bool ok = mx > 0.f && mx < INFINITY;
    if (ok) (void)frexpf(mx, &ex);
    const int e = ok ? 10 - ex : 0;  // mx * 2^e in [2^9, 2^10)
    scale[co] = ldexpf(1.0f, e);
    inv_scale[co] = ldexpf(1.0f, -e);
  }
}
__global__ void sparse_fused_pack_kernel(int cout, int cin, const float *__restrict__ w, const float *__restrict__ scale,
                                         unsigned short *__restrict__ wq) {
  const int G = (cin + 7) / 8;
  const long long total = (long long)G * 27 * cout;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(e % cout), tap = (int)((e / cout) % 27), g = (int)(e / (27ll * cout));
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = g * 8 + j;
      const float v = ci < cin ? w[((size_t)co * cin + ci) * 27 + tap] * scale[co] : 0.f;  // exact scaling
      split2s(v, h[j], l[j]);
    }
    unsigned short *ph = wq + ((((size_t)g * 27 + tap) * 2 + 0) * cout + co) * 8;
    unsigned short *pl = wq + ((((size_t)g * 27 + tap) * 2 + 1) * cout + co) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ph[j] = h[j]; pl[j] = l[j]; }
  }
}
extern "C" size_t bdm_sparse_conv_fused_weight_elems(int cout, int cin) { return (size_t)((cin + 7) / 8) * 27 * 2 * cout * 8; }
extern "C" int bdm_sparse_conv_fused_pack_weights(int cout, int cin, const float *w, void *packed, float *scale_ws,
                                                  float *inv_scale, void *stream) {
  BDM_REQUIRE(cout >= 1 && cin >= 1 && scale_ws != nullptr && inv_scale != nullptr, "sparse_conv_fused_pack_weights: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sparse_fused_weight_scale_kernel, dim3(cout), dim3(256), 0, s, cout, cin, w, scale_ws, inv_scale);
  hipLaunchKernelGGL(sparse_fused_pack_kernel, dim3(512), dim3(256), 0, s, cout, cin, w, scale_ws, (unsigned short *)packed);
  return launch_status("sparse_conv_fused_pack_weights");
}

// ---------------------------------------------------------------------------------------------------
// the fused convolution
// ---------------------------------------------------------------------------------------------------
template <int R, int SX>
__global__ __launch_bounds__(256) void sparse_conv_fused_kernel(int G, int Cout, int n_max, const float4 *__restrict__ xr,
                                                                const float *__restrict__ amax,
                                                                const uint4 *__restrict__ wq,
                                                                const float *__restrict__ inv_sw,
                                                                const int *__restrict__ occ_list,
                                                                const int *__restrict__ n_occ,
                                                                const float *__restrict__ bias, float *__restrict__ out) {
  constexpr int R2 = R * R, R3 = R2 * R, NV = SX * R2, LD = 33;
  constexpr int WREC = 2 * 18 * 32, WI = (WREC + 255) / 256;  // weight records of one K=16 step: [lh][tap*2+split][32 channels]
  extern __shared__ __align__(16) float smem[];
  float *accs = smem;                                            // [NV][LD], then 64 scratch floats (sink of masked-off lanes)
  constexpr int SINK = NV * LD;
  uint4 *Ws = reinterpret_cast<uint4 *>(smem + (NV * LD + 64 + 3) / 4 * 4);  // [2][18][32]
  __shared__ int s_bound[SX + 3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int x0 = blockIdx.x * SX, co0 = blockIdx.y * 32, bi = blockIdx.z;
  const int co = co0 + li;
  const bool co_ok = co < Cout;
  const int nocc = min(n_occ[bi], n_max);
  const int *ol = occ_list + (size_t)bi * n_max;

  // plane boundaries of the sorted compact list: s_bound[t] = first row whose voxel lies in plane >= x0 - 1 + t.
  // One coalesced sweep over the list: a row that starts a new plane records itself for every boundary it crosses.
  if (tid < SX + 3) s_bound[tid] = (x0 - 1 + tid <= 0) ? 0 : nocc;
  __syncthreads();
  for (int row = tid; row < nocc; row += 256) {
    const int pl = ol[row] / R2, pp = row > 0 ? ol[row - 1] / R2 : -1;
    if (pl != pp) {
      const int t_lo = max(pp + 1, x0 - 1) - (x0 - 1), t_hi = min(pl, x0 + SX + 1) - (x0 - 1);
      for (int t = max(t_lo, 0); t <= t_hi; ++t) s_bound[t] = row;  // planes pp+1 .. pl all start at this row
    }
  }
  for (int e = tid; e < NV * 32; e += 256) {
    const int c = e & 31, p = e >> 5;
    accs[p * LD + c] = (bias && co0 + c < Cout) ? bias[co0 + c] : 0.f;
  }
  __syncthreads();

  const float sx = act_scale_from_max(amax[bi]);
  const float post = (co_ok ? inv_sw[co] : 0.f) * (1.0f / sx);  // powers of two: exact
  const int K16 = (G + 1) >> 1;
  const float4 *xb = xr + (size_t)bi * G * n_max * 2;

  for (int kx = 0; kx < 3; ++kx) {
    // input planes [x0 + kx - 1, x0 + SX + kx - 1) reach output planes [x0, x0 + SX) through kernel column kx
    const int lo = s_bound[kx], hi = s_bound[kx + SX];
    const int nchunks = (hi - lo + 31) >> 5;
    for (int c0 = 0; c0 < nchunks; c0 += 4) {
      const int chunk = c0 + wave;
      const bool active = chunk < nchunks;
      const bool multi = nchunks - c0 > 1;  // more than one wave scatters in this round -> phases need barriers
      const int row = lo + chunk * 32 + li;
      const bool rv = active && row < hi;
      const int u = rv ? ol[row] : -1;
      f32x16 acc[9];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
      // ---- K loop: the step's weight records go through LDS once per workgroup (register-prefetched), the chunk's own
      //      activation records straight to registers ----------------------------------------------------------------
      uint4 wreg[WI], wnxt[WI];
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 xc0, xc1, xn0 = z4, xn1 = z4, xm0 = z4, xm1 = z4;
      auto wload = [&](int ks, uint4 *wdst) {
#pragma unroll
        for (int i = 0; i < WI; ++i) {
          const int e = tid + i * 256;
          const int col = e & 31, ts = (e >> 5) % 18, hh = e / (18 * 32);
          const int g = 2 * ks + hh;
          const bool ok = e < WREC && g < G && co0 + col < Cout;
          const uint4 v = wq[ok ? (((size_t)g * 27 + kx * 9) * 2 + ts) * Cout + co0 + col : 0];
          const unsigned k = ok ? 0xFFFFFFFFu : 0u;
          wdst[i] = make_uint4(v.x & k, v.y & k, v.z & k, v.w & k);
        }
      };
      auto xload = [&](int ks, float4 &d0, float4 &d1) {
        const int g = 2 * ks + lh;
        const bool ok = rv && g < G;
        const size_t xo = ((size_t)(ok ? g : 0) * n_max + (ok ? row : 0)) * 2;
        const float4 p = xb[xo], q = xb[xo + 1];
        const float m = ok ? 1.f : 0.f;  // (a select on the whole vector made the compiler go through scratch memory)
        d0 = make_float4(p.x * m, p.y * m, p.z * m, p.w * m);
        d1 = make_float4(q.x * m, q.y * m, q.z * m, q.w * m);
      };
      // loads run TWO steps ahead of the matrix work (one workgroup per CU at r = 32: nothing else hides the latency)
      wload(0, wreg);
      xload(0, xc0, xc1);
      if (K16 > 1) { wload(1, wnxt); xload(1, xn0, xn1); }
      for (int ks = 0; ks < K16; ++ks) {
        lds_barrier();  // the previous step's readers are done with Ws
#pragma unroll
        for (int i = 0; i < WI; ++i)
          if (tid + i * 256 < WREC) Ws[tid + i * 256] = wreg[i];
        lds_barrier();
#pragma unroll
        for (int i = 0; i < WI; ++i) wreg[i] = wnxt[i];
        if (ks + 2 < K16) { wload(ks + 2, wnxt); xload(ks + 2, xm0, xm1); }
        {  // waves without a chunk multiply zeros: a branch here makes the compiler shuttle the nine accumulators
           // between the two register files on every iteration (measured: 1.6 us per step)
          f16x8 ah, al;
          split_record(xc0, xc1, sx, ah, al);
          f16x8 bh[9], bl[9];
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const uint4 wh = Ws[(lh * 18 + t * 2) * 32 + li], wl = Ws[(lh * 18 + t * 2 + 1) * 32 + li];
            bh[t] = *reinterpret_cast<const f16x8 *>(&wh);
            bl[t] = *reinterpret_cast<const f16x8 *>(&wl);
          }
          // term-major: consecutive MFMAs hit nine independent accumulators (smallest terms first: lo.hi, hi.lo, hi.hi)
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t], acc[t], 0, 0, 0);
        }
        xc0 = xn0; xc1 = xn1;
        xn0 = xm0; xn1 = xm1;
      }
      // ---- scatter targets of this lane's 16 accumulator rows: LDS cell of the centre tap | uy << 16 | uz << 24, or -1
      int tgt[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * lh;
        const int um = __shfl(u, m, 64);
        int t = -1;
        if (um >= 0) {
          const int ux = um / R2, uy = (um / R) % R, uz = um % R;
          const int ox = ux - kx + 1 - x0;
          if (ox >= 0 && ox < SX) t = ((ox * R + uy) * R + uz) | (uy << 16) | (uz << 24);
        }
        tgt[i] = t;
      }
      lds_barrier();  // every wave has left the K loop (Ws is free) before the accumulators are touched
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        if (active && co_ok) {
          const int dy = 1 - t9 / 3, dz = 1 - t9 % 3;  // output cell = input cell + (dy, dz) in (y, z)
          // For ONE tap the 32 x 16 (row, lane) targets of a wave -- and those of the other waves -- are distinct cells,
          // so the read-modify-write needs no atomics: all reads, then all writes.
          int ad[16];
          float v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int t = tgt[i];
            const int oy = ((t >> 16) & 0xFF) + dy, oz = ((t >> 24) & 0xFF) + dz;
            // masked-off rows go to a per-lane sink cell: every lane issues the same 16 reads and 16 writes, no branches
            ad[i] = (t >= 0 && oy >= 0 && oy < R && oz >= 0 && oz < R) ? ((t & 0xFFFF) + dy * R + dz) * LD + li : SINK + lane;
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = accs[ad[i]];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] += acc[t9][i] * post;
#pragma unroll
          for (int i = 0; i < 16; ++i) accs[ad[i]] = v[i];
        }
        if (multi) lds_barrier();
      }
      if (!multi) lds_barrier();
    }
  }
  __syncthreads();
  float *ob = out + (size_t)bi * Cout * R3 + (size_t)x0 * R2;
  for (int e = tid; e < NV * 8; e += 256) {  // 4 consecutive cells of one channel per thread: 16-byte stores
    const int c = e / (NV / 4), p = (e % (NV / 4)) * 4;
    if (co0 + c < Cout) {
      const float4 v = make_float4(accs[p * LD + c], accs[(p + 1) * LD + c], accs[(p + 2) * LD + c], accs[(p + 3) * LD + c]);
      *reinterpret_cast<float4 *>(ob + (size_t)(co0 + c) * R3 + p) = v;
    }
  }
}

extern "C" int bdm_sparse_conv_fused(int b, int cin, int cout, int r, int n_max, const void *xr, const float *amax,
                                     const void *packed_w, const float *inv_scale, const int *occ_list, const int *n_occ,
                                     const float *bias, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && n_max >= 1 && amax != nullptr && inv_scale != nullptr,
              "sparse_conv_fused: bad arguments");
  if (r != 8 && r != 16 && r != 32) {
    set_error("sparse_conv_fused: resolution %d unsupported (8, 16, 32)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const int G = (cin + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
#define FUSED_LAUNCH(R, SX)                                                                                          \
  do {                                                                                                               \
    const size_t smem = sizeof(float) * (((size_t)(SX) * (R) * (R) * 33 + 64 + 3) / 4 * 4) + 16 * (size_t)(2 * 18 * 32); \
    BDM_ALLOW_LDS((sparse_conv_fused_kernel<R, SX>), smem);                                                          \
    hipLaunchKernelGGL((sparse_conv_fused_kernel<R, SX>), dim3((R) / (SX), cdiv(cout, 32), b), dim3(256), smem, s, G, \
                       cout, n_max, (const float4 *)xr, amax, (const uint4 *)packed_w, inv_scale, occ_list, n_occ,   \
                       bias, out);                                                                                   \
  } while (0)
  if (r == 32) FUSED_LAUNCH(32, 1);
  else if (r == 16) FUSED_LAUNCH(16, 2);
  else FUSED_LAUNCH(8, 4);
#undef FUSED_LAUNCH
  return launch_status("sparse_conv_fused");
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
                                                             float *__restrict__ Y) {
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
      const float post = nn < N ? inv_sw[nn % Cout] * inv_sx : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wr * MX + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M && nn < N) Yb[(size_t)m * N + nn] = acc[x][y][r] * post;
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

extern "C" int bdm_sparse_conv_gemm_h2(int b, int n_max, int cin, int cout, const void *xh, const float *amax, const void *packed_w,
                                       const float *inv_scale, const int *n_occ, float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && n_max >= 1 && cin >= 1 && cout >= 1 && amax != nullptr && inv_scale != nullptr, "sparse_conv_gemm_h2: bad arguments");
  if (b == 0) return BDM_OK;
  const int n27 = 27 * cout, G = (cin + 7) / 8;
  hipStream_t s = (hipStream_t)stream;
  if (n_max <= 256)
    hipLaunchKernelGGL(sparse_gemm_h2_kernel<64>, dim3(cdiv(n27, 128), cdiv(n_max, 64), b), dim3(256), 0, s, n_max, G, n27, cout,
                       (const uint4 *)xh, amax, (const uint4 *)packed_w, inv_scale, n_occ, y);
  else
    hipLaunchKernelGGL(sparse_gemm_h2_kernel<128>, dim3(cdiv(n27, 128), cdiv(n_max, 128), b), dim3(256), 0, s, n_max, G, n27, cout,
                       (const uint4 *)xh, amax, (const uint4 *)packed_w, inv_scale, n_occ, y);
  return launch_status("sparse_conv_gemm_h2");
}
