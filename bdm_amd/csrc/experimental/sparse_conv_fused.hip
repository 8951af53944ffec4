// experimental/sparse_conv_fused.hip (built with `make EXPERIMENTAL=1` only: correct, deterministic, measured SLOWER than GEMM + gather --
// DESIGN.md negative results)
// the FIRST 3x3x3 convolution of a PVConv in ONE kernel, without the 27x-expanded intermediate.
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

#include "../../../include/bdm_hip.h"
#include "../common.h"

using namespace bdm;

#include "../sparse_h2_common.h"

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
