// attention_s3.hip -- flash attention on the bf16 matrix cores at fp32-grade accuracy ("bf16x6", s3_split.h).
//
// Replaces the fp32-MFMA flash kernel of dense_ops.hip for the long sequences (the 16^3 = 4096-token voxel attention
// of sa_layers.1.0, pvconv.py:40-63, and the 197-token ViT heads) when the caller supplies a workspace.
//   pre-pass : q, k -> S3 records (B, C/8, 3, L) of 8 channels x bf16 (hi, mid, lo); v -> three bf16 planes
//              (B, 3, CP, Lp) with the KEY index contiguous.  One read of q, k, v; ~1.5x their size written.
//   kernel   : per 32-key step   S^T = K^T Q  : keys on the accumulator rows, queries on the lane
//                                softmax      : registers + one partner-lane exchange (online max / sum)
//                                O^T += V P   : the probability tile never leaves registers -- accumulator registers
//                                               8jj..8jj+7 of a lane ARE the 8 k-slots of MFMA jj's B operand once
//                                               split into bf16 triples; V is read from LDS in the same key order:
//                                               slot (lh, i) <-> key 16jj + 4lh + (i & 3) + 8 (i >> 2).
//              Every fp32 product is six exact bf16 partial products (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid)
//              accumulated in fp32, smallest first: ~2^-22 relative, 2.7x the fp32-input MFMA rate.
#include "../../include/bdm_hip.h"
#include "common.h"
#include "s3_split.h"

using namespace bdm;

// ---------------------------------------------------------------------------------------------------
// pre-pass
// ---------------------------------------------------------------------------------------------------
// q or k: (C, L) fp32 rows of stride ld  ->  records [c8][split][l]
__global__ void attn_split_qk_kernel(int C, int L, const float *__restrict__ q, const float *__restrict__ k, long long bs,
                                     int ld, unsigned short *__restrict__ qs, unsigned short *__restrict__ ks) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x, c8 = blockIdx.y, C8 = gridDim.y, bi = blockIdx.z >> 1;
  if (l >= L) return;
  const float *src = ((blockIdx.z & 1) ? k : q) + (size_t)bi * bs;
  unsigned short *dst = ((blockIdx.z & 1) ? ks : qs) + ((size_t)bi * C8 + c8) * 3 * (size_t)L * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c8 * 8 + j;
    v[j] = c < C ? src[(size_t)c * ld + l] : 0.f;
  }
  store_s3(dst, (size_t)l, (size_t)L, v);
}

// v: (C, L) fp32 -> planes [split][c][Lp] bf16, zero for c >= C and l >= L
__global__ void attn_split_v_kernel(int C, int CP, int L, int Lp, const float *__restrict__ v, long long bs, int ld,
                                    unsigned short *__restrict__ vt) {
  const int l0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8, c = blockIdx.y, bi = blockIdx.z;
  if (l0 >= Lp) return;
  unsigned short h[8], m[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = (c < C && l0 + j < L) ? v[(size_t)bi * bs + (size_t)c * ld + l0 + j] : 0.f;
    split3(x, h[j], m[j], lo[j]);
  }
  uint4 ph, pm, pl;
  ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
  pm.x = m[0] | (m[1] << 16); pm.y = m[2] | (m[3] << 16); pm.z = m[4] | (m[5] << 16); pm.w = m[6] | (m[7] << 16);
  pl.x = lo[0] | (lo[1] << 16); pl.y = lo[2] | (lo[3] << 16); pl.z = lo[4] | (lo[5] << 16); pl.w = lo[6] | (lo[7] << 16);
  unsigned short *base = vt + (size_t)bi * 3 * CP * Lp;
  *reinterpret_cast<uint4 *>(base + ((size_t)0 * CP + c) * Lp + l0) = ph;
  *reinterpret_cast<uint4 *>(base + ((size_t)1 * CP + c) * Lp + l0) = pm;
  *reinterpret_cast<uint4 *>(base + ((size_t)2 * CP + c) * Lp + l0) = pl;
}

// ---------------------------------------------------------------------------------------------------
// flash kernel: one wave = 32 queries; the 4 waves of a workgroup share the K / V tiles of a 32-key step
// ---------------------------------------------------------------------------------------------------
#define VROW 36  // bf16 per V row in LDS: 32 keys + 4 pad (72-byte pitch keeps the 8-byte operand reads spread over the banks)

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 u) { return __builtin_bit_cast(bf16x8, u); }

// e^x for x <= 0 as one v_exp_f32 (2^t, 1 ulp) of t = x * log2(e).  The product's rounding perturbs t by |t| * 2^-24,
// i.e. a relative error |x| * 6e-8 on a weight e^x that is itself <= e^-|x|: below fp32 resolution of the
// normalised sum for every key that contributes.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// Exact-to-2^-24 split of a pair of probabilities into bf16 (hi, mid, lo) pairs: the three terms are the upper
// halves of x, x - hi, x - hi - mid (truncation; the remainders are exact), picked straight out of the fp32 bit
// patterns by one v_perm_b32 per pair.
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // {u1.hi16, u0.hi16}
  const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  const float t0 = r0 - __uint_as_float(v0 & 0xFFFF0000u), t1 = r1 - __uint_as_float(v1 & 0xFFFF0000u);
  l = __builtin_amdgcn_perm(__float_as_uint(t1), __float_as_uint(t0), 0x07060302u);
}

template <int CB>  // channel blocks of 32 (C <= 32 * CB)
__global__ __launch_bounds__(256) void attn_flash_s3_kernel(int C, int L, int Lp, const uint4 *__restrict__ qs,
                                                            const uint4 *__restrict__ ks,
                                                            const unsigned short *__restrict__ vt,
                                                            float *__restrict__ out, long long bs_o, int ld_o) {
  constexpr int C8 = 4 * CB, CP = 32 * CB, KT = C8 * 3 * 32, VT = 3 * CP * 4;  // uint4 items per K / V tile
  constexpr int KI = (KT + 255) / 256, VI = (VT + 255) / 256;
  __shared__ uint4 Ksh[KT];                                   // [c8][split][key]
  __shared__ __align__(16) unsigned short Vsh[3 * CP * VROW];  // [split][c][VROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int bi = blockIdx.y;
  const int i0 = (blockIdx.x * 4 + wave) * 32;  // this wave's first query
  const uint4 *qb = qs + (size_t)bi * C8 * 3 * L, *kb = ks + (size_t)bi * C8 * 3 * L;
  const unsigned short *vb = vt + (size_t)bi * 3 * CP * Lp;
  const uint4 zero = make_uint4(0u, 0u, 0u, 0u);

  // Q as B operand of k16-step s: channels 16s + 8lh .. +7 of query i0 + li, three splits
  bf16x8 qreg[2 * CB][3];
#pragma unroll
  for (int s = 0; s < 2 * CB; ++s)
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      const bool ok = i0 + li < L;
      const uint4 u = qb[ok ? ((size_t)(2 * s + lh) * 3 + sp) * L + i0 + li : 0];
      qreg[s][sp] = as_bf16x8(ok ? u : zero);
    }
  f32x16 o[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[cb][r] = 0.f;
  float run_max = -INFINITY, run_sum = 0.f;

  // register-prefetched tile staging: the loads of step j0 + 32 are in flight during the MFMAs of step j0
  // native-vector staging registers; loads without branch or select on the loaded value (either makes the wave wait for
  // memory inside the load phase): keys beyond L read a clamped address -- their scores are overwritten with -inf below,
  // so their probabilities are exactly 0 whatever was read -- and pieces beyond the tile are never stored
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v kr[KI], vr[VI];
  auto load_tile = [&](int j0) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int e = tid + i * 256, key = e & 31, cs = min(e >> 5, KT / 32 - 1);
      kr[i] = *reinterpret_cast<const u32x4v *>(&kb[(size_t)cs * L + min(j0 + key, L - 1)]);
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int e = tid + i * 256, piece = e & 3, rowi = min(e >> 2, VT / 4 - 1);
      vr[i] = *reinterpret_cast<const u32x4v *>(vb + ((size_t)rowi * Lp + min(j0 + piece * 8, Lp - 8)));
    }
  };
  load_tile(0);
  for (int j0 = 0; j0 < L; j0 += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int e = tid + i * 256;
      if (e < KT) *reinterpret_cast<u32x4v *>(&Ksh[e]) = kr[i];
    }
#pragma unroll
    for (int i = 0; i < VI; ++i) {
      const int e = tid + i * 256, piece = e & 3, rowi = e >> 2;
      if (e < VT) {
        uint2 *d = reinterpret_cast<uint2 *>(Vsh + rowi * VROW + piece * 8);
        d[0] = make_uint2(vr[i].x, vr[i].y);
        d[1] = make_uint2(vr[i].z, vr[i].w);
      }
    }
    __syncthreads();
    if (j0 + 32 < L) load_tile(j0 + 32);

    // S^T[j][i] = sum_c k[c][j] q[c][i]
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2 * CB; ++s) {
      bf16x8 a[3];
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) a[sp] = as_bf16x8(Ksh[((2 * s + lh) * 3 + sp) * 32 + li]);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], qreg[s][1], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], qreg[s][0], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], qreg[s][2], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], qreg[s][0], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], qreg[s][1], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], qreg[s][0], st, 0, 0, 0);
    }
    // online softmax over the keys (accumulator rows); keys beyond L do not exist
    if (j0 + 32 > L) {  // only the last tile can hold keys beyond L (uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (j0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= L) st[r] = -INFINITY;
    }
    float tile_max = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tile_max = fmaxf(tile_max, st[r]);
    tile_max = fmaxf(tile_max, __shfl_xor(tile_max, 32, 64));
    const float new_max = fmaxf(run_max, tile_max);
    const float corr = fast_exp(run_max - new_max);  // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = fast_exp(st[r] - new_max);
      psum += st[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    run_sum = run_sum * corr + psum;
    run_max = new_max;

    // P as B operand: registers 8jj .. 8jj+7 -> the 8 k-slots of MFMA jj, split into bf16 triples
    bf16x8 pb[2][3];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      unsigned h[4], m[4], lo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) split3_pair(st[8 * jj + 2 * i], st[8 * jj + 2 * i + 1], h[i], m[i], lo[i]);
      pb[jj][0] = as_bf16x8(make_uint4(h[0], h[1], h[2], h[3]));
      pb[jj][1] = as_bf16x8(make_uint4(m[0], m[1], m[2], m[3]));
      pb[jj][2] = as_bf16x8(make_uint4(lo[0], lo[1], lo[2], lo[3]));
    }
    const bool rescale = __any(corr != 1.0f);  // the running maximum moves in the first few tiles only
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      f32x16 acc = o[cb];
      if (rescale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= corr;
      }
      // O^T[c][i] += sum_j v[c][j] P[j][i] : A row = channel cb*32 + li, k-slots in the key order of P's registers
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        bf16x8 a[3];
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
          const unsigned short *rowp = Vsh + (sp * CP + cb * 32 + li) * VROW + 16 * jj + 4 * lh;
          const uint2 p0 = *reinterpret_cast<const uint2 *>(rowp), p1 = *reinterpret_cast<const uint2 *>(rowp + 8);
          a[sp] = as_bf16x8(make_uint4(p0.x, p0.y, p1.x, p1.y));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[jj][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], pb[jj][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[jj][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], pb[jj][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[jj][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], pb[jj][0], acc, 0, 0, 0);
      }
      o[cb] = acc;
    }
  }
  const float inv = 1.0f / run_sum;
  float *ob = out + (size_t)bi * bs_o;
  if (i0 + li < L) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (c < C) ob[(size_t)c * ld_o + i0 + li] = o[cb][r] * inv;
      }
  }
}

static inline int attn_cp(int c) { return c <= 32 ? 32 : 64; }
static inline int attn_lp(int l) { return (l + 7) / 8 * 8; }

extern "C" size_t bdm_attention_workspace_bytes(int b, int c, int l) {
  if (l <= 64 || c > 64) return 0;  // small / unsupported shapes run without a workspace
  const size_t cp = attn_cp(c), qk = (size_t)(cp / 8) * 3 * l * 16, v = 3 * cp * (size_t)attn_lp(l) * 2;
  return (size_t)b * (2 * qk + v) + 64;
}

// called by bdm_attention_core (dense_ops.hip) when a workspace is supplied
int attention_flash_s3(int b, int c, int l, const float *q, const float *k, const float *v, long long bs_qkv, int ld_qkv,
                       float *out, long long bs_o, int ld_o, void *workspace, hipStream_t s) {
  const int cp = attn_cp(c), c8 = cp / 8, lp = attn_lp(l);
  const size_t qk_elems = (size_t)b * c8 * 3 * l * 8;
  unsigned short *qs = reinterpret_cast<unsigned short *>((reinterpret_cast<size_t>(workspace) + 15) & ~(size_t)15);
  unsigned short *ks = qs + qk_elems, *vt = ks + qk_elems;
  hipLaunchKernelGGL(attn_split_qk_kernel, dim3(cdiv(l, 128), c8, 2 * b), dim3(128), 0, s, c, l, q, k, bs_qkv, ld_qkv, qs, ks);
  hipLaunchKernelGGL(attn_split_v_kernel, dim3(cdiv(lp / 8, 64), cp, b), dim3(64), 0, s, c, cp, l, lp, v, bs_qkv, ld_qkv, vt);
  dim3 grid(cdiv(l, 128), b);
  if (cp == 32)
    hipLaunchKernelGGL(attn_flash_s3_kernel<1>, grid, dim3(256), 0, s, c, l, lp, (const uint4 *)qs, (const uint4 *)ks, vt, out,
                       bs_o, ld_o);
  else
    hipLaunchKernelGGL(attn_flash_s3_kernel<2>, grid, dim3(256), 0, s, c, l, lp, (const uint4 *)qs, (const uint4 *)ks, vt, out,
                       bs_o, ld_o);
  return launch_status("attention_flash_s3");
}
