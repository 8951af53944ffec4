// sa_mlp_fused.hip -- the first set-abstraction module's grouped MLP with RECOMPUTE instead of store (round 4).
//
// pointnet.py:80-90 at level 0: group the 32 neighbours of every centre ([xyz - centre, features]: 3 + c rows), two shared-MLP layers
// [Conv2d k1 -> GroupNorm(8) -> Swish] (3 + c -> 32 -> 64), max over the neighbours.  As separate operators this wrote the grouped
// tensor (b, 35, 1024 x 32: 73 MB at B = 16), the first layer's output (67 MB) and the second's (134 MB), each read back once:
// ~200 us of mostly memory time for 3.3 GFLOP.  A GroupNorm needs the statistics of the whole (shape, group) before anything can be
// normalised, so the chain is evaluated in THREE passes, each of which re-gathers the neighbours' feature rows from a point-major
// copy (b, n, P) -- 8 MB, one 128-byte line per point at c = 32 -- and recomputes what it needs in registers:
//   pass 1   y1 = W1 x + b1                                   -> GroupNorm-1 slice partials only
//   pass 2   a1 = Swish(GN1(y1)), y2 = W2 a1 + b2             -> GroupNorm-2 slice partials only
//   pass 3   a2 = Swish(GN2(y2)), max over the 32 neighbours  -> out (b, 64, m)
// Nothing of size m x u is written.  Products on v_mfma_f32_32x32x2_f32 (exact fp32 products, as the generic 1x1 GEMM); a 32 x 32
// accumulator tile is one centre's neighbours.  Statistics: fp32 per lane over the wave's centres, butterfly, waves in order, one
// fp64 slice per workgroup; consumers add the slices in a fixed order: deterministic.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

// GroupNorm statistics are summed along ONE binary tree over the centres (padded with zeros to a power of two), whatever the launch's
// centres-per-wave: a centre's value is a fixed-order fp32 sum (registers, then lanes); centres -> wave -> workgroup (= slice) ->
// shape are pairwise fp64 sums in tree order.  A shape's statistics therefore do not depend on the batch it is sampled in (the
// centres-per-wave follow the batch size).  Here: the slices' part of the tree -- lane l of a group's 32 holds the aligned block
// [l P / 32, (l + 1) P / 32) of the P = 2^k >= S slices (in-lane tree over <= 8 values), then the lanes pair up (xor 1, 2, 4, 8, 16).
__device__ __forceinline__ void group_stats8(const double *__restrict__ partial, int bi, int G, int S, double count, float eps,
                                             float *s_mean, float *s_rstd) {
  const int tid = threadIdx.x, l = tid & 31, g = tid >> 5;
  int P = 32;
  while (P < S) P <<= 1;
  const int per = P >> 5;   // 1, 2, 4 or 8 (S <= 256)
  double va[8], vq[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sl = l * per + i;
    const bool ok = g < G && i < per && sl < S;
    const double *pp = partial + (((size_t)bi * G + (ok ? g : 0)) * S + (ok ? sl : 0)) * 2;
    va[i] = ok ? pp[0] : 0.0;
    vq[i] = ok ? pp[1] : 0.0;
  }
#pragma unroll
  for (int span = 1; span < 8; span <<= 1)
#pragma unroll
    for (int i = 0; i + span < 8; i += 2 * span) { va[i] += va[i + span]; vq[i] += vq[i + span]; }
  double a = va[0], q = vq[0];
  a = half32_sum(a); q = half32_sum(q);   // (bit-identical to the xor butterfly over the group's 32 lanes)
  if (l == 0 && g < G) {
    const double mu = a / count;
    double var = q / count - mu * mu;
    if (var < 0) var = 0;
    s_mean[g] = (float)mu;
    s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
}

}  // namespace

// PASS 1 / 2 / 3 as above.  m1 = 32, m2 = 64, u = 32, groups = 8.  No LDS tile and no barrier on the data path:
//   layer 1   C1[channel][neighbour] = W1 (A operand, registers) . X (B operand).  The k index of an MFMA step is only a NAME shared by
//             the two operands, so the half-waves split a neighbour's feature row between them: lane (li, lh) reads floats
//             [lh P/2, (lh + 1) P/2) of its neighbour's row with 16-byte loads (at c = 32: the two halves of one 128-byte line) and
//             step (q, i) multiplies feature lh P/2 + 4 q + i; two more steps take (dx, dz) and (dy, 0) from the coordinate planes;
//   layer 2   C1's layout leaves lane (li, lh) holding its neighbour's channels (r & 3) + 8 (r >> 2) + 4 lh, r < 16: again a partition of
//             k between the half-waves, so Swish(GN1(C1)) IS the A operand of C2[neighbour][channel] = a1 . W2^T (B operand, registers):
//             no transpose.  C2 has the channels on the lanes: the GroupNorm-2 constants are per lane, the max over neighbours is a max
//             over the lane's 16 registers + one exchange between the half-waves.
// A wave walks `tpw` consecutive centres, the next centre's rows in flight while the current one is multiplied; workgroup = 4 tpw
// centres; the (shape, centre block) items are dealt so that a shape's workgroups share an XCD (its rows stay in that XCD's L2).
constexpr int SA_Q = 4;      // 16-byte loads per lane and centre: rows of 2 * 4 * SA_Q = 32 features (zero-padded)
constexpr int SA_P = 8 * SA_Q;
constexpr int SA_TPW = 8;    // most centres per wave

struct SaTile {
  float4 f[SA_Q];            // this lane's half of its neighbour's feature row
  float px, py, qx, qy;      // lh = 0: point (x, y), centre (x, y); lh = 1: point (z, y), centre (z, y) -- the y pair unused
};

template <int PASS, int TPW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void sa_mlp2_kernel(int c, int n, int m, int nb, const float *__restrict__ rows,
                                                      const float *__restrict__ coords, const float *__restrict__ centers,
                                                      const int *__restrict__ idx, const float *__restrict__ w1,
                                                      const float *__restrict__ b1, const float *__restrict__ g1w,
                                                      const float *__restrict__ g1b, float eps1, const float *__restrict__ w2,
                                                      const float *__restrict__ b2, const float *__restrict__ g2w,
                                                      const float *__restrict__ g2b, float eps2, double *__restrict__ partial1,
                                                      double *__restrict__ partial2, int S, float *__restrict__ out, long long bs_o,
                                                      int ld_o) {
  __shared__ float s_mean[8], s_rstd[8];
  __shared__ float s_a1[32], s_c1[32];
  __shared__ double s_red[4][8][2];
  __shared__ float s_out[PASS == 3 ? 64 : 1][4 * TPW + 1];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int U = 32, G = 8;
  // XCD-aware item order: workgroup ids are dealt round-robin to the 8 XCDs; XCD x takes the contiguous items [x * per, (x + 1) * per)
  const int total = S * nb, per = (total + 7) >> 3;
  const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (item >= total) return;
  const int bi = item / S, wg = item - bi * S;
  constexpr int tpw = TPW, p = SA_P, half = SA_P / 2;
  const int j0 = (wg * 4 + wave) * tpw;            // this wave's first centre
  const int k1 = 3 + c, mu_total = m * U;
  const float *rows_b = rows + (size_t)bi * n * p + half * lh;
  const float *xyz = coords + (size_t)bi * 3 * n + (size_t)(2 * lh) * n, *cen = centers + (size_t)bi * 3 * m + (size_t)(2 * lh) * m;
  const float *xyz_y = coords + (size_t)bi * 3 * n + n, *cen_y = centers + (size_t)bi * 3 * m + m;
  const int *idx_b = idx + (size_t)bi * m * U + li;

  // neighbour indices of all the wave's centres first: the row loads below then have ONE memory round trip in front of them
  int src[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) src[t] = idx_b[(size_t)min(j0 + t, m - 1) * U];
  auto fetch = [&](SaTile &x, int t) {
    const int jj = min(j0 + t, m - 1), sp = src[t];
    const float4 *row = reinterpret_cast<const float4 *>(rows_b + (size_t)sp * p);
#pragma unroll
    for (int q = 0; q < SA_Q; ++q) x.f[q] = row[q];
    x.px = xyz[sp]; x.py = xyz_y[sp]; x.qx = cen[jj]; x.qy = cen_y[jj];
  };
  SaTile x[3];   // ring: two centres' rows in flight behind the one being multiplied
  fetch(x[0], 0);
  if (TPW > 1) fetch(x[1], 1);

  // layer-1 weights as the A operand (row = channel li), in the k order of the loads above
  float wa[SA_Q][4], wc0, wc1;
  {
    const float *wr = w1 + (size_t)li * k1;
    wc0 = lh == 0 ? wr[0] : wr[2];
    wc1 = lh == 0 ? wr[1] : 0.f;
#pragma unroll
    for (int q = 0; q < SA_Q; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = half * lh + 4 * q + i;
        wa[q][i] = f < c ? wr[3 + f] : 0.f;
      }
  }

  if (PASS >= 2) {   // GroupNorm-1 as an affine form per channel
    group_stats8(partial1, bi, G, S, 4.0 * mu_total, eps1, s_mean, s_rstd);
    if (tid < 32) {
      const float ga = g1w[tid] * s_rstd[tid >> 2];
      s_a1[tid] = ga;
      s_c1[tid] = __builtin_fmaf(b1[tid] - s_mean[tid >> 2], ga, g1b[tid]);   // bias folded: GN1(y + b) = ga y + (b - mean) ga + beta
    }
    __syncthreads();
  }
  float a2[2] = {0.f, 0.f}, c2[2] = {0.f, 0.f};
  if (PASS == 3) {   // GroupNorm-2: this lane's two channels (li, li + 32)
    group_stats8(partial2, bi, G, S, 8.0 * mu_total, eps2, s_mean, s_rstd);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int ch = mt * 32 + li;
      a2[mt] = g2w[ch] * s_rstd[ch >> 3];
      c2[mt] = __builtin_fmaf(b2[ch] - s_mean[ch >> 3], a2[mt], g2b[ch]);
    }
  }
  // per register r of a layer-1 tile: channel kr = (r & 3) + 8 (r >> 2) + 4 lh
  float k_a[16], k_b[16];          // passes 2, 3: GroupNorm-1 (scale, shift)
  float wb[PASS >= 2 ? 2 : 1][16];  // layer-2 weights as the B operand: column = channel li + 32 mt, k = kr
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int ch = (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (PASS == 1) { k_a[r] = 0.f; k_b[r] = 0.f; }
    else { k_a[r] = s_a1[ch]; k_b[r] = s_c1[ch]; }
  }
  if (PASS >= 2) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float4 v = *reinterpret_cast<const float4 *>(w2 + (size_t)(mt * 32 + li) * 32 + 8 * qd + 4 * lh);
        wb[mt][4 * qd] = v.x; wb[mt][4 * qd + 1] = v.y; wb[mt][4 * qd + 2] = v.z; wb[mt][4 * qd + 3] = v.w;
      }
  }
  const float bias2[2] = {PASS == 2 ? b2[li] : 0.f, PASS == 2 ? b2[32 + li] : 0.f};

  // per centre of this wave: the statistics of ITS 32 columns (pass 1: group li / 4 in every lane; pass 2: groups li / 8 and 4 + li / 8)
  // (each centre's value is an fp32 sum in a fixed order; fp64 from there on.)  The tree is built as the centres arrive: lvl[k] holds
  // the finished LEFT subtree of 2^k centres until its right sibling is complete (a binary counter; t is a compile-time constant).
  constexpr int NLVL = TPW >= 8 ? 4 : (TPW >= 4 ? 3 : (TPW >= 2 ? 2 : 1));
  double lvl_s[NLVL][2], lvl_q[NLVL][2];
  auto tree_push = [&](int t, int v, float sv, float qv) {
    double a = (double)sv, q = (double)qv;
#pragma unroll
    for (int k = 0; k < NLVL - 1; ++k)
      if ((t >> k) & 1) { a = lvl_s[k][v] + a; q = lvl_q[k][v] + q; }
      else { lvl_s[k][v] = a; lvl_q[k][v] = q; return; }
    lvl_s[NLVL - 1][v] = a; lvl_q[NLVL - 1][v] = q;
  };
  const float bias1 = PASS == 1 ? b1[li] : 0.f;

  auto compute = [&](const SaTile &x, int t) {
    const int j = j0 + t;
    const bool live = j < m;
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    if (PASS == 1) {
      // statistics only: the operands SWAPPED, C1^T[neighbour][channel] -- a channel's 32 neighbours are then 16 registers of two lanes
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.px - x.qx, wc0, acc1, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.py - x.qy, wc1, acc1, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < SA_Q; ++q) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.f[q].x, wa[q][0], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.f[q].y, wa[q][1], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.f[q].z, wa[q][2], acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.f[q].w, wa[q][3], acc1, 0, 0, 0);
      }
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float v = acc1[r] + bias1; sv += v; qv = __builtin_fmaf(v, v, qv); }
      sv += __shfl_xor(sv, 32, 64); qv += __shfl_xor(qv, 32, 64);   // the other 16 neighbours
#pragma unroll
      for (int o = 1; o < 4; o <<= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }   // the group's 4 channels
      tree_push(t, 0, live ? sv : 0.f, live ? qv : 0.f);
      return;
    }
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc0, x.px - x.qx, acc1, 0, 0, 0);   // lh = 0: dx, lh = 1: dz
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc1, x.py - x.qy, acc1, 0, 0, 0);   // lh = 0: dy, lh = 1: weight 0
#pragma unroll
    for (int q = 0; q < SA_Q; ++q) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q][0], x.f[q].x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q][1], x.f[q].y, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q][2], x.f[q].z, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q][3], x.f[q].w, acc1, 0, 0, 0);
      }
    // C layout: register r of lane (li, lh) = channel (r & 3) + 8 (r >> 2) + 4 lh, neighbour li
    // a1 = Swish(GN1(y1)) in place: the A operand of layer 2
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = swishf(__builtin_fmaf(acc1[r], k_a[r], k_b[r]));
    f32x16 acc2[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc1[r], wb[PASS >= 2 ? mt : 0][r], acc2[mt], 0, 0, 0);
    // C2 layout: register r of lane (li, lh) = neighbour (r & 3) + 8 (r >> 2) + 4 lh, channel li + 32 mt
    if (PASS == 2) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float sv = 0.f, qv = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc2[mt][r] + bias2[mt]; sv += v; qv = __builtin_fmaf(v, v, qv); }
        sv += __shfl_xor(sv, 32, 64); qv += __shfl_xor(qv, 32, 64);   // the other 16 neighbours
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }   // the group's 8 channels
        tree_push(t, mt, live ? sv : 0.f, live ? qv : 0.f);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        float v = swishf(__builtin_fmaf(acc2[mt][0], a2[mt], c2[mt]));
#pragma unroll
        for (int r = 1; r < 16; ++r) v = fmaxf(v, swishf(__builtin_fmaf(acc2[mt][r], a2[mt], c2[mt])));
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        if (lh == 0) s_out[PASS == 3 ? mt * 32 + li : 0][wave * tpw + t] = v;
      }
    }
  };

#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    if (t + 2 < TPW) fetch(x[(t + 2) % 3], t + 2);
    compute(x[t % 3], t);
  }

  if (PASS == 3) {
    // the workgroup's 4 tpw consecutive centres, 64 channels: coalesced rows of the output
    __syncthreads();
    const int cw = 4 * tpw, jb = wg * cw;
    for (int e = tid; e < 64 * cw; e += 256) {
      const int ch = e / cw, jj = e - ch * cw;
      if (jb + jj < m) out[(size_t)bi * bs_o + (size_t)ch * ld_o + jb + jj] = s_out[PASS == 3 ? ch : 0][jj];
    }
    return;
  }
  // centres -> wave (tree over TPW), waves -> workgroup (tree over 4): one fp64 slice per workgroup
  if (PASS != 3) {
    constexpr int NV = PASS == 1 ? 1 : 2;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double sa = lvl_s[NLVL - 1][v], sq = lvl_q[NLVL - 1][v];
      if (PASS == 1) {
        if (lh == 0 && (li & 3) == 0) { s_red[wave][li >> 2][0] = sa; s_red[wave][li >> 2][1] = sq; }
      } else {
        if (lh == 0 && (li & 7) == 0) { s_red[wave][v * 4 + (li >> 3)][0] = sa; s_red[wave][v * 4 + (li >> 3)][1] = sq; }
      }
    }
    __syncthreads();
    if (tid < G) {
      const double a = (s_red[0][tid][0] + s_red[1][tid][0]) + (s_red[2][tid][0] + s_red[3][tid][0]);
      const double q = (s_red[0][tid][1] + s_red[1][tid][1]) + (s_red[2][tid][1] + s_red[3][tid][1]);
      double *dst = (PASS == 1 ? partial1 : partial2) + (((size_t)bi * G + tid) * S + wg) * 2;
      dst[0] = a; dst[1] = q;
    }
  }
}

// features (b, c, n) -> point-major rows (b, n, P = 32), zero-padded: tile transpose through LDS
__global__ __launch_bounds__(256) void sa_rows_kernel(int c, int n, int p, const float *__restrict__ feat, long long bs_f, int ld_f,
                                                      float *__restrict__ rows) {
  __shared__ float tile[32][65];
  const int bi = blockIdx.z, n0 = blockIdx.x * 64, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = c0 + ty * 8 + i, pt = n0 + tx;
    tile[ty * 8 + i][tx] = (ch < c && pt < n) ? feat[(size_t)bi * bs_f + (size_t)ch * ld_f + pt] : 0.f;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int pt = n0 + py * 8 + i;
    if (pt < n && c0 + cx < p) rows[((size_t)bi * n + pt) * p + c0 + cx] = tile[cx][py * 8 + i];
  }
}

static int sa_fused_tpw(int b, int m) {
  // centres per wave: as many as keep >= 2 workgroups per CU (the prefetch pipeline wants a few), at most SA_TPW
  int tpw = SA_TPW;
  while (tpw > 1 && (long long)b * ((m + 4 * tpw - 1) / (4 * tpw)) < 512) tpw >>= 1;
  return tpw;
}

extern "C" size_t bdm_sa_mlp2_fused_rows_bytes(int b, int c, int n) { return (size_t)b * n * SA_P * sizeof(float); }

extern "C" int bdm_sa_mlp2_fused_slices(int b, int m) {
  const int tpw = sa_fused_tpw(b, m);
  return (m + 4 * tpw - 1) / (4 * tpw);
}

extern "C" int bdm_sa_mlp2_fused(int b, int c, int n, int m, int u, int m1, int m2, const float *coords, const float *features,
                                 long long bs_f, int ld_f, const float *centers, const int *indices, const float *w1, const float *b1,
                                 const float *g1w, const float *g1b, float eps1, const float *w2, const float *b2, const float *g2w,
                                 const float *g2b, float eps2, int groups, void *rows, void *partial1, void *partial2, float *out,
                                 long long bs_o, int ld_o, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 1 && c <= 8 * SA_Q && n >= 1 && m >= 1 && u == 32 && m1 == 32 && m2 == 64 && groups == 8 && coords &&
                  features && centers && indices && w1 && b1 && g1w && g1b && w2 && b2 && g2w && g2b && rows && partial1 && partial2 && out,
              "sa_mlp2_fused: supports 1 <= c <= %d features (+ 3 coordinates) -> 32 -> 64 channels, 32 neighbours, GroupNorm(8) "
              "(got c=%d u=%d m1=%d m2=%d groups=%d)", 8 * SA_Q, c, u, m1, m2, groups);
  BDM_REQUIRE(((reinterpret_cast<size_t>(rows) | reinterpret_cast<size_t>(w2)) & 15) == 0, "sa_mlp2_fused: rows and w2 must be 16-byte aligned");
  if (b == 0) return BDM_OK;
  const int p = SA_P, tpw = sa_fused_tpw(b, m);
  const int S = (m + 4 * tpw - 1) / (4 * tpw);
  const long long total = (long long)S * b;
  BDM_REQUIRE(total < (1ll << 28) && S <= 256, "sa_mlp2_fused: at most 8192 centres per shape (got m=%d)", m);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sa_rows_kernel, dim3(cdiv(n, 64), cdiv(p, 32), b), dim3(256), 0, s, c, n, p, features, bs_f, ld_f, (float *)rows);
  dim3 grid((unsigned)(8 * ((total + 7) / 8)));
#define SA_PASS(P, T)                                                                                                              \
  hipLaunchKernelGGL((sa_mlp2_kernel<P, T>), grid, dim3(256), 0, s, c, n, m, b, (const float *)rows, coords, centers, indices, w1, b1, \
                     g1w, g1b, eps1, w2, b2, g2w, g2b, eps2, (double *)partial1, (double *)partial2, S, out, bs_o, ld_o)
#define SA_PASSES(T) \
  do {               \
    SA_PASS(1, T);   \
    SA_PASS(2, T);   \
    SA_PASS(3, T);   \
  } while (0)
  if (tpw == 8) SA_PASSES(8);
  else if (tpw == 4) SA_PASSES(4);
  else if (tpw == 2) SA_PASSES(2);
  else SA_PASSES(1);
#undef SA_PASSES
#undef SA_PASS
  return launch_status("sa_mlp2_fused");
}
