// EXPERIMENTAL family (`make EXPERIMENTAL=1`): built, parity-tested, measured and NOT faster than the operator chain --
// FP0 PVConv 167 -> 180 us, FP1 218 -> 220 us per module at B = 16 with both kernels; the step 5.40 ms either way.  Kept as the
// record of VERDICT r4 next-1 (a) + (c); the default library does not contain it.
//
// pvconv_small.hip -- the glue of a PVConv on the SMALL voxel grids (8^3: 64 - 256 points, 128 - 256 channels per shape), where a
// module is a chain of ~10 dependent launches of 5 - 20 us each and the dense second convolution is the only kernel that fills the
// chip.  Everything here is PER SHAPE: a workgroup owns (shape, channel slab) and never
// waits for another workgroup, so no device-scope hand-off is needed (the round-1/2 "last workgroup" forms paid an L2 write-back
// per release on the 8-XCD part).
//
//   bdm_pvconv_tail_small   SE gate (se.py:8-19: both FC layers, from the channel means bdm_se_gate_gn(w1 = NULL) left)
//                           + GroupNorm-2 + Swish + gate evaluated once per CELL into LDS + trilinear devoxelisation of the slab
//                           (trilinear_devox.cu:37-104) + the point branch's GroupNorm + Swish + add  (pvconv.py:91-97)
//                           [+ optionally the HEAD of the next PVConv on the same voxel plan: mean of the fused features per
//                           occupied cell (vox.cu:18-72 on the plan's ordered lists) and their two-term fp16 split, i.e. the A
//                           operand of that PVConv's sparse GEMM -- the feature tensor is voxelised by the workgroup that just
//                           produced it].  Replaces se_fc + devox_gn_fused (+ sparse_vox_features + sparse_split_h2).  (A one-workgroup-
//                           per-shape form that also computed the channel means -- no row-mean launch -- was built and measured at
//                           57 - 126 us per launch against 9 + 25 for row means + this kernel: one CU streaming a shape's 512 KB grid
//                           twice through eight slab barriers is slower than 512 workgroups; removed)
//   bdm_sparse_conv_gather_h2_small
//                           the sparse first convolution's gather (sparse_conv.hip) with GroupNorm-1 + Swish + the fp16 operand
//                           split of the second convolution in its epilogue: a workgroup owns (shape, GroupNorm group) over the
//                           whole grid, so the statistics never leave it.  Replaces gather + to_h2_stats; the dense fp32 output of
//                           the first convolution is not written.
//
// Arithmetic: the same expressions, in the same order, as the kernels they replace wherever a value is shared with them (gate,
// devoxelised sums, per-cell means), so the two-launch forms stay usable as bit-exact references in the tests.
#include "../../../include/bdm_hip.h"
#include "../common.h"
#include "../se_fc.h"

using namespace bdm;

#include "../sparse_h2_common.h"

namespace {

constexpr int TAIL_CS = 8;        // channels per workgroup = one fp16 record group of the next convolution's operand
constexpr int TAIL_T = 256;

struct TailHead {                 // the next PVConv's first-convolution operand (all NULL / 0: no head)
  const int *cnt, *start, *sorted, *occ_list, *n_occ;
  int n_max;
  float x_scale;                  // power of two; the GEMM divides by it through amax_out
  uint4 *xh;                      // (b, c/8, 2, n_max) records of 8 fp16
  float *amax_out;                // (b): written so that act_scale_from_max(amax_out[b]) == x_scale
  int *saturated;                 // OR-ed with 1 when a scaled value leaves fp16's range
};

__global__ __launch_bounds__(TAIL_T) void pv_tail_small_kernel(int c, int n, int r, int hidden, const float *__restrict__ coords,
                                                               const float *__restrict__ grid, const float2 *__restrict__ coef,
                                                               const float *__restrict__ se_mean, const float *__restrict__ w1,
                                                               const float *__restrict__ w2, const float *__restrict__ add,
                                                               long long bs_a, int ld_a, const float2 *__restrict__ add_coef,
                                                               float *__restrict__ out, long long bs_o, int ld_o, TailHead hd) {
  extern __shared__ __align__(16) float smem[];
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x;
  const int slabs = c / TAIL_CS, bi = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * TAIL_CS;
  float *vals = smem;                       // [CS][r3]  swish(a g + b) * gate per cell
  float *fs = vals + TAIL_CS * r3;          // [CS][n]   the fused features of the slab (head only)
  float *s_mean = fs + (hd.xh ? TAIL_CS * n : 0);   // [c]
  float *s_hid = s_mean + c;                // [hidden]
  float *s_gate = s_hid + hidden;           // [CS]
  // ---- SE gate of the slab's channels: the two FC layers as every other kernel evaluates them (se_fc.h) -------------------------------
  for (int i = tid; i < c; i += TAIL_T) s_mean[i] = se_mean[(size_t)bi * c + i];
  __syncthreads();
  se_hidden_layer(c, hidden, w1, s_mean, s_hid);
  __syncthreads();
  if (tid < TAIL_CS) s_gate[tid] = se_gate_of(c0 + tid, hidden, w2, s_hid);
  __syncthreads();
  {
#pragma clang fp contract(off)
    // ---- per cell: GroupNorm-2 + Swish + gate, once (devox_gn_lds_kernel's expression) ------------------------------------------
    for (int cl = 0; cl < TAIL_CS; ++cl) {
      const int ci = c0 + cl;
      const float2 ab = coef[(size_t)bi * c + ci];
      const float s = s_gate[cl];
      const float4 *g4 = reinterpret_cast<const float4 *>(grid + ((size_t)bi * c + ci) * r3);
      float4 *v4 = reinterpret_cast<float4 *>(vals + (size_t)cl * r3);
      for (int e = tid; e < r3 / 4; e += TAIL_T) {
        const float4 g = g4[e];
        v4[e] = make_float4(swishf(g.x * ab.x + ab.y) * s, swishf(g.y * ab.x + ab.y) * s, swishf(g.z * ab.x + ab.y) * s,
                            swishf(g.w * ab.x + ab.y) * s);
      }
    }
  }
  __syncthreads();
  {
#pragma clang fp contract(off)
    // ---- per (point, channel): 8 corners from LDS + the point branch ------------------------------------------------------------
    const float *pc = coords + (size_t)bi * 3 * n;
    for (int item = tid; item < n * TAIL_CS; item += TAIL_T) {
      const int cl = item / n, i = item - cl * n, ci = c0 + cl;
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
      const float *v = vals + (size_t)cl * r3;
      float cv[8] = {v[i000], v[i001], v[i010], v[i011], v[i100], v[i101], v[i110], v[i111]};
      lds_settle8(cv);
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
      if (hd.xh) fs[cl * n + i] = acc;
    }
  }
  if (!hd.xh) return;
  __syncthreads();
  // ---- head of the next PVConv: per occupied cell the mean of the slab's features over its points (ascending point index: the
  // arithmetic of sparse_vox_features_lds_kernel, i.e. of the dense voxel grid), scaled by the power of two x_scale and split
  // into (hi, lo) fp16 -- one record of 8 channels per cell ----------------------------------------------------------------------
  {
#pragma clang fp contract(off)
    const int g = c0 / 8, G = c / 8;
    const int nocc = min(hd.n_occ[bi], hd.n_max);
    int sat = 0;
    for (int k = tid; k < hd.n_max; k += TAIL_T) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      if (k < nocc) {
        const int v = hd.occ_list[(size_t)bi * hd.n_max + k];
        const int cv = hd.cnt[(size_t)bi * r3 + v];
        const int *so = hd.sorted + (size_t)bi * n + hd.start[(size_t)bi * r3 + v];
        const float inv = cv > 0 ? (float)(1.0 / (double)(float)cv) : 0.f;
        for (int q = 0; q < cv; ++q) {
          const int p = so[q];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = acc[j] + fs[j * n + p] * inv;
        }
      }
      f16x8 hi, lo;
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(acc[j] * hd.x_scale));
      sat |= m > 65504.f;
      split_record(make_float4(acc[0], acc[1], acc[2], acc[3]), make_float4(acc[4], acc[5], acc[6], acc[7]), hd.x_scale, hi, lo);
      hd.xh[(((size_t)bi * G + g) * 2 + 0) * hd.n_max + k] = *reinterpret_cast<const uint4 *>(&hi);
      hd.xh[(((size_t)bi * G + g) * 2 + 1) * hd.n_max + k] = *reinterpret_cast<const uint4 *>(&lo);
    }
    if (sat && hd.saturated) atomicOr(hd.saturated, 1);
    if (c0 == 0 && tid == 0) hd.amax_out[bi] = 24576.0f / hd.x_scale;  // 0.75 * 2^15 / s: act_scale_from_max() of it is s
  }
}


// The same kernel for the sizes of the 8^3 levels (c <= 256, hidden <= 32, n <= 256 points, n_max <= 256 rows) with EVERY global load that
// does not depend on the gate issued up front: the means, W1's slices (a wave owns hidden units w, w + 4, ..: 32 registers), W2's 8 rows,
// the slab's cells, the point's coordinates, the point branch's values and, for the head, the cell's point list.  The generic kernel
// above walks ~17 dependent global round trips (means -> W1 -> W2 -> cells -> per item: coordinates + point branch -> per row: cell ->
// count / start -> list), ~25 us for 4096 cells; here it is one trip plus LDS work.  Same expressions in the same order: same bits.
__global__ __launch_bounds__(TAIL_T) void pv_tail_small_fast_kernel(int c, int n, int r, int hidden, const float *__restrict__ coords,
                                                                    const float *__restrict__ grid, const float2 *__restrict__ coef,
                                                                    const float *__restrict__ se_mean, const float *__restrict__ w1,
                                                                    const float *__restrict__ w2, const float *__restrict__ add,
                                                                    long long bs_a, int ld_a, const float2 *__restrict__ add_coef,
                                                                    float *__restrict__ out, long long bs_o, int ld_o, TailHead hd) {
  extern __shared__ __align__(16) float smem[];
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slabs = c / TAIL_CS, bi = blockIdx.x / slabs, c0 = (blockIdx.x % slabs) * TAIL_CS;
  float *vals = smem;                       // [CS][r3]
  float *fs = vals + TAIL_CS * r3;          // [CS][n] (head only)
  float *s_mean = fs + (hd.xh ? TAIL_CS * n : 0);
  float *s_hid = s_mean + c;                // [32]
  float *s_gate = s_hid + 32;               // [CS]
  float *s_w2 = s_gate + TAIL_CS;           // [CS][32]
  // ---- every load that does not wait for the gate -----------------------------------------------------------------------------------
  const float m_reg = tid < c ? se_mean[(size_t)bi * c + tid] : 0.f;
  float w1r[8][4];                          // W1[j][k]: j = wave + 4 u, k = lane + 64 v (clamped addresses, zeroed beyond the matrix)
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int j = wave + 4 * u, k = lane + 64 * v;
      const float w = w1[(size_t)min(j, hidden - 1) * c + min(k, c - 1)];
      w1r[u][v] = (j < hidden && k < c) ? w : 0.f;
    }
  const int g_ch = tid >> 5, g_k = tid & 31;  // gate: 32 lanes per channel of the slab
  const float w2r = g_k < hidden ? w2[(size_t)(c0 + g_ch) * hidden + g_k] : 0.f;
  const int q4 = r3 >> 2;
  float4 cell[4];
  float2 cab[4];
  int ccl[4], ce[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {          // cells: e = tid + 256 it over CS * q4 (4096 cells at 8^3: four float4 per thread)
    const int e = tid + TAIL_T * it, cl = min(e / q4, TAIL_CS - 1);
    ccl[it] = cl; ce[it] = e - cl * q4;
    const bool ok = e < TAIL_CS * q4;
    cell[it] = reinterpret_cast<const float4 *>(grid + ((size_t)bi * c + c0 + cl) * r3)[ok ? ce[it] : 0];
    cab[it] = coef[(size_t)bi * c + c0 + cl];
  }
  // points: slot pi of PS (a power of two >= n), channel lane cln of CL = 256 / PS; the thread's channels: cln, cln + CL, ...
  const int PS = n > 128 ? 256 : (n > 64 ? 128 : 64), CL = TAIL_T / PS, pi = tid & (PS - 1), cln = tid / PS;
  const bool pok = pi < n;
  const float *pc = coords + (size_t)bi * 3 * n;
  const float px = pc[pok ? pi : 0], py = pc[n + (pok ? pi : 0)], pz = pc[2 * n + (pok ? pi : 0)];
  float av[8];
  float2 apf[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int cl = cln + u * CL;
    av[u] = 0.f; apf[u] = make_float2(0.f, 0.f);
    if (cl < TAIL_CS) {
      if (add) av[u] = add[(size_t)bi * bs_a + (size_t)(c0 + cl) * ld_a + (pok ? pi : 0)];
      if (add_coef) apf[u] = add_coef[(size_t)bi * c + c0 + cl];
    }
  }
  // head: the cell of row k = tid and the head of its point list (three dependent loads, in flight behind everything above)
  int h_cv = 0;
  const int *h_so = nullptr;
  int h_p[4] = {0, 0, 0, 0};
  if (hd.xh) {
    const int nocc = min(hd.n_occ[bi], hd.n_max);
    if (tid < nocc) {
      const int v = hd.occ_list[(size_t)bi * hd.n_max + tid];
      h_cv = hd.cnt[(size_t)bi * r3 + v];
      h_so = hd.sorted + (size_t)bi * n + hd.start[(size_t)bi * r3 + v];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) h_p[qq] = h_so[min(qq, h_cv - 1)];
    }
  }
  // ---- SE gate (se_fc.h's sums: lanes k = lane + 64 v ascending, se_wave_sum; gate: k ascending) ---------------------------------------
  if (tid < c) s_mean[tid] = m_reg;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int j = wave + 4 * u;
    if (j < hidden) {                       // wave-uniform
      float a = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (lane + 64 * v < c) a += w1r[u][v] * s_mean[lane + 64 * v];
      a = se_wave_sum(a);
      if (lane == 0) s_hid[j] = fmaxf(a, 0.f);
    }
  }
  s_w2[tid] = w2r;                          // [channel of the slab][k]: the eight rows of W2, read back below by one thread per channel
  __syncthreads();
  if (tid < TAIL_CS) {                      // se_gate_of's sum (k ascending) from LDS
    float a = 0.f;
    for (int k = 0; k < hidden; ++k) a += s_w2[tid * 32 + k] * s_hid[k];
    s_gate[tid] = 1.0f / (1.0f + expf(-a));
  }
  __syncthreads();
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (tid + TAIL_T * it < TAIL_CS * q4) {
        const float s = s_gate[ccl[it]];
        const float4 g = cell[it];
        const float2 ab = cab[it];
        reinterpret_cast<float4 *>(vals + (size_t)ccl[it] * r3)[ce[it]] =
            make_float4(swishf(g.x * ab.x + ab.y) * s, swishf(g.y * ab.x + ab.y) * s, swishf(g.z * ab.x + ab.y) * s, swishf(g.w * ab.x + ab.y) * s);
      }
    }
  }
  __syncthreads();
  {
#pragma clang fp contract(off)
    const float x = px, y = py, z = pz;
    const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
    const float x1 = x - xl, y1 = y - yl, z1 = z - zl;
    const float x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
    const float w000 = x0 * y0 * z0, w001 = x0 * y0 * z1, w010 = x0 * y1 * z0, w011 = x0 * y1 * z1,
                w100 = x1 * y0 * z0, w101 = x1 * y0 * z1, w110 = x1 * y1 * z0, w111 = x1 * y1 * z1;
    const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
    const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
    const int i001 = i000 + sz, i010 = i000 + sy, i011 = i010 + sz;
    const int i100 = i000 + sx, i101 = i100 + sz, i110 = i100 + sy, i111 = i110 + sz;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int cl = cln + u * CL;
      if (cl < TAIL_CS && pok) {
        const float *v = vals + (size_t)cl * r3;
        float cv[8] = {v[i000], v[i001], v[i010], v[i011], v[i100], v[i101], v[i110], v[i111]};
        lds_settle8(cv);
        float acc = w000 * cv[0];
        acc += w001 * cv[1];
        acc += w010 * cv[2];
        acc += w011 * cv[3];
        acc += w100 * cv[4];
        acc += w101 * cv[5];
        acc += w110 * cv[6];
        acc += w111 * cv[7];
        if (add) {
          float a2 = av[u];
          if (add_coef) a2 = swishf(a2 * apf[u].x + apf[u].y);
          acc += a2;
        }
        out[(size_t)bi * bs_o + (size_t)(c0 + cl) * ld_o + pi] = acc;
        if (hd.xh) fs[cl * n + pi] = acc;
      }
    }
  }
  if (!hd.xh) return;
  __syncthreads();
  {
#pragma clang fp contract(off)
    const int g = c0 / 8, G = c / 8;
    int sat = 0;
    if (tid < hd.n_max) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      const float inv = h_cv > 0 ? (float)(1.0 / (double)(float)h_cv) : 0.f;
      for (int qq = 0; qq < h_cv; ++qq) {
        const int p = qq < 4 ? h_p[qq < 4 ? qq : 0] : h_so[qq];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = acc[j] + fs[j * n + p] * inv;
      }
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(acc[j] * hd.x_scale));
      sat |= m > 65504.f;
      f16x8 hi, lo;
      split_record(make_float4(acc[0], acc[1], acc[2], acc[3]), make_float4(acc[4], acc[5], acc[6], acc[7]), hd.x_scale, hi, lo);
      hd.xh[(((size_t)bi * G + g) * 2 + 0) * hd.n_max + tid] = *reinterpret_cast<const uint4 *>(&hi);
      hd.xh[(((size_t)bi * G + g) * 2 + 1) * hd.n_max + tid] = *reinterpret_cast<const uint4 *>(&lo);
    }
    if (sat && hd.saturated) atomicOr(hd.saturated, 1);
    if (c0 == 0 && tid == 0) hd.amax_out[bi] = 24576.0f / hd.x_scale;
  }
}

}  // namespace

extern "C" int bdm_pvconv_tail_small(int b, int c, int n, int r, int hidden, const float *coords, const float *grid,
                                     const float *coef, const float *se_mean, const float *w1, const float *w2, const float *add,
                                     long long bs_a, int ld_a, const float *add_coef, float *out, long long bs_o, int ld_o,
                                     const int *cnt, const void *plan_workspace, const int *occ_list, const int *n_occ, int n_max,
                                     float x_scale, void *xh, float *amax_out, int *saturated, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= TAIL_CS && c % TAIL_CS == 0 && n >= 1 && r >= 1 && hidden >= 1 && hidden <= 256 && coords && grid && coef &&
              se_mean && w1 && w2 && out, "pvconv_tail_small: bad arguments (c=%d n=%d r=%d hidden=%d)", c, n, r, hidden);
  const int r3 = r * r * r;
  BDM_REQUIRE((r3 & 3) == 0 && (reinterpret_cast<size_t>(grid) & 15) == 0, "pvconv_tail_small: needs 4 | r^3 and a 16-byte aligned grid");
  BDM_REQUIRE(xh == nullptr || (cnt && plan_workspace && occ_list && n_occ && n_max >= 1 && amax_out && x_scale > 0.f),
              "pvconv_tail_small: incomplete head arguments");
  if (b == 0) return BDM_OK;
  TailHead hd{};
  if (xh != nullptr) {
    VoxWs w = vox_ws(const_cast<void *>(plan_workspace), b, n, r3);
    hd = TailHead{cnt, w.start, w.sorted, occ_list, n_occ, n_max, x_scale, (uint4 *)xh, amax_out, saturated};
  }
  const size_t smem = sizeof(float) * ((size_t)TAIL_CS * r3 + (xh ? (size_t)TAIL_CS * n : 0) + c + (hidden > 32 ? hidden : 32) + TAIL_CS + TAIL_CS * 32);
  BDM_REQUIRE(smem <= 160 * 1024, "pvconv_tail_small: %zu bytes of LDS (r=%d, n=%d): not a small grid", smem, r, n);
  static const bool generic_only = getenv("BDM_TAIL_SMALL_GENERIC") != nullptr;   // (tests: the generic kernel on the fast kernel's sizes)
  if (!generic_only && c <= 256 && hidden <= 32 && n <= 256 && (xh == nullptr || n_max <= 256) && TAIL_CS * (r3 >> 2) <= 4 * TAIL_T) {
    BDM_ALLOW_LDS(pv_tail_small_fast_kernel, smem);
    hipLaunchKernelGGL(pv_tail_small_fast_kernel, dim3(b * (c / TAIL_CS)), dim3(TAIL_T), smem, (hipStream_t)stream, c, n, r, hidden, coords,
                       grid, (const float2 *)coef, se_mean, w1, w2, add, bs_a, ld_a, (const float2 *)add_coef, out, bs_o, ld_o, hd);
    return launch_status("pvconv_tail_small");
  }
  BDM_ALLOW_LDS(pv_tail_small_kernel, smem);
  hipLaunchKernelGGL(pv_tail_small_kernel, dim3(b * (c / TAIL_CS)), dim3(TAIL_T), smem, (hipStream_t)stream, c, n, r, hidden, coords,
                     grid, (const float2 *)coef, se_mean, w1, w2, add, bs_a, ld_a, (const float2 *)add_coef, out, bs_o, ld_o, hd);
  return launch_status("pvconv_tail_small");
}

// =====================================================================================================================================
// Gather of the sparse first convolution + GroupNorm-1 + Swish + fp16 operand split, one workgroup per (shape, GroupNorm group).
//   y (b, n_max, 27, cout) = the batched GEMM's output (row k = occupied cell k, sparse_conv_h2.hip); occ_index (b, r^3) = row of a
//   cell or -1.  out[v][co] = bias[co] + sum over taps t (ascending) of y[occ_index[v + off(t)]][t][co]  -- the sums of
//   sparse_gather_v4_kernel in its order.  The group's tile (cg channels x r^3 cells) stays in LDS; its statistics are reduced in a
//   fixed order inside the workgroup (per item fp32, across items / lanes / waves fp64: independent of the batch); then every record of
//   8 channels is normalised, Swished, scaled by the power of two act_scale and split into (hi, lo) fp16 in the layout of
//   bdm_group_norm_to_h2 (b, cout/8, 2, r^3) x 8.
// =====================================================================================================================================
namespace {

constexpr int GATHER_T = 1024;   // 16 waves: a (shape, group) tile is 4096 (cell, channel quad) items -- four per thread, i.e. four
                                 // dependent trips to the GEMM's output instead of sixteen (256 threads: 64 us per launch, latency-bound)

__device__ __forceinline__ void split2h(float v, unsigned short &h, unsigned short &l) {   // conv3d_h2.hip's split2
  v = fminf(fmaxf(v, -65504.f), 65504.f);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  h = __builtin_bit_cast(unsigned short, hi);
  l = __builtin_bit_cast(unsigned short, lo);
}

__global__ __launch_bounds__(GATHER_T) void gather_h2_small_kernel(int cout, int r, int n_max, int cg, const float *__restrict__ y,
                                                              const int *__restrict__ occ_index, const float *__restrict__ bias,
                                                              const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                              float act_scale, uint4 *__restrict__ out, unsigned *__restrict__ saturated) {
  extern __shared__ __align__(16) float smem[];
  const int r2 = r * r, r3 = r2 * r, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = cout / cg, bi = blockIdx.x / G, g = blockIdx.x % G, Q = cg >> 2, ldt = r3 + 1;
  float *tile = smem;                                        // [cg][r3 + 1]
  int *oi = reinterpret_cast<int *>(tile + (size_t)cg * ldt);  // [r3]
  __shared__ double s_red[GATHER_T / 64][2];
  __shared__ float s_ab[64][2];
  for (int e = tid; e < r3; e += GATHER_T) oi[e] = occ_index[(size_t)bi * r3 + e];
  __syncthreads();
  const float *yb = y + (size_t)bi * n_max * 27 * cout + g * cg;
  const int q = tid % Q;                                     // the thread's channel quad (GATHER_T % Q == 0: the same for all its items)
  const float4 b4 = bias ? *reinterpret_cast<const float4 *>(bias + g * cg + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  double ds = 0.0, dq = 0.0;
  for (int item = tid; item < r3 * Q; item += GATHER_T) {
    const int v = item / Q;
    const int x = v / r2, yy = (v / r) % r, z = v % r;
    // which taps have an occupied input cell (bit t), then ONLY those row pieces, up to 14 in flight, added in ascending tap order
    // (a trip to the GEMM's output per batch: one for almost every cell -- a cell has 3 - 11 occupied neighbours on these levels)
    unsigned mask = 0u;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int gx = x + t / 9 - 1, gy = yy + (t / 3) % 3 - 1, gz = z + t % 3 - 1;
      const bool in = gx >= 0 && gx < r && gy >= 0 && gy < r && gz >= 0 && gz < r;
      mask |= ((in && oi[in ? (gx * r + gy) * r + gz : 0] >= 0) ? 1u : 0u) << t;
    }
    float4 acc = b4;
    while (mask) {
      float4 vv[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) {
        vv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask) {
          const int t = __ffs((int)mask) - 1;
          mask &= mask - 1;
          const int k = oi[((x + t / 9 - 1) * r + (yy + (t / 3) % 3 - 1)) * r + (z + t % 3 - 1)];
          vv[u] = *reinterpret_cast<const float4 *>(yb + ((size_t)k * 27 + t) * cout + q * 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 14; ++u) { acc.x += vv[u].x; acc.y += vv[u].y; acc.z += vv[u].z; acc.w += vv[u].w; }
    }
    float *tp = tile + (size_t)(q * 4) * ldt + v;
    tp[0] = acc.x; tp[ldt] = acc.y; tp[2 * ldt] = acc.z; tp[3 * ldt] = acc.w;
    ds += (double)((acc.x + acc.y) + (acc.z + acc.w));
    dq += (double)((acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { ds += __shfl_xor(ds, o, 64); dq += __shfl_xor(dq, o, 64); }
  if (lane == 0) { s_red[wave][0] = ds; s_red[wave][1] = dq; }
  __syncthreads();
  if (tid < cg) {
    double a = 0.0, qq = 0.0;
    for (int w = 0; w < GATHER_T / 64; ++w) { a += s_red[w][0]; qq += s_red[w][1]; }   // waves in order
    const double cnt = (double)cg * r3, mean = a / cnt;
    double var = qq / cnt - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const int ch = g * cg + tid;
    const float ga = gamma[ch] * rstd;
    s_ab[tid][0] = ga;
    s_ab[tid][1] = beta[ch] - (float)mean * ga;
  }
  __syncthreads();
  // records: (8 channels) x cell, cells on the lanes
  const int C8 = cout / 8, recs = cg / 8;
  bool sat = false;
  for (int item = tid; item < recs * r3; item += GATHER_T) {
    const int rl = item / r3, v = item - rl * r3;
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cl = rl * 8 + j;
      float t = swishf(tile[(size_t)cl * ldt + v] * s_ab[cl][0] + s_ab[cl][1]) * act_scale;
      sat |= !(fabsf(t) <= 65504.f);
      split2h(t, h[j], l[j]);
    }
    uint4 ph, pl;
    ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
    pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
    uint4 *o = out + ((size_t)bi * C8 + (g * cg) / 8 + rl) * 2 * (size_t)r3;
    o[v] = ph;
    o[(size_t)r3 + v] = pl;
  }
  if (saturated != nullptr && __ballot(sat) != 0ull && lane == __ffsll((long long)__ballot(sat)) - 1) atomicOr(saturated, 1u);
}

}  // namespace

extern "C" int bdm_sparse_conv_gather_h2_small(int b, int cout, int r, int n_max, const float *y, const int *occ_index, const float *bias,
                                               int groups, const float *gamma, const float *beta, float eps, float act_scale, void *out_h2,
                                               unsigned int *saturated, void *stream) {
  const int cg = groups >= 1 && cout % groups == 0 ? cout / groups : 0;
  BDM_REQUIRE(b >= 0 && r >= 1 && n_max >= 1 && y && occ_index && gamma && beta && out_h2 && cg >= 8 && cg % 8 == 0 && cg <= 64 &&
              GATHER_T % (cg / 4) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0,
              "sparse_conv_gather_h2_small: needs 8 | channels per group <= 64, (cg / 4) | 256 (cout=%d groups=%d)", cout, groups);
  {
    int ex = 0;
    BDM_REQUIRE(act_scale > 0.f && act_scale < INFINITY && frexpf(act_scale, &ex) == 0.5f,
                "sparse_conv_gather_h2_small: act_scale must be a power of two (got %g)", (double)act_scale);
  }
  if (b == 0) return BDM_OK;
  const int r3 = r * r * r;
  const size_t smem = sizeof(float) * (size_t)cg * (r3 + 1) + sizeof(int) * (size_t)r3;
  BDM_REQUIRE(smem <= 150 * 1024, "sparse_conv_gather_h2_small: %zu bytes of LDS (r=%d, %d channels per group): not a small grid", smem, r, cg);
  BDM_ALLOW_LDS(gather_h2_small_kernel, smem);
  hipLaunchKernelGGL(gather_h2_small_kernel, dim3(b * groups), dim3(GATHER_T), smem, (hipStream_t)stream, cout, r, n_max, cg, y, occ_index, bias,
                     gamma, beta, eps, act_scale, (uint4 *)out_h2, saturated);
  return launch_status("sparse_conv_gather_h2_small");
}

