// sparse_conv_os.hip -- the FIRST 3x3x3 convolution of a PVConv as ONE output-stationary implicit GEMM over the occupied input
// cells, with tap skipping (round 4; replaces features-split + batched GEMM + gather of sparse_conv.hip / sparse_conv_h2.hip
// wherever the input is not the hoisted conditioning map).
//
// The input of that convolution is the freshly voxelised cloud (pvconv.py:93-94): 2 - 16 % of the r^3 cells are non-zero.  The
// GEMM + gather form writes and re-reads a (n_occ, 27 * Cout) fp32 intermediate (27x the output's non-trivial part: ~5x the
// algorithmic traffic, profiles/r03_pmc_traffic.json).  Here a workgroup owns a TX x TY x R brick of OUTPUT voxels x BM output
// channels, exactly like the dense fp16x3 convolution (conv3d_h2.hip, same MFMA shape v_mfma_f32_16x16x32_f16, same weight image,
// same K order: 8 channels x tap quads), but
//   * its B operand is built in LDS from the COMPACT fp32 rows of the occupied cells (bdm_sparse_voxel_features_f32: (B, C/8, n_max)
//     records of 8 channels + per-shape max |value|): the brick's halo is zero-filled once, the occupied cells found through
//     occ_index are listed once, and per 8-channel chunk only those cells are fetched (32 bytes each), scaled by the shape's power
//     of two, split into (hi, lo) fp16 and written to their halo records -- no dense input grid, no fp16 split pass;
//   * a wave knows, per (16-voxel block, tap quad), whether ANY of the 64 (voxel, tap) neighbours is occupied (one ballot per pair
//     in the prologue, a 28-bit wave-uniform mask): all-zero fragments skip their LDS reads and MFMAs.  For Gaussian-like clouds
//     35 % of the fragments are live at 32^3, ~50 % at 16^3 / 8^3 (tools/tile_activity.py); bricks without an occupied cell in
//     their halo write bias and leave;
//   * no intermediate: the output grid is written once, with the GroupNorm-1 slice partials of the dense kernel's canonical
//     decomposition (bdm_group_norm_to_h2_stats consumes them).
// Arithmetic: fp16x3 (lo.hi + hi.lo + hi.hi, fp32 accumulate), per-output-channel weight scale, per-SHAPE activation scale:
// fp32-grade (<= 3e-7 relative L2 vs fp64) and independent of a shape's batch-mates.  Deterministic: the order in which occupied
// cells are listed varies (LDS atomic), the value written to each halo record and every sum do not.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

#include "sparse_h2_common.h"

#define OS_PAIRS 14  // the dense kernel's weight image: [C8][14 tap pairs][2 splits][2 halves][Cout] records of 8 fp16

typedef __attribute__((ext_vector_type(4))) float f32x4a;

template <int MT, int NT, int R, int TX, int TY, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void sconv_os_kernel(
    int C8, int Cout, int n_max, const float4 *__restrict__ xr, const float *__restrict__ amax, const int *__restrict__ occ_index,
    const float4 *__restrict__ wq, const float *__restrict__ inv_scale, const float *__restrict__ bias, float *__restrict__ y,
    int gn_cg, double *__restrict__ gn_partial, int dbg) {
  extern __shared__ __align__(16) float4 smem4[];
  constexpr int BM = 16 * MT;
  constexpr int RSV = R + 2;
  constexpr int ROWS = (TX + 2) * (TY + 2);
  constexpr int HALO = ROWS * RSV;
  constexpr int CELLS = ROWS * R;            // halo cells inside a full grid row range
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int NT_ = NW * 64;
  constexpr int WV = OS_PAIRS * 2 * 2 * BM, WI = (WV + NT_ - 1) / NT_;
  constexpr int NQ = OS_PAIRS / 2;
  constexpr int PF = 2;                      // occupied cells per thread whose rows are register-prefetched one chunk ahead
  static_assert(TX * TY * R == NT * NW * 16, "tile = NT * NW blocks of 16 voxels");
  static_assert(NT * NQ <= 32, "skip mask fits one word");
  float4 *Xs = smem4;                        // [2][HALO]
  float4 *Ws = smem4 + 2 * HALO;             // [14][2][2][BM]
  int *list_s = reinterpret_cast<int *>(Ws + WV);                            // [CELLS]: (halo record << 18) | compact row
  unsigned char *occ_s = reinterpret_cast<unsigned char *>(list_s + CELLS);  // [HALO] 1 = occupied cell
  int *count_s = reinterpret_cast<int *>(occ_s + ((HALO + 15) & ~15));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  constexpr int tiles_y = R / TY, tiles_x = R / TX;
  const int m0 = blockIdx.y * BM, bi = blockIdx.z;
  // Brick of this workgroup.  Workgroups go round-robin over the 8 XCDs by their linear id, and the work of a brick depends on where
  // it lies (centre bricks of a cloud: every fragment live; rim bricks: pure bias), so a plain row-major decode hands all the heavy
  // bricks to the same XCDs (measured: no gain from skipping 65 % of the MFMA groups).  The decode below gives every residue class
  // mod 8 two bricks of each y-column at four different x (32^3), or rotates the x-slab with the shape and channel block (16^3).
  int tile;
  if (tiles_x * tiles_y == 64 && tiles_y == 4) {
    const int c = (blockIdx.x + bi) & 7, sl = blockIdx.x >> 3, yi = sl & 3, h = sl >> 2;
    const int xi = ((c - 4 * (yi & 1) - 2 * (yi >> 1)) & 7) + 8 * h;
    tile = xi * tiles_y + yi;
  } else if (tiles_x * tiles_y == 8) {
    tile = (blockIdx.x + 3 * blockIdx.y + bi) & 7;
  } else {
    tile = blockIdx.x;
  }
  const int X0 = (tile / tiles_y) * TX, Y0 = (tile % tiles_y) * TY;
  float *yb = y + (size_t)bi * Cout * R3;

  constexpr int RQ = NW * 16 / R;
  static_assert(NW * 16 % R == 0 && (RQ % TY == 0 || TY % RQ == 0), "the blocks of one q are whole rows that tile the brick");
  const int v0 = wave * 16 + l16, r0 = v0 / R, tz = v0 % R;
  const int lbase0 = ((r0 / TY + 1) * (TY + 2) + (r0 % TY + 1)) * RSV + 1 + tz;
  auto row_of = [&](int q, int &tx, int &ty) {
    if (RQ % TY == 0) { tx = q * (RQ / TY); ty = 0; }
    else { tx = (q * RQ) / TY; ty = (q * RQ) % TY; }
  };
  auto lbase = [&](int q) { int tx, ty; row_of(q, tx, ty); return lbase0 + (tx * (TY + 2) + ty) * RSV; };
  int toff[NQ];
#pragma unroll
  for (int Q = 0; Q < NQ; ++Q) {
    const int t = min(4 * Q + kg, 26);
    toff[Q] = ((t / 9 - 1) * (TY + 2) + ((t / 3) % 3 - 1)) * RSV + (t % 3 - 1);
  }
  const float4 *wbase = Ws + ((kg >> 1) * 4 + (kg & 1)) * BM + l16;
  f32x4a acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[a][q] = f32x4a{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: zero halo, list the occupied cells of the halo, per-wave skip mask ------------------------------------------
  for (int e = tid; e < 2 * HALO; e += NT_) Xs[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = tid; e < (HALO + 3) / 4; e += NT_) reinterpret_cast<unsigned *>(occ_s)[e] = 0u;
  if (tid == 0) *count_s = 0;
  __syncthreads();
  {
    const int *oi = occ_index + (size_t)bi * R3;
    for (int e = tid; e < CELLS; e += NT_) {
      const int z = e % R, row = e / R;
      const int gx = X0 + row / (TY + 2) - 1, gy = Y0 + row % (TY + 2) - 1;
      if (gx >= 0 && gx < R && gy >= 0 && gy < R) {
        const int k = oi[(gx * R + gy) * R + z];
        if (k >= 0) {
          const int pos = row * RSV + 1 + z;
          list_s[atomicAdd(count_s, 1)] = (pos << 18) | k;
          occ_s[pos] = 1;
        }
      }
    }
  }
  __syncthreads();
  const int count = *count_s;
  unsigned amask = 0u;  // bit Q * NT + q: fragment (block q, tap quad Q) has an occupied neighbour
  if (count > 0) {
#pragma unroll
    for (int Q = 0; Q < NQ; ++Q)
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const bool live = occ_s[lbase(q) + toff[Q]] != 0;
        if (__ballot(live) != 0ull) amask |= 1u << (Q * NT + q);
      }
    amask = __builtin_amdgcn_readfirstlane(amask);
    if (dbg == 1) amask = 0u;
    if (dbg == 2) amask = 0xfffffffu;
    if (dbg == 4) amask = 0x0000fffu;
    if (dbg == 5) amask &= 0x1111111u;
  }

  if (count > 0 && dbg != 3) {
    const float sx = act_scale_from_max(amax[bi]);
    int my_pos[PF], my_k[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int e = tid + u * NT_;
      const int ent = list_s[min(e, count - 1)];
      my_pos[u] = ent >> 18;
      my_k[u] = ent & 0x3ffff;
    }
    const float4 *xb = xr + (size_t)bi * C8 * n_max * 2;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v xa[PF][2], wr[WI];
    auto load_chunk = [&](int c8) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {   // clamped entries re-read a valid row: never stored
        const f32x4v *src = reinterpret_cast<const f32x4v *>(xb + ((size_t)c8 * n_max + my_k[u]) * 2);
        xa[u][0] = src[0];
        xa[u][1] = src[1];
      }
#pragma unroll
      for (int i = 0; i < WI; ++i) {
        const int e = tid + i * NT_;
        const int m = e % BM, psh = e / BM;
        const bool ok = e < WV && m0 + m < Cout;
        wr[i] = *reinterpret_cast<const f32x4v *>(&wq[ok ? (unsigned)((c8 * (OS_PAIRS * 4) + psh) * Cout + m0 + m) : 0u]);
      }
    };
    auto put = [&](int pos, const float4 &p, const float4 &q) {
      f16x8 hi, lo;
      split_record(p, q, sx, hi, lo);
      *reinterpret_cast<f16x8 *>(&Xs[pos]) = hi;
      *reinterpret_cast<f16x8 *>(&Xs[HALO + pos]) = lo;
    };
    auto store_chunk = [&](int c8) {
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (tid + u * NT_ < count) {
          const float4 p = make_float4(xa[u][0][0], xa[u][0][1], xa[u][0][2], xa[u][0][3]);
          const float4 q = make_float4(xa[u][1][0], xa[u][1][1], xa[u][1][2], xa[u][1][3]);
          put(my_pos[u], p, q);
        }
      for (int e = tid + PF * NT_; e < count; e += NT_) {  // bricks with more than PF * NT_ occupied halo cells: unprefetched
        const int ent = list_s[e];
        const float4 *src = xb + ((size_t)c8 * n_max + (ent & 0x3ffff)) * 2;
        put(ent >> 18, src[0], src[1]);
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
      store_chunk(c8);
      __syncthreads();
      if (c8 + 1 < C8) load_chunk(c8 + 1);
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q) {
        const unsigned qm = (amask >> (Q * NT)) & ((1u << NT) - 1u);
        if (qm == 0u) continue;   // wave-uniform: no block of this wave has an occupied neighbour under this tap quad
        f16x8 fa[MT][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float4 t = wbase[Q * 8 * BM + s * 2 * BM + mt * 16];
            fa[mt][s] = *reinterpret_cast<const f16x8 *>(&t);
          }
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          if ((qm & (1u << q)) == 0u) continue;
          f16x8 fb[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const float4 t = Xs[s * HALO + lbase(q) + toff[Q]];
            fb[s] = *reinterpret_cast<const f16x8 *>(&t);
          }
          // smallest terms first: lo.hi, hi.lo, hi.hi; term-major over the MT independent accumulators
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt][term == 0 ? 1 : 0], fb[term == 1 ? 1 : 0], acc[mt][q], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: scale + bias + store (+ GroupNorm slice partials on the dense kernel's canonical decomposition) --------------
  const float x_inv_scale = count > 0 ? 1.0f / act_scale_from_max(amax[bi]) : 0.f;
  constexpr int UB = (R == 32 ? TX * TY * R : R * R) / 16;
  constexpr int NBLK = NT * NW, UN = NBLK / UB;
  constexpr int NB = MT * 4 * 2;
  static_assert(NBLK % UB == 0 && UN >= 1, "a tile holds whole canonical units");
  float *red = reinterpret_cast<float *>(smem4);
  if (gn_partial != nullptr) __syncthreads();
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
          const float v = acc[mt][q][i] * (inv_scale[m] * x_inv_scale) + (bias ? bias[m] : 0.f);
          yb[(size_t)m * R3 + gvox] = v;
          bs += v;
          bq = __builtin_fmaf(v, v, bq);
        }
      }
      if (gn_partial != nullptr) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          bs += __shfl_xor(bs, o, 64);
          bq += __shfl_xor(bq, o, 64);
        }
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
      for (int j = 0; j < UB; ++j) a += red[(un * UB + j) * NB + k];
      red2[e] = a;
    }
    __syncthreads();
    const int ngt = BM / gn_cg;
    if (tid < UN * ngt) {
      const int un = tid / ngt, gi = tid % ngt;
      if (m0 + gi * gn_cg < Cout) {
        double a = 0.0, qq = 0.0;
        const int nb4 = gn_cg / 4;
        for (int j = 0; j < nb4; ++j) {
          a += (double)red2[un * NB + (gi * nb4 + j) * 2 + 0];
          qq += (double)red2[un * NB + (gi * nb4 + j) * 2 + 1];
        }
        const int G = Cout / gn_cg, g = m0 / gn_cg + gi, S = gridDim.x * UN;
        double *dst = gn_partial + (((size_t)bi * G + g) * S + tile * UN + un) * 2;   // slice = brick position, not launch order
        dst[0] = a;
        dst[1] = qq;
      }
    }
  }
}

static int sconv_os_launch(int b, int cin, int cout, int r, int n_max, const void *xr, const float *amax, const int *occ_index,
                           const void *packed_w, const float *inv_scale, const float *bias, float *y, int gn_cg,
                           double *gn_partial, int *slices_out, void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && n_max >= 1 && n_max < (1 << 18) && xr != nullptr && amax != nullptr &&
                  occ_index != nullptr && inv_scale != nullptr,
              "sparse_conv_os: bad arguments");
  if (r != 8 && r != 16 && r != 32) {
    set_error("sparse_conv_os: resolution %d unsupported (8, 16, 32)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const int c8 = (cin + 7) / 8;
  int tx, ty;
  if (r == 32) { tx = 2; ty = 8; }
  else if (r == 16) { tx = 2; ty = 16; }
  else { tx = 4; ty = 8; }
  const int mi = (cout > 32 && r != 8) ? 2 : 1;
  const int rows = (tx + 2) * (ty + 2), halo = rows * (r + 2);
  const size_t smem = 16 * ((size_t)2 * halo + (size_t)OS_PAIRS * 4 * 32 * mi) + 4 * (size_t)rows * r + ((halo + 15) & ~15) + 16;
  dim3 grid((r / tx) * (r / ty), cdiv(cout, 32 * mi), b);
  hipStream_t s = (hipStream_t)stream;
  const int dbg = getenv("BDM_OS_DBG") ? atoi(getenv("BDM_OS_DBG")) : 0;
  if (gn_partial != nullptr) {
    BDM_REQUIRE(gn_cg >= 4 && (gn_cg & (gn_cg - 1)) == 0 && (32 * mi) % gn_cg == 0 && cout % gn_cg == 0 && (int)grid.x <= 64,
                "sparse_conv_os: GroupNorm statistics need a power-of-two channels-per-group dividing %d (got cg=%d)", 32 * mi, gn_cg);
    if (slices_out) *slices_out = r == 16 ? 16 : (r == 8 ? 8 : (int)grid.x);
  }
#define OS_LAUNCH(MT, NT, R, TX, TY, NW)                                                                              \
  do {                                                                                                                \
    BDM_ALLOW_LDS((sconv_os_kernel<MT, NT, R, TX, TY, NW>), smem);                                                    \
    hipLaunchKernelGGL((sconv_os_kernel<MT, NT, R, TX, TY, NW>), grid, dim3(NW * 64), smem, s, c8, cout, n_max,       \
                       (const float4 *)xr, amax, occ_index, (const float4 *)packed_w, inv_scale, bias, y, gn_cg,      \
                       gn_partial, dbg);                                                                                   \
  } while (0)
  if (r == 32) { if (mi == 2) OS_LAUNCH(4, 4, 32, 2, 8, 8); else OS_LAUNCH(2, 4, 32, 2, 8, 8); }
  else if (r == 16) { if (mi == 2) OS_LAUNCH(4, 4, 16, 2, 16, 8); else OS_LAUNCH(2, 4, 16, 2, 16, 8); }
  else OS_LAUNCH(2, 2, 8, 4, 8, 8);
#undef OS_LAUNCH
  return launch_status("sparse_conv_os");
}

extern "C" int bdm_sparse_conv_os(int b, int cin, int cout, int r, int n_max, const void *xr, const float *amax,
                                  const int *occ_index, const void *packed_w, const float *inv_scale, const float *bias, float *y,
                                  void *stream) {
  return sconv_os_launch(b, cin, cout, r, n_max, xr, amax, occ_index, packed_w, inv_scale, bias, y, 0, nullptr, nullptr, stream);
}

extern "C" int bdm_sparse_conv_os_gn(int b, int cin, int cout, int r, int n_max, const void *xr, const float *amax,
                                     const int *occ_index, const void *packed_w, const float *inv_scale, const float *bias, float *y,
                                     int groups, void *gn_workspace, int *slices_out, void *stream) {
  BDM_REQUIRE(groups >= 1 && cout % groups == 0 && gn_workspace != nullptr && slices_out != nullptr, "sparse_conv_os_gn: bad arguments");
  return sconv_os_launch(b, cin, cout, r, n_max, xr, amax, occ_index, packed_w, inv_scale, bias, y, cout / groups,
                         (double *)gn_workspace, slices_out, stream);
}
