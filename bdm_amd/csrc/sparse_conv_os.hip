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

// =====================================================================================================================
// Compact output-stationary form (the default): the output voxels that can be non-trivial -- the once-dilated occupied set,
// 23 % of a 32^3 grid for a Gaussian-like cloud, listed in voxel order by bdm_voxel_dilate -- are processed in tiles of TILE
// CONSECUTIVE list entries, so every MFMA column is a voxel that needs computing and every workgroup carries the same number of them
// (the brick form above spends its matrix work on 16-voxel z-runs of a fixed brick: 34 % live fragments, all of them in the centre
// bricks).  Per tile:
//   * inputs: the tile's voxels span x-planes [x0, x1]; their occupied neighbours are compact rows [plane_start[x0-1], plane_start[x1+2])
//     (occ_list is sorted by voxel index), a CONTIGUOUS range: it is staged per 8-channel chunk into LDS as (hi, lo) fp16 records,
//     coalesced, no halo, no zero fill; a lane keeps the LDS slot of its (voxel, tap) neighbour for all 7 tap quads in registers
//     (found once through occ_index; absent neighbours point at a zero record).  A range beyond the LDS budget (a dense slab of
//     cells) is walked in several passes of `xcap` rows, each pass adding the products of the neighbours inside its rows;
//   * skipping: (16-voxel block, tap quad) groups without a present neighbour skip LDS reads and MFMAs as above;
//   * output: the workgroup owns the linear voxel range [first voxel of its tile, first voxel of the next tile) (tile 0 from voxel 0,
//     the last live tile to r^3): it streams bias over that range (16-byte stores), waits, then scatters its computed voxels;
//   * GroupNorm partials: one slice per tile index = the tile's computed voxels + the closed-form share of its bias-filled voxels.
// =====================================================================================================================
__global__ __launch_bounds__(1024) void vox_dilate_kernel(int r, int n_dil_max, int tile, int xcap, int tiles_max,
                                                          const int *__restrict__ cnt, int *__restrict__ dil_list,
                                                          int *__restrict__ plane_start, int *__restrict__ tile_start) {
  // one workgroup per shape: occupancy bit rows (x, y) -> dilated bit rows -> ordered compaction -> tile table
  //   plane_start[x]  occupied cells in planes < x (r + 2 entries: [r] = [r + 1] = n_occ): compact rows of planes [a, b) = [ps[a], ps[b])
  //   tile_start[t]   first dil_list entry of tile t, t = 0 .. n_tiles (tile_start[tiles_max + 1] = n_tiles); a tile holds <= `tile`
  //                   consecutive entries and is cut at an x-plane boundary where the compact rows of planes [x0 - 1, x1 + 1] would
  //                   exceed `xcap` (xcap >= 3 r^2, so a tile inside one plane always fits)
  extern __shared__ unsigned bits[];   // [r*r] occupancy rows, 2 x 16 wave totals, 2 x (r + 2) plane prefixes
  const int bi = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int r2 = r * r, r3 = r2 * r;
  unsigned *occ = bits;
  int *wtot = reinterpret_cast<int *>(bits + r2);         // [2][16]
  int *ps_o = wtot + 32, *ps_d = ps_o + (r + 2);           // occupied / dilated cells before plane x
  const int *c = cnt + (size_t)bi * r3;
  for (int row = tid; row < r2; row += T) {
    unsigned m = 0u;
    for (int z = 0; z < r; ++z) m |= (c[row * r + z] > 0 ? 1u : 0u) << z;
    occ[row] = m;
  }
  __syncthreads();
  const unsigned full = r == 32 ? 0xffffffffu : ((1u << r) - 1u);
  unsigned d = 0u;
  int nocc_row = 0;
  if (tid < r2) {   // one (x, y) row per thread (r2 <= 1024)
    const int x = tid / r, y = tid % r;
    for (int dx = -1; dx <= 1; ++dx)
      for (int dy = -1; dy <= 1; ++dy) {
        const int gx = x + dx, gy = y + dy;
        if (gx < 0 || gx >= r || gy < 0 || gy >= r) continue;
        const unsigned m = occ[gx * r + gy];
        d |= m | (m << 1) | (m >> 1);
      }
    d &= full;
    nocc_row = __popc(occ[tid]);
  }
  int incl_d = __popc(d), incl_o = nocc_row;
  const int mine_d = incl_d, mine_o = incl_o;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int td = __shfl_up(incl_d, o, 64), to = __shfl_up(incl_o, o, 64);
    if (lane >= o) { incl_d += td; incl_o += to; }
  }
  if (lane == 63) { wtot[wave] = incl_d; wtot[16 + wave] = incl_o; }
  __syncthreads();
  int off_d = 0, off_o = 0, tot_d = 0, tot_o = 0;
  for (int w = 0; w < (T >> 6); ++w) {
    if (w < wave) { off_d += wtot[w]; off_o += wtot[16 + w]; }
    tot_d += wtot[w]; tot_o += wtot[16 + w];
  }
  if (tid < r2) {
    int run = off_d + incl_d - mine_d;
    if (tid % r == 0) { ps_o[tid / r] = off_o + incl_o - mine_o; ps_d[tid / r] = run; }
    int *dl = dil_list + (size_t)bi * n_dil_max;
    unsigned m = d;
    while (m) {
      const int z = __ffs((int)m) - 1;
      m &= m - 1;
      if (run < n_dil_max) dl[run] = tid * r + z;
      ++run;
    }
  }
  if (tid == 0) { ps_o[r] = ps_o[r + 1] = tot_o; ps_d[r] = ps_d[r + 1] = tot_d; }
  __syncthreads();
  for (int x = tid; x < r + 2; x += T) plane_start[(size_t)bi * (r + 2) + x] = ps_o[x];
  if (tid == 0) {   // the tile table: a short serial walk over LDS-resident prefixes
    const int nd = min(tot_d, n_dil_max);
    int *ts = tile_start + (size_t)bi * (tiles_max + 2);
    int t = 0, j = 0, x0 = 0;
    while (j < nd && t < tiles_max) {
      while (ps_d[x0 + 1] <= j) ++x0;                     // plane of entry j
      int jend = min(j + tile, nd), x1 = x0;
      while (ps_d[x1 + 1] < jend) ++x1;                    // plane of entry jend - 1
      while (x1 > x0 && ps_o[min(x1 + 2, r)] - ps_o[max(x0 - 1, 0)] > xcap) { jend = ps_d[x1]; --x1; }   // cut at the plane boundary
      ts[t++] = j;
      j = jend;
    }
    ts[t] = nd;                                            // (t == tiles_max with j < nd cannot happen: tiles_max = r^3 / tile + r)
    for (int u = t + 1; u <= tiles_max; ++u) ts[u] = nd;
    ts[tiles_max + 1] = t;
  }
}

static void sconv_dil_geometry(int r, int *tile, int *xcap, int *tiles_max) {
  *tile = r == 32 ? 512 : (r == 16 ? 256 : 128);
  *xcap = 3 * r * r;                                        // compact rows of three full x-planes: a one-plane tile always fits
  *tiles_max = (r * r * r) / *tile + r;
}

extern "C" int bdm_voxel_dilate_slices(int r) {
  int tile, xcap, tiles_max;
  sconv_dil_geometry(r, &tile, &xcap, &tiles_max);
  return tiles_max;
}

extern "C" int bdm_voxel_dilate(int b, int r, int n_dil_max, const int *cnt, int *dil_list, int *plane_start, int *tile_start,
                                void *stream) {
  BDM_REQUIRE(b >= 0 && (r == 8 || r == 16 || r == 32) && n_dil_max >= 1 && cnt && dil_list && plane_start && tile_start,
              "voxel_dilate: bad arguments (r in {8, 16, 32})");
  if (b == 0) return BDM_OK;
  int tile, xcap, tiles_max;
  sconv_dil_geometry(r, &tile, &xcap, &tiles_max);
  const size_t smem = sizeof(unsigned) * ((size_t)r * r + 32 + 2 * (r + 2));
  hipLaunchKernelGGL(vox_dilate_kernel, dim3(b), dim3(1024), smem, (hipStream_t)stream, r, n_dil_max, tile, xcap, tiles_max, cnt,
                     dil_list, plane_start, tile_start);
  return launch_status("voxel_dilate");
}

template <int MT, int NT, int NW, int R>
__global__ __launch_bounds__(NW * 64) void sconv_dil_kernel(
    int C8, int Cout, int n_max, int n_dil_max, int xcap, const float4 *__restrict__ xr, const float *__restrict__ amax,
    const int *__restrict__ occ_index, const int *__restrict__ dil_list, const int *__restrict__ tile_start,
    const int *__restrict__ plane_start, const float4 *__restrict__ wq, const float *__restrict__ inv_scale,
    const float *__restrict__ bias, float *__restrict__ y, int gn_cg, double *__restrict__ gn_partial) {
  extern __shared__ __align__(16) float4 smem4[];
  constexpr int BM = 16 * MT;        // a tile holds <= NT * NW * 16 voxels (sconv_dil_geometry)
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int NT_ = NW * 64;
  constexpr int WV = OS_PAIRS * 2 * 2 * BM, WI = (WV + NT_ - 1) / NT_;
  constexpr int NQ = OS_PAIRS / 2;
  constexpr int PF = 2;
  constexpr int NBLK = NT * NW, NB = MT * 4 * 2;
  float4 *Ws = smem4;                 // [14][2][2][BM]
  float4 *Xs = smem4 + WV;            // [2][xcap + 1]: record xcap of each split is zero
  const int XS = xcap + 1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
  const int tile = blockIdx.x, m0 = blockIdx.y * BM, bi = blockIdx.z;
  const int *ts = tile_start + (size_t)bi * (gridDim.x + 2);   // gridDim.x = tiles_max
  const int tiles_live = ts[gridDim.x + 1];
  const int G = gn_partial != nullptr ? Cout / gn_cg : 0, S = gridDim.x;
  if (tile >= max(tiles_live, 1)) {   // nothing to compute here: an empty slice of the statistics
    if (gn_partial != nullptr) {
      const int ngt = BM / gn_cg;
      if (tid < ngt && m0 + tid * gn_cg < Cout) {
        double *dst = gn_partial + (((size_t)bi * G + m0 / gn_cg + tid) * S + tile) * 2;
        dst[0] = 0.0; dst[1] = 0.0;
      }
    }
    return;
  }
  const int *dl = dil_list + (size_t)bi * n_dil_max;
  const int j0 = ts[tile], jn = ts[tile + 1];                   // <= TILE consecutive entries of the dilated list
  const bool nothing = jn <= j0;                                // (a grid without an occupied cell: tile 0 fills it with bias)
  const int v_first = tile == 0 ? 0 : dl[j0], v_end = tile + 1 < tiles_live ? dl[jn] : R3;   // the linear range this tile owns
  float *yb = y + (size_t)bi * Cout * R3;

  // ---- bias over the owned range (the stores drain while the tile is set up): a wave streams one channel row at a time ----------
  {
    const int a0 = min((v_first + 3) & ~3, v_end), a1 = max(v_end & ~3, a0);   // [v_first, a0) head, [a0, a1) 16-byte body, [a1, v_end) tail
    for (int m = wave; m < BM; m += NW) {
      if (m0 + m >= Cout) break;
      const float bv = bias ? bias[m0 + m] : 0.f;
      float *row = yb + (size_t)(m0 + m) * R3;
      if (lane < a0 - v_first) row[v_first + lane] = bv;
      if (lane < v_end - a1) row[a1 + lane] = bv;
      const float4 b4 = make_float4(bv, bv, bv, bv);
      for (int p = a0 + lane * 4; p < a1; p += 256) *reinterpret_cast<float4 *>(row + p) = b4;
    }
  }

  // ---- tile set-up: input range, per-lane neighbour slots, skip mask --------------------------------------------------------
  const int x0 = nothing ? 0 : dl[j0] / R2, x1 = nothing ? 0 : dl[jn - 1] / R2;
  const int *ps = plane_start + (size_t)bi * (R + 2);
  const int k_lo = ps[max(x0 - 1, 0)], k_hi = ps[min(x1 + 2, R)];
  const int nrows = nothing ? 0 : k_hi - k_lo;
  int rec[NT][NQ];                    // LDS record of this lane's (voxel, tap) neighbour; xcap = the zero record (nrows <= xcap by construction)
  int vox[NT];
  unsigned amask = 0u;
  {
    const int *oi = occ_index + (size_t)bi * R3;
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int j = j0 + (q * NW + wave) * 16 + l16;
      const int v = j < jn ? dl[j] : -1;
      vox[q] = v;
      const int vx = v / R2, vy = (v / R) % R, vz = v % R;
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q) {
        const int t = 4 * Q + kg;
        int k = -1;
        if (v >= 0 && t < 27) {
          const int gx = vx + t / 9 - 1, gy = vy + (t / 3) % 3 - 1, gz = vz + t % 3 - 1;
          if (gx >= 0 && gx < R && gy >= 0 && gy < R && gz >= 0 && gz < R) k = oi[(gx * R + gy) * R + gz];
        }
        const bool here = k >= k_lo && k < k_hi;               // (always, for a present neighbour: the range covers planes x0-1 .. x1+1)
        if (__ballot(here) != 0ull) amask |= 1u << (Q * NT + q);
        rec[q][Q] = here ? k - k_lo : xcap;
      }
    }
    amask = __builtin_amdgcn_readfirstlane(amask);
  }
  const float sx = act_scale_from_max(amax[bi]);
  const float4 *wbase = Ws + ((kg >> 1) * 4 + (kg & 1)) * BM + l16;
  f32x4a acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[a][q] = f32x4a{0.f, 0.f, 0.f, 0.f};
  if (tid < 2) Xs[tid * XS + xcap] = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero record of both splits

  const float4 *xb = xr + (size_t)bi * C8 * n_max * 2;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v xa[PF][2], wr[WI];
  const int stage_rows = min(nrows, xcap);
  auto load_chunk = [&](int c8) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int row = min(tid + u * NT_, max(stage_rows - 1, 0));
      const f32x4v *src = reinterpret_cast<const f32x4v *>(xb + ((size_t)c8 * n_max + k_lo + row) * 2);
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
  auto put = [&](int row, const float4 &p, const float4 &q) {
    f16x8 hi, lo;
    split_record(p, q, sx, hi, lo);
    *reinterpret_cast<f16x8 *>(&Xs[row]) = hi;
    *reinterpret_cast<f16x8 *>(&Xs[XS + row]) = lo;
  };
  auto store_chunk = [&](int c8) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (tid + u * NT_ < stage_rows) {
        const float4 p = make_float4(xa[u][0][0], xa[u][0][1], xa[u][0][2], xa[u][0][3]);
        const float4 q = make_float4(xa[u][1][0], xa[u][1][1], xa[u][1][2], xa[u][1][3]);
        put(tid + u * NT_, p, q);
      }
    for (int row = tid + PF * NT_; row < stage_rows; row += NT_) {
      const float4 *src = xb + ((size_t)c8 * n_max + k_lo + row) * 2;
      put(row, src[0], src[1]);
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * NT_;
      if (e < WV) *reinterpret_cast<f32x4v *>(&Ws[e]) = wr[i];
    }
  };

  if (!nothing) {   // (uniform per workgroup)
    load_chunk(0);
    for (int c8 = 0; c8 < C8; ++c8) {
      __syncthreads();
      store_chunk(c8);
      __syncthreads();
      if (c8 + 1 < C8) load_chunk(c8 + 1);
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q) {
        const unsigned qm = (amask >> (Q * NT)) & ((1u << NT) - 1u);
        if (qm == 0u) continue;
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
            const float4 t = Xs[s * XS + rec[q][Q]];
            fb[s] = *reinterpret_cast<const f16x8 *>(&t);
          }
#pragma unroll
          for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt][term == 0 ? 1 : 0], fb[term == 1 ? 1 : 0], acc[mt][q], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: the bias stores of this workgroup have landed (the barrier drains vmcnt); scatter the computed voxels ---------
  __syncthreads();
  const float x_inv_scale = 1.0f / sx;
  float *red = reinterpret_cast<float *>(smem4);            // [NBLK][NB], then [NB]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      float bs = 0.f, bq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + mt * 16 + 4 * kg + i;
        if (m < Cout && vox[q] >= 0) {
          const float v = acc[mt][q][i] * (inv_scale[m] * x_inv_scale) + (bias ? bias[m] : 0.f);
          yb[(size_t)m * R3 + vox[q]] = v;
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
    for (int e = tid; e < NB; e += NT_) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < NBLK; ++j) a += red[j * NB + e];   // the tile's blocks in ascending order
      red2[e] = a;
    }
    __syncthreads();
    const int ngt = BM / gn_cg;
    if (tid < ngt && m0 + tid * gn_cg < Cout) {
      double a = 0.0, qq = 0.0;
      const int nb4 = gn_cg / 4;
      for (int j = 0; j < nb4; ++j) {
        a += (double)red2[(tid * nb4 + j) * 2 + 0];
        qq += (double)red2[(tid * nb4 + j) * 2 + 1];
      }
      // the bias-filled voxels of the owned range, in closed form
      const double nfill = (double)((v_end - v_first) - (jn - j0));
      for (int c = 0; c < gn_cg; ++c) {
        const double bv = bias ? (double)bias[m0 + tid * gn_cg + c] : 0.0;
        a += nfill * bv;
        qq += nfill * bv * bv;
      }
      double *dst = gn_partial + (((size_t)bi * G + m0 / gn_cg + tid) * S + tile) * 2;
      dst[0] = a;
      dst[1] = qq;
    }
  }
}

static int sconv_dil_launch(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                            const int *occ_index, const int *dil_list, const int *tile_start, const int *plane_start, const void *packed_w,
                            const float *inv_scale, const float *bias, float *y, int gn_cg, double *gn_partial, int *slices_out,
                            void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && n_max >= 1 && n_dil_max >= 1 && xr && amax && occ_index && dil_list && tile_start &&
                  plane_start && inv_scale,
              "sparse_conv_dil: bad arguments");
  if (r != 8 && r != 16 && r != 32) {
    set_error("sparse_conv_dil: resolution %d unsupported (8, 16, 32)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  if (b == 0) return BDM_OK;
  const int c8 = (cin + 7) / 8;
  const int mi = cout > 32 ? 2 : 1;                     // 64 or 32 output channels per workgroup
  int tile, xcap, tiles;
  sconv_dil_geometry(r, &tile, &xcap, &tiles);
  const size_t smem = 16 * ((size_t)OS_PAIRS * 4 * 32 * mi + 2 * ((size_t)xcap + 1));
  dim3 grid(tiles, cdiv(cout, 32 * mi), b);
  hipStream_t s = (hipStream_t)stream;
  if (gn_partial != nullptr) {
    BDM_REQUIRE(gn_cg >= 4 && (gn_cg & (gn_cg - 1)) == 0 && (32 * mi) % gn_cg == 0 && cout % gn_cg == 0,
                "sparse_conv_dil: GroupNorm statistics need a power-of-two channels-per-group dividing %d (got cg=%d)", 32 * mi, gn_cg);
    if (slices_out) *slices_out = tiles;
  }
#define DIL_LAUNCH(MT, NT, NW, R)                                                                                         \
  do {                                                                                                                    \
    BDM_ALLOW_LDS((sconv_dil_kernel<MT, NT, NW, R>), smem);                                                               \
    hipLaunchKernelGGL((sconv_dil_kernel<MT, NT, NW, R>), grid, dim3(NW * 64), smem, s, c8, cout, n_max, n_dil_max, xcap, \
                       (const float4 *)xr, amax, occ_index, dil_list, tile_start, plane_start, (const float4 *)packed_w,  \
                       inv_scale, bias, y, gn_cg, gn_partial);                                                            \
  } while (0)
  if (r == 32) { if (mi == 2) DIL_LAUNCH(4, 4, 8, 32); else DIL_LAUNCH(2, 4, 8, 32); }
  else if (r == 16) { if (mi == 2) DIL_LAUNCH(4, 2, 8, 16); else DIL_LAUNCH(2, 2, 8, 16); }
  else { if (mi == 2) DIL_LAUNCH(4, 1, 8, 8); else DIL_LAUNCH(2, 1, 8, 8); }
#undef DIL_LAUNCH
  return launch_status("sparse_conv_dil");
}

// slices of the GroupNorm partials the _gn form leaves per (shape, group) = tiles of the grid (bdm_voxel_dilate_slices(r))
extern "C" int bdm_sparse_conv_dil(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                                   const int *occ_index, const int *dil_list, const int *tile_start, const int *plane_start,
                                   const void *packed_w, const float *inv_scale, const float *bias, float *y, void *stream) {
  return sconv_dil_launch(b, cin, cout, r, n_max, n_dil_max, xr, amax, occ_index, dil_list, tile_start, plane_start, packed_w, inv_scale,
                          bias, y, 0, nullptr, nullptr, stream);
}

extern "C" int bdm_sparse_conv_dil_gn(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                                      const int *occ_index, const int *dil_list, const int *tile_start, const int *plane_start,
                                      const void *packed_w, const float *inv_scale, const float *bias, float *y, int groups,
                                      void *gn_partial, int *slices_out, void *stream) {
  BDM_REQUIRE(groups >= 1 && cout % groups == 0 && gn_partial != nullptr && slices_out != nullptr, "sparse_conv_dil_gn: bad arguments");
  return sconv_dil_launch(b, cin, cout, r, n_max, n_dil_max, xr, amax, occ_index, dil_list, tile_start, plane_start, packed_w, inv_scale,
                          bias, y, cout / groups, (double *)gn_partial, slices_out, stream);
}
