// sparse_conv_os.hip -- the FIRST 3x3x3 convolution of a PVConv as ONE output-stationary implicit GEMM over the voxels that can be
// non-trivial, with tap skipping (round 4; replaces features-split + batched GEMM + gather of sparse_conv.hip / sparse_conv_h2.hip
// wherever the input is not the hoisted conditioning map).
//
// The input of that convolution is the freshly voxelised cloud (pvconv.py:93-94): 2 - 16 % of the r^3 cells are non-zero, so the
// output differs from the bias only on the once-DILATED occupied set (23 % of a 32^3 grid for a Gaussian-like cloud, 37 % at 16^3,
// ~60 % at 8^3).  The GEMM + gather form writes and re-reads a (n_occ, 27 * Cout) fp32 intermediate: ~5x the algorithmic traffic
// (profiles/r03_pmc_traffic.json).  Here
//   * bdm_voxel_dilate (part of the voxel plan, once per level) lists the dilated voxels in voxel order, with their rank
//     (dil_index), the per-x-plane prefix of the occupied cells and a table of TILES: <= TILE consecutive list entries whose occupied
//     neighbours -- the compact rows of planes x0-1 .. x1+1, ONE contiguous range because the occupied list is sorted -- fit the LDS
//     budget (tiles are cut at plane boundaries where they would not; the budget holds three full planes, so a tile inside one
//     plane always fits);
//   * sconv_dil_kernel: a workgroup owns one tile x BM output channels.  Same MFMA core as the dense fp16x3 convolution
//     (conv3d_h2.hip: v_mfma_f32_16x16x32_f16, same weight image, K = 8 channels x tap quads), but the B operand is gathered: per
//     8-channel chunk the tile's row range is staged (32 bytes per row, coalesced, scaled by the shape's power of two, split into
//     hi / lo fp16), and a lane reads the record of ITS (voxel, tap) neighbour, found once through occ_index and kept in registers
//     for all 7 tap quads (absent neighbours point at a zero record).  (16-voxel block, tap quad) groups without any present
//     neighbour skip their LDS reads and MFMAs (a 28-bit wave-uniform mask);
//   * output, COMPACT form (what the PVConv uses): one row of Cout floats per list entry, (b, n_dil_max, Cout) -- written as 16-byte
//     pieces straight from the accumulators; the dense grid is never materialised: bdm_group_norm_to_h2_stats_compact
//     (conv3d_h2.hip) reads the rows through dil_index and knows every other voxel is bias.  DENSE form: (b, Cout, r^3), the tile
//     goes through LDS and the workgroup streams the linear voxel range it owns, bias in the gaps (for callers that want the grid);
//   * GroupNorm-1 partials: one slice per tile = its computed voxels (fixed summation order) + the closed-form share of the
//     bias voxels of the linear range it owns.
// A fixed-brick form of the same idea (output bricks of the dense kernel, halo built from the occupied cells, same skipping) was
// built first and measured: 34 % of its fragments are live, all in the centre bricks, and a workgroup's waves wait for its busiest
// one at every chunk -- 255 us against 287 us without skipping for 64 -> 64 at 32^3 (profiles/r04_sparse_brick_probe.txt); removed.
// Arithmetic: fp16x3 (lo.hi + hi.lo + hi.hi, fp32 accumulate), per-output-channel weight scale, per-SHAPE activation scale:
// fp32-grade (<= 3e-7 relative L2 vs fp64) and independent of a shape's batch-mates.  Deterministic.
#include <stdlib.h>

#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

#include "sparse_h2_common.h"

#define OS_PAIRS 14  // the dense kernel's weight image: [C8][14 tap pairs][2 splits][2 halves][Cout] records of 8 fp16

typedef __attribute__((ext_vector_type(4))) float f32x4a;
typedef float f32x4v_t __attribute__((ext_vector_type(4)));

// (row16_sum: common.h)

// ---------------------------------------------------------------------------------------------------------------------
// plan side: dilated voxel list, ranks, plane prefixes, tile table
// ---------------------------------------------------------------------------------------------------------------------
#define TILE_REC 16      // ints per tile record (both forms; the readers of other files only use [1] of the LAST record: entries of the list)
#define DILH_XCAP 1376   // half-tile form: compact input rows a tile keeps in LDS (two 80-KB workgroups per CU)

// form 0 ("full" tiles, round 4): <= 512 / 256 / 128 entries, ONE input row range (planes x0-1 .. x1+1) of <= 3 r^2 rows, cut at plane
// boundaries to fit.  form `tile` in {64, 128, 256} ("half" tiles, round 6): `tile` entries, never across an x-plane unless the
// planes' rows fit DILH_XCAP; a tile inside one plane lists THREE row ranges (planes x-1, x, x+1, y-rows y0-1 .. y1+1 of each).
static void sconv_dil_geometry(int r, int half_tile, int *tile, int *xcap, int *tiles_max) {
  if (half_tile > 0) {
    *tile = half_tile;
    *xcap = DILH_XCAP;
  } else {
    *tile = r == 32 ? 512 : (r == 16 ? 256 : 128);
    *xcap = 3 * r * r;                                      // compact rows of three full x-planes: a one-plane tile always fits
  }
  *tiles_max = (r * r * r) / *tile + r;
}

__global__ __launch_bounds__(1024) void vox_dilate_kernel(int r, int n_dil_max, int tile, int xcap, int tiles_max, int half_form,
                                                          const int *__restrict__ cnt, int thr, int *__restrict__ dil_list,
                                                          int *__restrict__ dil_index, int *__restrict__ plane_start,
                                                          int *__restrict__ tile_start, int *__restrict__ class_count) {
  // one workgroup per shape: occupancy bit rows (x, y) -> dilated bit rows -> ordered compaction -> tile table
  //   dil_index[v]    rank of voxel v in dil_list, -1 outside the dilated set
  //   plane_start[x]  occupied cells in planes < x (r + 2 entries: [r] = [r + 1] = n_occ): compact rows of planes [a, b) = [ps[a], ps[b])
  //   tile_start[t]   8 ints per tile (see the end of this kernel); a tile holds <= `tile` consecutive entries and is cut at an x-plane
  //                   boundary where the compact rows of planes [x0 - 1, x1 + 1] would exceed `xcap` (xcap >= 3 r^2, so a tile inside
  //                   one plane always fits)
  extern __shared__ unsigned bits[];   // [r*r] occupancy rows, 2 x 16 wave totals, 2 x (r + 2) plane prefixes, [r*r] dilated rows, [r*r] their prefix,
                                       // [r*r + 1] input cells before each (x, y) row (half form)
  const int bi = blockIdx.x, tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int r2 = r * r, r3 = r2 * r;
  unsigned *occ = bits;
  int *wtot = reinterpret_cast<int *>(bits + r2);         // [2][16]
  int *ps_o = wtot + 32, *ps_d = ps_o + (r + 2);           // occupied / dilated cells before plane x
  const int *c = cnt + (size_t)bi * r3;
  // a wave turns 64 consecutive cells into bits with one ballot (coalesced reads): r = 32: two rows, r = 16: four, r = 8: eight.
  // Eight loads in flight per lane: one load -> ballot -> next load was 32 dependent round trips at 32^3 (r^3 / 1024 threads), i.e.
  // most of this single-workgroup-per-shape kernel's 32 - 40 us
  for (int v0 = wave * 64; v0 < r3; v0 += 8 * T) {
    int cv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) cv[u] = v0 + u * T < r3 ? c[v0 + u * T + lane] : thr;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int vu = v0 + u * T;
      if (vu < r3) {   // (wave-uniform)
        const unsigned long long m = __ballot(cv[u] > thr);   // a cell of the input set (thr = 0: point counts, -1: ranks of a list)
        const int rows = 64 / r;
        if (lane < rows) occ[vu / r + lane] = (unsigned)((m >> (lane * r)) & (r == 32 ? 0xffffffffull : ((1ull << r) - 1ull)));
      }
    }
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
  unsigned *drow = bits + r2 + 32 + 2 * (r + 2);          // [r*r] dilated bit rows
  int *drun = reinterpret_cast<int *>(drow + r2);         // [r*r] dilated cells before the row
  int *rs = drun + r2;                                    // [r*r + 1] input cells before the row
  if (tid < r2) {
    const int run = off_d + incl_d - mine_d;
    if (tid % r == 0) { ps_o[tid / r] = off_o + incl_o - mine_o; ps_d[tid / r] = run; }
    drow[tid] = d;
    drun[tid] = run;
    rs[tid] = off_o + incl_o - mine_o;
  }
  if (tid == 0) rs[r2] = tot_o;
  __syncthreads();
  {   // ranks and list, one voxel per thread and step: coalesced index writes (a thread walking its own row wrote 32 separate lines)
    int *dl = dil_list + (size_t)bi * n_dil_max;
    int *di = dil_index + (size_t)bi * r3;
    for (int v = tid; v < r3; v += T) {
      const int row = v / r, z = v - row * r;
      const unsigned m = drow[row];
      const bool in = (m >> z) & 1u;
      const int rank = drun[row] + __popc(m & ((1u << z) - 1u));
      const bool ok = in && rank < n_dil_max;
      di[v] = ok ? rank : -1;
      if (ok) dl[rank] = v;
    }
  }
  if (class_count != nullptr) {
    // cells OUTSIDE the dilated set per boundary class (class = 9 cx + 3 cy + cz, c = 0 / 1 / 2: first plane, interior, last plane of
    // the axis): the convolution of a constant field takes one value per class there (bdm_conv3d_class_constants)
    __shared__ int s_cls[27];
    if (tid < 27) s_cls[tid] = 0;
    __syncthreads();
    if (tid < r2) {
      const int x = tid / r, y = tid % r;
      const int cx = x == 0 ? 0 : (x == r - 1 ? 2 : 1), cy = y == 0 ? 0 : (y == r - 1 ? 2 : 1);
      const unsigned out = ~d & full;
      const int lo = out & 1u, hi = (out >> (r - 1)) & 1u, mid = __popc(out) - lo - hi;
      if (lo) atomicAdd(&s_cls[(cx * 3 + cy) * 3 + 0], lo);
      if (mid) atomicAdd(&s_cls[(cx * 3 + cy) * 3 + 1], mid);
      if (hi) atomicAdd(&s_cls[(cx * 3 + cy) * 3 + 2], hi);
    }
    __syncthreads();
    if (tid < 27) class_count[(size_t)bi * 27 + tid] = s_cls[tid];
  }
  if (tid == 0) { ps_o[r] = ps_o[r + 1] = tot_o; ps_d[r] = ps_d[r + 1] = tot_d; }
  __syncthreads();
  for (int x = tid; x < r + 2; x += T) plane_start[(size_t)bi * (r + 2) + x] = ps_o[x];
  if (half_form) {
    // ---- half-tile form: flat tiles of `tile` entries; one that crosses x-planes whose rows (planes x0-1 .. x1+1) exceed xcap is split
    // at the plane boundaries (<= r - 1 extra tiles in total); a tile inside one plane lists three row ranges.  All in parallel: thread t
    // owns flat tile t, a workgroup scan numbers the pieces.
    __threadfence();
    __syncthreads();                                        // dil_list of this shape is visible to the whole workgroup
    const int *dl = dil_list + (size_t)bi * n_dil_max;
    const int nd = min(tot_d, n_dil_max), nt_flat = (nd + tile - 1) / tile;
    int pieces = 0, j = 0, jend = 0, x0 = 0, x1 = 0;
    if (tid < nt_flat) {
      j = tid * tile; jend = min(j + tile, nd);
      x0 = dl[j] / r2; x1 = dl[jend - 1] / r2;
      pieces = (x0 == x1 || rs[min(x1 + 2, r) * r] - rs[max(x0 - 1, 0) * r] <= xcap) ? 1 : x1 - x0 + 1;
    }
    int incl = pieces;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int tq = __shfl_up(incl, o, 64);
      if (lane >= o) incl += tq;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int off = 0, nt = 0;
    for (int w = 0; w < (T >> 6); ++w) {
      if (w < wave) off += wtot[w];
      nt += wtot[w];
    }
    nt = min(nt, tiles_max);
    off += incl - pieces;
    for (int k = 0; k < pieces; ++k) {
      const int t = off + k;
      if (t >= tiles_max) break;                           // (cannot happen: tiles_max = r^3 / tile + r)
      int a = j, e_ = jend, xa = x0, xb = x1;
      if (pieces > 1) { xa = xb = x0 + k; a = max(j, ps_d[xa]); e_ = min(jend, ps_d[xa + 1]); if (e_ < a) e_ = a; }
      int lo[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
      if (e_ > a) {
        if (xa == xb) {
          const int y0 = (dl[a] / r) % r, y1 = (dl[e_ - 1] / r) % r;
          for (int dx = 0; dx < 3; ++dx) {
            const int px = xa + dx - 1;
            if (px < 0 || px >= r) continue;
            lo[dx] = rs[px * r + max(y0 - 1, 0)];
            nn[dx] = rs[px * r + min(y1 + 1, r - 1) + 1] - lo[dx];
          }
        } else {
          lo[0] = rs[max(xa - 1, 0) * r];
          nn[0] = rs[min(xb + 2, r) * r] - lo[0];
        }
      }
      int *e = tile_start + ((size_t)bi * tiles_max + t) * TILE_REC;
      const int vf = t == 0 ? 0 : (a < nd ? dl[a] : r3), ve = (t + 1 < nt && e_ < nd) ? dl[e_] : r3;
      e[0] = a; e[1] = e_; e[2] = vf; e[3] = ve; e[4] = lo[0]; e[5] = nn[0]; e[6] = (xa << 8) | xb; e[7] = nt;
      e[8] = lo[1]; e[9] = nn[1]; e[10] = lo[2]; e[11] = nn[2]; e[12] = tile; e[13] = e[14] = e[15] = 0;
    }
    for (int t = nt + tid; t < tiles_max; t += T) {       // the dead records (readers take the list length from the LAST record's [1])
      int *e = tile_start + ((size_t)bi * tiles_max + t) * TILE_REC;
      const bool first_empty = t == 0;                     // a grid without an input cell: tile 0 owns everything (all bias)
      e[0] = nd; e[1] = nd; e[2] = first_empty ? 0 : r3; e[3] = r3; e[4] = 0; e[5] = 0; e[6] = 0; e[7] = nt;
      for (int q = 8; q < TILE_REC; ++q) e[q] = q == 12 ? tile : 0;
    }
    return;
  }
  int *tinfo = reinterpret_cast<int *>(occ);              // (the occupancy rows are dead) [tiles_max + 1][2]: first entry, first plane
  __shared__ int s_tiles, s_cut;
  // The tile table.  Without a cut every tile is simply [t * tile, (t + 1) * tile): thread t derives its tile's planes by itself
  // (binary search in the plane prefixes) and checks the row budget; only if SOME tile would exceed it -- a cloud with more than
  // xcap occupied cells in three neighbouring planes: never on the denoisers' levels -- thread 0 redoes the table with the serial
  // walk below, which cuts at plane boundaries.  The walk alone (a chain of ~6 dependent LDS reads per tile, 64 - 96 tiles, ONE
  // thread) was about half of this kernel's 32 us.
  if (tid == 0) s_cut = 0;
  __syncthreads();
  {
    const int nd = min(tot_d, n_dil_max), nt_flat = (nd + tile - 1) / tile;
    if (tid < nt_flat && tid < tiles_max) {
      const int j = tid * tile, jend = min(j + tile, nd);
      int lo = 0, hi = r;                    // plane of entry j: the largest x with ps_d[x] <= j  (ps_d[r] = tot_d > j)
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ps_d[mid] <= j) lo = mid; else hi = mid; }
      const int x0 = lo;
      lo = x0; hi = r;                       // plane of entry jend - 1
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ps_d[mid] <= jend - 1) lo = mid; else hi = mid; }
      const int x1 = lo;
      if (x1 > x0 && ps_o[min(x1 + 2, r)] - ps_o[max(x0 - 1, 0)] > xcap) s_cut = 1;
      tinfo[2 * tid] = j; tinfo[2 * tid + 1] = (x0 << 8) | x1;
    }
    if (tid == 0) { tinfo[2 * min(nt_flat, tiles_max)] = nd; s_tiles = min(nt_flat, tiles_max); }
  }
  __syncthreads();
  if (tid == 0 && s_cut) {   // the general case: a short serial walk over LDS-resident prefixes
    const int nd = min(tot_d, n_dil_max);
    int t = 0, j = 0, x0 = 0;
    while (j < nd && t < tiles_max) {
      while (ps_d[x0 + 1] <= j) ++x0;                     // plane of entry j
      int jend = min(j + tile, nd), x1 = x0;
      while (ps_d[x1 + 1] < jend) ++x1;                    // plane of entry jend - 1
      while (x1 > x0 && ps_o[min(x1 + 2, r)] - ps_o[max(x0 - 1, 0)] > xcap) { jend = ps_d[x1]; --x1; }   // cut at the plane boundary
      tinfo[2 * t] = j; tinfo[2 * t + 1] = (x0 << 8) | x1;
      ++t;
      j = jend;
    }
    tinfo[2 * t] = nd;                                     // (t == tiles_max with j < nd cannot happen: tiles_max = r^3 / tile + r)
    s_tiles = t;
  }
  __threadfence();
  __syncthreads();
  // tile_start[t][0..7] = first entry, end entry, first voxel of the linear range the tile owns, its end, first compact row of the
  // tile's input range, rows in it, (first x-plane << 8) | last x-plane of its voxels, live tiles of the shape: everything a workgroup
  // needs in ONE 32-byte read
  const int nt = s_tiles, nd = min(tot_d, n_dil_max);
  for (int t = tid; t < tiles_max; t += T) {
    int *e = tile_start + ((size_t)bi * tiles_max + t) * TILE_REC;
    const int *dl = dil_list + (size_t)bi * n_dil_max;
    int j0 = nd, jn = nd, vf = r3, ve = r3, klo = 0, nr = 0;
    if (t < nt) {
      j0 = tinfo[2 * t]; jn = tinfo[2 * t + 2];
      const int x0 = tinfo[2 * t + 1] >> 8, x1 = tinfo[2 * t + 1] & 255;
      vf = t == 0 ? 0 : dl[j0];                            // (written above by this workgroup; visible after the fence + barrier)
      ve = t + 1 < nt ? dl[jn] : r3;
      klo = ps_o[max(x0 - 1, 0)];
      nr = min(ps_o[min(x1 + 2, r)] - klo, xcap);
    } else if (t == 0) { vf = 0; ve = r3; }                // a grid without an occupied cell: tile 0 owns everything (all bias)
    e[0] = j0; e[1] = jn; e[2] = vf; e[3] = ve; e[4] = klo; e[5] = nr; e[6] = t < nt ? tinfo[2 * t + 1] : 0; e[7] = nt;
    for (int q = 8; q < TILE_REC; ++q) e[q] = 0;
  }
}

#ifdef BDM_EXPERIMENTAL
static bool half_tile_ok(int r, int tile) { return tile == 0 || ((r == 16 || r == 32) && (tile == 64 || tile == 128 || tile == 256)) || (r == 8 && (tile == 64 || tile == 128)); }
#else
static bool half_tile_ok(int, int tile) { return tile == 0; }   // the half-tile convolution is part of the EXPERIMENTAL=1 build
#endif

extern "C" int bdm_voxel_dilate_slices(int r, int tile) {
  int tl, xcap, tiles_max;
  sconv_dil_geometry(r, half_tile_ok(r, tile) ? tile : 0, &tl, &xcap, &tiles_max);
  return tiles_max;
}

static int voxel_dilate_launch(int b, int r, int n_dil_max, const int *src, int thr, int *dil_list, int *dil_index, int *plane_start,
                               int *tile_start, int *class_count, int half_tile, void *stream) {
  BDM_REQUIRE(b >= 0 && (r == 8 || r == 16 || r == 32) && n_dil_max >= 1 && src && dil_list && dil_index && plane_start && tile_start,
              "voxel_dilate: bad arguments (r in {8, 16, 32})");
  BDM_REQUIRE(half_tile_ok(r, half_tile), "voxel_dilate: tile %d unsupported (0 = full tiles; 64 / 128 / 256: EXPERIMENTAL=1 builds)", half_tile);
  if (b == 0) return BDM_OK;
  int tile, xcap, tiles_max;
  sconv_dil_geometry(r, half_tile, &tile, &xcap, &tiles_max);
  BDM_REQUIRE(cdiv(r * r * r, tile) <= 1024, "voxel_dilate: more flat tiles than threads");
  const size_t smem = sizeof(unsigned) * (4 * (size_t)r * r + 1 + 32 + 2 * (r + 2));
  hipLaunchKernelGGL(vox_dilate_kernel, dim3(b), dim3(1024), smem, (hipStream_t)stream, r, n_dil_max, tile, xcap, tiles_max,
                     half_tile > 0 ? 1 : 0, src, thr, dil_list, dil_index, plane_start, tile_start, class_count);
  return launch_status("voxel_dilate");
}

extern "C" int bdm_voxel_dilate(int b, int r, int n_dil_max, const int *cnt, int *dil_list, int *dil_index, int *plane_start,
                                int *tile_start, int tile, void *stream) {
  return voxel_dilate_launch(b, r, n_dil_max, cnt, 0, dil_list, dil_index, plane_start, tile_start, nullptr, tile, stream);
}

// The same for a set given by the RANKS of a previous list (cells with index >= 0): the dilation of the once-dilated set is where the
// SECOND convolution of a PVConv can differ from its per-class constants; its tiles' input ranges are rows of the FIRST list.
extern "C" int bdm_voxel_dilate_again(int b, int r, int n_dil_max, const int *dil_index_in, int *dil_list, int *dil_index,
                                      int *plane_start, int *tile_start, int *class_count, int tile, void *stream) {
  BDM_REQUIRE(class_count != nullptr, "voxel_dilate_again: class_count is NULL");
  return voxel_dilate_launch(b, r, n_dil_max, dil_index_in, -1, dil_list, dil_index, plane_start, tile_start, class_count, tile, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// the convolution
// ---------------------------------------------------------------------------------------------------------------------
#ifdef DIL_TIMING   // phase timestamps of every live workgroup (wall_clock64: 100 MHz), a debug build for tools/sparse_os_probe.py only
__device__ long long *g_dil_ts = nullptr;
extern "C" int bdm_debug_dil_timestamps(long long *buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_dil_ts), &buf, sizeof(buf)) == hipSuccess ? 0 : 1; }
// stamp 0 also leaves the CU the workgroup runs on in slot 7: HW_ID (wave / SIMD / CU / SH / SE) | XCC_ID << 32
#define DIL_STAMP(i) do { if (g_dil_ts && tid == 0) { g_dil_ts[(size_t)item * 8 + (i)] = wall_clock64();                                  \
    if ((i) == 0) g_dil_ts[(size_t)item * 8 + 7] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(63508) << 32); } } while (0)
#else
#define DIL_STAMP(i)
#endif

// H2IN = false: FIRST convolution -- rows are fp32 records of the occupied cells (scaled by the shape's power of two and split here),
//               an absent neighbour is zero.
// H2IN = true : SECOND convolution over the twice-dilated set -- rows are the (hi, lo) fp16 records of the once-dilated voxels
//               (bdm_group_norm_to_h2_rows: (b, C8, 2, n_max) records), a neighbour inside the grid but outside that set holds the
//               shape's CONSTANT record of the chunk (`xconst` (b, C8, 2): GroupNorm + Swish of the first convolution's bias), one
//               outside the grid is zero; `amax` is unused, the activation scale is x_inv_scale.  No closed-form bias share in
//               the GroupNorm partials: the voxels outside the list take per-class constants (bdm_conv3d_class_constants).
template <int MT, int NT, int NW, int R, bool H2IN>
__global__ __launch_bounds__(NW * 64) void sconv_dil_kernel(
    int C8, int Cout, int n_max, int n_dil_max, int xcap, const float4 *__restrict__ xr, const float *__restrict__ amax,
    const float4 *__restrict__ xconst, float x_inv_scale,
    const int *__restrict__ occ_index, const int *__restrict__ dil_list, const int *__restrict__ dil_index,
    const int *__restrict__ tile_start, const float4 *__restrict__ wq,
    const float *__restrict__ inv_scale, const float *__restrict__ bias, float *__restrict__ y, int compact, int gn_cg,
    double *__restrict__ gn_partial, int gn_slices, int nb_shapes, int ncb, int tiles_max, int *__restrict__ work_counter) {
  extern __shared__ __align__(16) float4 smem4[];
  constexpr int BM = 16 * MT, TILE = NT * NW * 16;   // a tile holds <= TILE voxels (sconv_dil_geometry)
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int NT_ = NW * 64;
  constexpr int WV = OS_PAIRS * 2 * 2 * BM, WI = (WV + NT_ - 1) / NT_;
  constexpr int NQ = OS_PAIRS / 2;
  constexpr int PF = MT * NT >= 16 ? 1 : 2;          // rows per thread whose records are register-prefetched one chunk ahead (the
                                                     // 64 x 64 wave tile has registers for one)
  constexpr int NBLK = NT * NW, NB = MT * 4 * 2;
  float4 *Ws = smem4;                 // [14][2][2][BM]
  float4 *Xs = smem4 + WV;            // [2][xcap + 128]: records xcap .. xcap + 63 of each split are zero -- absent neighbours are
  const int XS = xcap + 128;          // the common case (one copy per lane); xcap + 64 .. xcap + 127: the chunk's CONSTANT record (H2IN), per lane
  __shared__ int s_item;
  __shared__ float s_osc[64], s_obi[64];

  const int tid0 = threadIdx.x;
  const int G = gn_partial != nullptr ? Cout / gn_cg : 0, S = gn_slices;   // (H2IN: 27 more slices follow the tiles': the class constants')
  const int items = tiles_max * ncb * nb_shapes;
  // PERSISTENT workgroups (one per CU: a workgroup needs the CU's whole LDS) pull work items -- (tile, channel block, shape), tile
  // slowest, so the live tiles come first -- from a device-side counter.  With one workgroup per item the ~250 live tiles of a
  // B = 16 batch had to land on 256 CUs in ONE round; workgroups go round-robin over the 8 XCDs by index, a few XCDs got 33 of them
  // for their 32 CUs, and those stragglers started when the first round ended: the kernel took two workgroup lifetimes (measured
  // with wall-clock stamps, tools/sparse_dil_timeline.py: 159 us for an 84 us workgroup).  The counter is zero on entry
  // (caller) and is not reset: which workgroup computes a tile does not change any result.  (Round 5, measured and NOT done: a fixed
  // first item per workgroup, or fetching the next item while the matrix phase runs, hands LIVE tiles to workgroups that cannot start
  // them yet -- a late starter then holds one, or an early one reserves one behind its own: 96 -> 121 .. 154 us.)
  for (;;) {
    // the thread index is laundered once per item: everything derived from it (staging offsets, fragment addresses, ...) would
    // otherwise be hoisted out of the item loop and held in registers across it (+50 VGPRs: the 64 x 64 wave tile spilled)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
    __syncthreads();                  // the previous item's LDS (operand tiles, reduction scratch, s_item) is dead
    if (tid == 0) s_item = atomicAdd(work_counter, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= items) break;
    const int bi = item % nb_shapes, m0 = ((item / nb_shapes) % ncb) * BM, tile = item / (nb_shapes * ncb);
  const int4 *te = reinterpret_cast<const int4 *>(tile_start + ((size_t)bi * tiles_max + tile) * TILE_REC);
  const int4 ta = te[0], tb = te[1];
  const int j0 = ta.x, jn = ta.y, v_first = ta.z, v_end = ta.w, k_lo = tb.x;   // entries [j0, jn) of the list; linear range owned; input rows
  const int tiles_live = tb.w, tile_x0 = tb.z >> 8, tile_x1 = tb.z & 255;
  if (tile >= max(tiles_live, 1)) {   // nothing to compute here: an empty slice of the statistics
    if (gn_partial != nullptr) {
      const int ngt = BM / gn_cg;
      if (tid < ngt && m0 + tid * gn_cg < Cout) {
        double *dst = gn_partial + (((size_t)bi * G + m0 / gn_cg + tid) * S + tile) * 2;
        dst[0] = 0.0; dst[1] = 0.0;
      }
    }
    continue;
  }
  DIL_STAMP(0);
  const int *dl = dil_list + (size_t)bi * n_dil_max;
  const bool nothing = jn <= j0;                                // (a grid without an occupied cell: tile 0 is all bias)
  const int nrows = nothing ? 0 : tb.y;                         // compact rows [k_lo, k_lo + nrows): planes x0-1 .. x1+1 (<= xcap by construction)

  const float sx = H2IN ? 1.0f / x_inv_scale : act_scale_from_max(amax[bi]);
  // the tile's BM output channels: scale and bias once, in LDS (the epilogue's MT * NT (tile, block) pairs read them from there: the
  // 64 x 64 wave tile has no registers to spare, and per-use global loads made the epilogue the longest phase)
  if (tid < BM) {
    const int m = min(m0 + tid, Cout - 1);
    s_osc[tid] = inv_scale[m] * (1.0f / sx);
    s_obi[tid] = bias ? bias[m] : 0.f;
  }
  const float4 *wbase = Ws + ((kg >> 1) * 4 + (kg & 1)) * BM + l16;
  f32x4a acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[a][q] = f32x4a{0.f, 0.f, 0.f, 0.f};
  f32x4v_t cr = {0.f, 0.f, 0.f, 0.f};                                                            // (H2IN) this thread's copy of the constant record

  const float4 *xb = xr + (size_t)bi * C8 * n_max * 2;
  typedef f32x4v_t f32x4v;
  f32x4v xa[PF][2], wr[WI];
  auto load_chunk = [&](int c8) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // clamped rows re-read a valid record: never stored
      const int row = min(tid + u * NT_, max(nrows - 1, 0));
      if (H2IN) {
        const f32x4v *src = reinterpret_cast<const f32x4v *>(xr + (((size_t)bi * C8 + c8) * 2) * n_max + k_lo + row);
        xa[u][0] = src[0];
        xa[u][1] = src[n_max];
      } else {
        const f32x4v *src = reinterpret_cast<const f32x4v *>(xb + ((size_t)c8 * n_max + k_lo + row) * 2);
        xa[u][0] = src[0];
        xa[u][1] = src[1];
      }
    }
    if (H2IN && tid < 128) cr = *reinterpret_cast<const f32x4v *>(xconst + ((size_t)bi * C8 + c8) * 2 + (tid >> 6));
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * NT_;
      const int m = e % BM, psh = e / BM;
      const bool ok = e < WV && m0 + m < Cout;
      wr[i] = *reinterpret_cast<const f32x4v *>(&wq[ok ? (unsigned)((c8 * (OS_PAIRS * 4) + psh) * Cout + m0 + m) : 0u]);
    }
  };
  auto put = [&](int row, const float4 &p, const float4 &q) {
    if (H2IN) {   // already (hi, lo) records
      Xs[row] = p;
      Xs[XS + row] = q;
    } else {
      f16x8 hi, lo;
      split_record(p, q, sx, hi, lo);
      *reinterpret_cast<f16x8 *>(&Xs[row]) = hi;
      *reinterpret_cast<f16x8 *>(&Xs[XS + row]) = lo;
    }
  };
  auto store_chunk = [&](int c8) {
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (tid + u * NT_ < nrows) {
        const float4 p = make_float4(xa[u][0][0], xa[u][0][1], xa[u][0][2], xa[u][0][3]);
        const float4 q = make_float4(xa[u][1][0], xa[u][1][1], xa[u][1][2], xa[u][1][3]);
        put(tid + u * NT_, p, q);
      }
    for (int row0 = tid + PF * NT_; row0 < nrows; row0 += 2 * NT_) {   // ranges beyond PF * NT_ rows: unprefetched, two rows in flight
      const int row1 = row0 + NT_, rc1 = min(row1, nrows - 1);
      float4 a0, a1, b0, b1;
      if (H2IN) {
        const float4 *src = xr + (((size_t)bi * C8 + c8) * 2) * n_max + k_lo;
        a0 = src[row0]; a1 = src[n_max + row0]; b0 = src[rc1]; b1 = src[n_max + rc1];
      } else {
        const float4 *src = xb + ((size_t)c8 * n_max + k_lo) * 2;
        a0 = src[2 * row0]; a1 = src[2 * row0 + 1]; b0 = src[2 * rc1]; b1 = src[2 * rc1 + 1];
      }
      put(row0, a0, a1);
      if (row1 < nrows) put(row1, b0, b1);
    }
    if (H2IN && tid < 128) *reinterpret_cast<f32x4v *>(&Xs[(tid >> 6) * XS + xcap + 64 + (tid & 63)]) = cr;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int e = tid + i * NT_;
      if (e < WV) *reinterpret_cast<f32x4v *>(&Ws[e]) = wr[i];
    }
  };

  if (!nothing) load_chunk(0);   // chunk 0's operands are in flight while the neighbour records are looked up

  // ---- tile set-up: per-lane neighbour records, skip mask ----------------------------------------------------------------------
  int rec[NT][NQ];                    // LDS record of this lane's (voxel, tap) neighbour; xcap + lane = this lane's zero record
  unsigned amask = 0u;
  {
    // The index planes the tile's 27-neighbourhoods touch (x0 - 1 .. x1 + 1) are copied into the operand area of LDS first (coalesced,
    // the area is free until chunk 0 is stored) and the 28 look-ups per lane read LDS: as scattered 4-byte GLOBAL loads -- 64 different
    // lines per wave instruction -- they were 9 us of an 84 us tile (tools/sparse_dil_timeline.py), bound by the address unit.
    const int pl0 = max(tile_x0 - 1, 0), pl1 = min(tile_x1 + 1, R - 1), n_oi = (pl1 - pl0 + 1) * R2;
    const bool staged = !nothing && n_oi * (int)sizeof(int) <= 2 * XS * (int)sizeof(float4);   // (uniform; r = 32 tiles spanning > 25 planes: global)
    int *s_oi = reinterpret_cast<int *>(Xs);
    const int *oi = occ_index + (size_t)bi * R3;
    int vq[NT];                       // this lane's voxels: read BEFORE the planes are staged (one round trip for both, not two)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int j = j0 + (q * NW + wave) * 16 + l16;
      vq[q] = j < jn ? dl[j] : -1;
    }
    if (staged) {
      const int4 *src = reinterpret_cast<const int4 *>(oi + pl0 * R2);
      for (int e = tid; e < n_oi / 4; e += NT_) reinterpret_cast<int4 *>(s_oi)[e] = src[e];
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int v = vq[q];
      const int vx = v / R2, vy = (v / R) % R, vz = v % R;
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q) {
        const int t = 4 * Q + kg;
        int k = -1;
        bool inside = false;
        if (v >= 0 && t < 27) {
          const int gx = vx + t / 9 - 1, gy = vy + (t / 3) % 3 - 1, gz = vz + t % 3 - 1;
          inside = gx >= 0 && gx < R && gy >= 0 && gy < R && gz >= 0 && gz < R;
          if (inside) k = staged ? s_oi[((gx - pl0) * R + gy) * R + gz] : oi[(gx * R + gy) * R + gz];
        }
        const bool here = k >= k_lo && k - k_lo < nrows;       // (always, for a present neighbour: the range covers its plane)
        const bool live = here || (H2IN && inside);            // a constant neighbour contributes too
        if (__ballot(live) != 0ull) amask |= 1u << (Q * NT + q);
        rec[q][Q] = here ? k - k_lo : ((H2IN && inside) ? xcap + 64 + lane : xcap + lane);
      }
    }
    amask = __builtin_amdgcn_readfirstlane(amask);
  }
  DIL_STAMP(1);
  if (!nothing) {   // (uniform per workgroup)
    for (int c8 = 0; c8 < C8; ++c8) {
      __syncthreads();
      if (c8 == 0) DIL_STAMP(2);
      if (c8 == 0 && tid < 128) Xs[(tid >> 6) * XS + xcap + (tid & 63)] = make_float4(0.f, 0.f, 0.f, 0.f);   // the zero records of both splits
      store_chunk(c8);
      __syncthreads();
      if (c8 == 0) DIL_STAMP(3);
      if (c8 == 1) DIL_STAMP(4);
      if (c8 + 1 < C8) load_chunk(c8 + 1);
#pragma unroll
      for (int Q = 0; Q < NQ; ++Q) {
        const unsigned qm = (amask >> (Q * NT)) & ((1u << NT) - 1u);
        if (qm == 0u) continue;   // wave-uniform: no voxel of this wave has a present neighbour under this tap quad
        f16x8 fa[MT][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float4 t = wbase[Q * 8 * BM + s * 2 * BM + mt * 16];
            fa[mt][s] = *reinterpret_cast<const f16x8 *>(&t);
          }
        // block by block (an absent neighbour reads the lane's zero record; blocks are almost always live in a compact tile, so they
        // are not skipped one by one).  Reading all NT blocks' fragments first, as the dense kernel does, was measured no faster
        // and costs the 64 x 64 wave tile 24 registers it does not have
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          f16x8 fb[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const float4 t = Xs[s * XS + rec[q][Q]];
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

  // ---- epilogue: scale + bias, GroupNorm partials, output ----------------------------------------------------------------------
  DIL_STAMP(5);
  __syncthreads();                                          // the operand tiles are dead (LDS is reused below)
  float *red = reinterpret_cast<float *>(smem4);            // [NBLK][NB], then [NB]
  float *otile = red + (NBLK + 1) * NB;                     // dense form: [BM][TILE] results of the tile (fits: checked by the launcher)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int jl = (q * NW + wave) * 16 + l16;            // position in the tile
      const bool live = j0 + jl < jn;
      float bs = 0.f, bq = 0.f;
      f32x4a o = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + mt * 16 + 4 * kg + i;
        if (m < Cout && live) {
          const float v = acc[mt][q][i] * s_osc[mt * 16 + 4 * kg + i] + s_obi[mt * 16 + 4 * kg + i];
          o[i] = v;
          bs += v;
          bq = __builtin_fmaf(v, v, bq);
        }
      }
      if (live) {
        const int mb = m0 + mt * 16 + 4 * kg;
        if (compact) {   // one row of Cout floats per list entry: 16 bytes per lane, 64 contiguous bytes per voxel and instruction
          float *dst = y + ((size_t)bi * n_dil_max + j0 + jl) * Cout + mb;
          if (mb + 3 < Cout && (Cout & 3) == 0) *reinterpret_cast<f32x4a *>(dst) = o;
          else
            for (int i = 0; i < 4 && mb + i < Cout; ++i) dst[i] = o[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) otile[(mt * 16 + 4 * kg + i) * TILE + jl] = o[i];
        }
      }
      if (gn_partial != nullptr) {
        bs = row16_sum(bs);
        bq = row16_sum(bq);
        if (l16 == 0) {
          const int nb = q * NW + wave;
          red[nb * NB + (mt * 4 + kg) * 2 + 0] = bs;
          red[nb * NB + (mt * 4 + kg) * 2 + 1] = bq;
        }
      }
    }
  if (gn_partial != nullptr || !compact) __syncthreads();
  if (!compact) {
    // the linear voxel range this tile owns, one channel row per wave at a time, coalesced: computed voxels from LDS, bias elsewhere
    const int *di = dil_index + (size_t)bi * R3;
    float *yb = y + (size_t)bi * Cout * R3;
    for (int m = wave; m < BM; m += NW) {
      if (m0 + m >= Cout) break;
      const float bv = bias ? bias[m0 + m] : 0.f;
      float *row = yb + (size_t)(m0 + m) * R3;
      for (int v = v_first + lane; v < v_end; v += 64) {
        const int j = di[v];
        row[v] = j >= 0 ? otile[m * TILE + (j - j0)] : bv;
      }
    }
  }
  if (gn_partial != nullptr) {
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
      // the bias voxels of the owned range, in closed form
      const double nfill = H2IN ? 0.0 : (double)((v_end - v_first) - (jn - j0));
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
  DIL_STAMP(6);
  }   // work items
}


#ifdef BDM_EXPERIMENTAL   // measured NOT faster than the full-tile kernel (profiles/r06_sparse_dil_half_tiles.txt): EXPERIMENTAL=1 builds only
// ---------------------------------------------------------------------------------------------------------------------
// the convolution, half-tile form (round 6): TWO workgroups per CU
// ---------------------------------------------------------------------------------------------------------------------
// The full-tile kernel above owns the CU's whole LDS (57 KB of weights per 8-channel chunk + 3 r^2 input rows), so a CU runs ONE
// workgroup and nothing hides its dependent look-up chain at the start, its two barriers per chunk or its store burst at the end
// (18 of a tile's 80 us at 64 -> 64 channels, 32^3: profiles/r05_sparse_dil_timeline.txt; at C2 every CU holds exactly one tile).  Here
//   * a workgroup is FOUR waves (the same 64 x 64 wave tile: MT x NT fragments of 16 x 16) and owns <= 64 NT list entries x BM channels;
//   * a chunk's weights pass through LDS in TWO phases (tap quads 0-3: 32 KB at BM = 64, then quads 4-6 into the same slots), and the
//     input rows are cut to what the tile's voxels can reach -- three (x-plane, y-row range) ranges for a tile inside one plane, ONE
//     range of whole planes otherwise (bdm_voxel_dilate, half form) -- at most DILH_XCAP rows: 80 KB per workgroup, two per CU, eight
//     waves per CU as before, and one workgroup's barriers / look-ups / stores run under the other's matrix phase;
//   * the ranges of a tile never exceed DILH_XCAP rows: a tile across planes is only formed when its planes' rows fit, and a tile
//     of T entries inside plane x, y-rows y0 .. y1, reaches at most 3 T + 12 r input cells -- every input cell of planes x-1 .. x+1 in
//     rows y0+1 .. y1-1 dilates onto a listed voxel (y, z) of plane x that belongs to THIS tile, at most three cells per voxel, plus
//     four boundary rows of three planes -- 1152 for T = 256, r = 32 (tests/test_hip_dense.py checks the bound on dense slabs).
// Items are finer (256 / 128 / 64 entries), so the persistent workgroups (two per CU) balance a batch that does not fill the chip
// with full tiles (the small-batch tile rule: ops.dil_tile).
template <int MT, int NT, int R, bool H2IN>
__global__ __launch_bounds__(256, 2) void sconv_dilh_kernel(
    int C8, int Cout, int n_max, int n_dil_max, const float4 *__restrict__ xr, const float *__restrict__ amax,
    const float4 *__restrict__ xconst, float x_inv_scale,
    const int *__restrict__ occ_index, const int *__restrict__ dil_list, const int *__restrict__ dil_index,
    const int *__restrict__ tile_start, const float4 *__restrict__ wq,
    const float *__restrict__ inv_scale, const float *__restrict__ bias, float *__restrict__ y, int compact, int gn_cg,
    double *__restrict__ gn_partial, int gn_slices, int nb_shapes, int ncb, int tiles_max, int *__restrict__ work_counter) {
  extern __shared__ __align__(16) float4 smem4[];
  constexpr int NW = 4, NT_ = NW * 64;
  constexpr int BM = 16 * MT, TILE = NT * NW * 16;
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int NQ = OS_PAIRS / 2;                   // tap quads
  constexpr int PF = 3;                              // input rows per thread prefetched in registers one chunk ahead
  constexpr int NBLK = NT * NW, NB = MT * 4 * 2;
  constexpr int XCAP = DILH_XCAP, XS = XCAP + 128;   // records XCAP .. XCAP + 63 of each split: zero; XCAP + 64 .. + 127: the chunk's constant (H2IN)
  float4 *Xs = smem4;                 // [2][XS]
  __shared__ int s_item;
  __shared__ float s_osc[64], s_obi[64];

  const int tid0 = threadIdx.x;
  const int G = gn_partial != nullptr ? Cout / gn_cg : 0, S = gn_slices;
  const int items = tiles_max * ncb * nb_shapes;
  for (;;) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));     // (see the full-tile kernel: keeps per-item address arithmetic out of registers across items)
    const int lane = tid & 63, wave = tid >> 6, l16 = lane & 15, kg = lane >> 4;
    __syncthreads();
    if (tid == 0) s_item = atomicAdd(work_counter, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= items) break;
    const int bi = item % nb_shapes, m0 = ((item / nb_shapes) % ncb) * BM, tile = item / (nb_shapes * ncb);
    const int4 *te = reinterpret_cast<const int4 *>(tile_start + ((size_t)bi * tiles_max + tile) * TILE_REC);
    const int4 ta = te[0], tb = te[1], tc = te[2];
    const int j0 = ta.x, jn = ta.y, v_first = ta.z, v_end = ta.w;
    const int tiles_live = tb.w, tile_x0 = tb.z >> 8, tile_x1 = tb.z & 255;
    if (tile >= max(tiles_live, 1)) {   // an empty slice of the statistics
      if (gn_partial != nullptr) {
        const int ngt = BM / gn_cg;
        if (tid < ngt && m0 + tid * gn_cg < Cout) {
          double *dst = gn_partial + (((size_t)bi * G + m0 / gn_cg + tid) * S + tile) * 2;
          dst[0] = 0.0; dst[1] = 0.0;
        }
      }
      continue;
    }
    DIL_STAMP(0);
    const int *dl = dil_list + (size_t)bi * n_dil_max;
    const bool nothing = jn <= j0;
    const int lo0 = tb.x, n0 = nothing ? 0 : tb.y, lo1 = tc.x, n1 = nothing ? 0 : tc.y, lo2 = tc.z, n2 = nothing ? 0 : tc.w;
    const int n01 = n0 + n1, nrows = min(n01 + n2, XCAP);         // (<= XCAP by the plan's construction, see above; clamped for memory safety)
    auto grow = [&](int i) { return i < n0 ? lo0 + i : (i < n01 ? lo1 + (i - n0) : lo2 + (i - n01)); };   // logical row -> compact row

    const float sx = H2IN ? 1.0f / x_inv_scale : act_scale_from_max(amax[bi]);
    if (tid < BM) {
      const int m = min(m0 + tid, Cout - 1);
      s_osc[tid] = inv_scale[m] * (1.0f / sx);
      s_obi[tid] = bias ? bias[m] : 0.f;
    }
    // A fragments (weights) come STRAIGHT FROM GLOBAL MEMORY in fragment order: record (chunk c8, quad Q, split s, channel tile mt) of
    // this lane = wlane[((c8 * NQ + Q) * 8 + s * 2) * Cout + mt * 16] -- 16 lanes read 256 contiguous bytes.  The image is the same for
    // every wave of the chip (L2-resident, 0.9 MB at 64 -> 64 channels) and the waves of a CU read it nearly in step (L1).
    // Buffer loads: ONE lane offset (this lane's record inside a (quad, split) block) + a wave-uniform scalar offset per fragment --
    // no 64-bit per-lane address arithmetic; reads past the image (channel tiles beyond Cout) return zeros and are never stored.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(wq), 0, C8 * (OS_PAIRS * 4) * Cout * 16, 0x00020000);
    const int wlane = (((kg >> 1) * 4 + (kg & 1)) * Cout + m0 + l16) * 16;
    f32x4a acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[a][q] = f32x4a{0.f, 0.f, 0.f, 0.f};
    typedef f32x4v_t f32x4v;
    f32x4v cr = {0.f, 0.f, 0.f, 0.f};
    f32x4v xa[PF][2];
    const float4 *xb = xr + (size_t)bi * C8 * n_max * 2;

    auto load_x = [&](int c8) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {   // clamped rows re-read a valid record: never stored
        const int g = grow(min(tid + u * NT_, max(nrows - 1, 0)));
        if (H2IN) {
          const f32x4v *src = reinterpret_cast<const f32x4v *>(xr + (((size_t)bi * C8 + c8) * 2) * n_max + g);
          xa[u][0] = src[0];
          xa[u][1] = src[n_max];
        } else {
          const f32x4v *src = reinterpret_cast<const f32x4v *>(xb + ((size_t)c8 * n_max + g) * 2);
          xa[u][0] = src[0];
          xa[u][1] = src[1];
        }
      }
      if (H2IN && tid < 128) cr = *reinterpret_cast<const f32x4v *>(xconst + ((size_t)bi * C8 + c8) * 2 + (tid >> 6));
    };
    auto put = [&](int row, const float4 &p, const float4 &q) {
      if (H2IN) {
        Xs[row] = p;
        Xs[XS + row] = q;
      } else {
        f16x8 hi, lo;
        split_record(p, q, sx, hi, lo);
        *reinterpret_cast<f16x8 *>(&Xs[row]) = hi;
        *reinterpret_cast<f16x8 *>(&Xs[XS + row]) = lo;
      }
    };
    auto store_x = [&](int c8) {
      const int rows_w = nrows;
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (tid + u * NT_ < rows_w) {
          const float4 p = make_float4(xa[u][0][0], xa[u][0][1], xa[u][0][2], xa[u][0][3]);
          const float4 q = make_float4(xa[u][1][0], xa[u][1][1], xa[u][1][2], xa[u][1][3]);
          put(tid + u * NT_, p, q);
        }
      for (int row0 = tid + PF * NT_; row0 < rows_w; row0 += 2 * NT_) {   // beyond the prefetched rows: two rows in flight
        const int row1 = row0 + NT_, rc1 = min(row1, rows_w - 1);
        const int g0 = grow(row0), g1 = grow(rc1);
        float4 a0, a1, b0, b1;
        if (H2IN) {
          const float4 *src = xr + (((size_t)bi * C8 + c8) * 2) * n_max;
          a0 = src[g0]; a1 = src[n_max + g0]; b0 = src[g1]; b1 = src[n_max + g1];
        } else {
          const float4 *src = xb + ((size_t)c8 * n_max) * 2;
          a0 = src[2 * g0]; a1 = src[2 * g0 + 1]; b0 = src[2 * g1]; b1 = src[2 * g1 + 1];
        }
        put(row0, a0, a1);
        if (row1 < rows_w) put(row1, b0, b1);
      }
      if (H2IN && tid < 128) *reinterpret_cast<f32x4v *>(&Xs[(tid >> 6) * XS + XCAP + 64 + (tid & 63)]) = cr;
    };

    if (!nothing) load_x(0);   // chunk 0's rows are in flight while the neighbour records are looked up

    // ---- tile set-up: per-lane neighbour records, skip mask ------------------------------------------------------------------
    // rec: LDS record of this lane's (voxel, tap) neighbour; XCAP + lane = this lane's zero record, XCAP + 64 + lane = its copy of the
    // chunk's constant
    int rec[NT][NQ];
    unsigned amask = 0u;
    {
      const int pl0 = max(tile_x0 - 1, 0), pl1 = min(tile_x1 + 1, R - 1), n_oi = (pl1 - pl0 + 1) * R2;
      const bool staged = !nothing && n_oi * (int)sizeof(int) <= 2 * XS * (int)sizeof(float4);
      int *s_oi = reinterpret_cast<int *>(Xs);
      const int *oi = occ_index + (size_t)bi * R3;
      int vq[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const int j = j0 + (q * NW + wave) * 16 + l16;
        vq[q] = j < jn ? dl[j] : -1;
      }
      if (staged) {
        const int4 *src = reinterpret_cast<const int4 *>(oi + pl0 * R2);
        for (int e = tid; e < n_oi / 4; e += NT_) reinterpret_cast<int4 *>(s_oi)[e] = src[e];
        __syncthreads();
      }
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const int v = vq[q];
        const int vx = v / R2, vy = (v / R) % R, vz = v % R;
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q) {
          const int t = 4 * Q + kg;
          int k = -1;
          bool inside = false;
          if (v >= 0 && t < 27) {
            const int gx = vx + t / 9 - 1, gy = vy + (t / 3) % 3 - 1, gz = vz + t % 3 - 1;
            inside = gx >= 0 && gx < R && gy >= 0 && gy < R && gz >= 0 && gz < R;
            if (inside) k = staged ? s_oi[((gx - pl0) * R + gy) * R + gz] : oi[(gx * R + gy) * R + gz];
          }
          int row = -1;                                           // logical row of a present neighbour (its range covers it by construction)
          if (k >= 0) {
            if ((unsigned)(k - lo0) < (unsigned)n0) row = k - lo0;
            else if ((unsigned)(k - lo1) < (unsigned)n1) row = n0 + (k - lo1);
            else if ((unsigned)(k - lo2) < (unsigned)n2) row = n01 + (k - lo2);
          }
          const bool here = row >= 0 && row < XCAP;
          const bool cst = H2IN && inside && !here;
          if (__ballot(here || cst) != 0ull) amask |= 1u << (Q * NT + q);
          rec[q][Q] = here ? row : (cst ? XCAP + 64 + lane : XCAP + lane);
        }
      }
      amask = __builtin_amdgcn_readfirstlane(amask);
    }
    DIL_STAMP(1);

    if (!nothing) {
      f32x4v fa[MT][2], fn[MT][2];                           // this step's weight fragments / the next live step's, in flight
      auto load_a = [&](f32x4v (&dst)[MT][2], int c8, int Q) {
        const int so = ((c8 * NQ + Q) * 8) * Cout * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            dst[mt][s] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + mt * 256, so + s * 2 * Cout * 16, 0));
      };
      const unsigned live_q = [&] {                          // bit Q: some 16-voxel block of this wave has a neighbour under quad Q
        unsigned m = 0u;
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q)
          if ((amask >> (Q * NT)) & ((1u << NT) - 1u)) m |= 1u << Q;
        return m;
      }();
      int q_first = 0;
      while (q_first < NQ && !((live_q >> q_first) & 1u)) ++q_first;
      if (q_first < NQ) load_a(fa, 0, q_first);
      for (int c8 = 0; c8 < C8; ++c8) {
        __syncthreads();                                   // (A) the previous chunk's row reads are done; chunk c8's rows have landed
        if (c8 == 0) DIL_STAMP(2);
        if (c8 == 0 && tid < 128) Xs[(tid >> 6) * XS + XCAP + (tid & 63)] = make_float4(0.f, 0.f, 0.f, 0.f);
        store_x(c8);
        lds_barrier();                                     // (B)
        if (c8 == 0) DIL_STAMP(3);
        if (c8 == 1) DIL_STAMP(4);
        if (c8 + 1 < C8) load_x(c8 + 1);
#pragma unroll
        for (int Q = 0; Q < NQ; ++Q) {
          if (!((live_q >> Q) & 1u)) continue;             // wave-uniform: no voxel of this wave has a present neighbour under this tap quad
          {   // the next live quad's fragments (of this chunk, else the next chunk's first live quad)
            int nq = Q + 1;
            while (nq < NQ && !((live_q >> nq) & 1u)) ++nq;
            if (nq < NQ) load_a(fn, c8, nq);
            else if (c8 + 1 < C8) load_a(fn, c8 + 1, q_first);
          }
#pragma unroll
          for (int q = 0; q < NT; ++q) {
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
                acc[mt][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[mt][term == 0 ? 1 : 0]), fb[term == 1 ? 1 : 0],
                                                                    acc[mt][q], 0, 0, 0);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) fa[mt][s] = fn[mt][s];
        }
      }
    }

    // ---- epilogue: scale + bias, GroupNorm partials, output (as the full-tile kernel) -------------------------------------------
    DIL_STAMP(5);
    __syncthreads();
    float *red = reinterpret_cast<float *>(smem4);            // [NBLK][NB], then [NB]
    float *otile = red + (NBLK + 1) * NB;                     // dense form: [BM][TILE]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const int jl = (q * NW + wave) * 16 + l16;
        const bool live = j0 + jl < jn;
        float bs = 0.f, bq = 0.f;
        f32x4a o = f32x4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + mt * 16 + 4 * kg + i;
          if (m < Cout && live) {
            const float v = acc[mt][q][i] * s_osc[mt * 16 + 4 * kg + i] + s_obi[mt * 16 + 4 * kg + i];
            o[i] = v;
            bs += v;
            bq = __builtin_fmaf(v, v, bq);
          }
        }
        if (live) {
          const int mb = m0 + mt * 16 + 4 * kg;
          if (compact) {
            float *dst = y + ((size_t)bi * n_dil_max + j0 + jl) * Cout + mb;
            if (mb + 3 < Cout && (Cout & 3) == 0) *reinterpret_cast<f32x4a *>(dst) = o;
            else
              for (int i = 0; i < 4 && mb + i < Cout; ++i) dst[i] = o[i];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) otile[(mt * 16 + 4 * kg + i) * TILE + jl] = o[i];
          }
        }
        if (gn_partial != nullptr) {
          bs = row16_sum(bs);
          bq = row16_sum(bq);
          if (l16 == 0) {
            const int nb = q * NW + wave;
            red[nb * NB + (mt * 4 + kg) * 2 + 0] = bs;
            red[nb * NB + (mt * 4 + kg) * 2 + 1] = bq;
          }
        }
      }
    if (gn_partial != nullptr || !compact) __syncthreads();
    if (!compact) {
      const int *di = dil_index + (size_t)bi * R3;
      float *yb = y + (size_t)bi * Cout * R3;
      for (int m = wave; m < BM; m += NW) {
        if (m0 + m >= Cout) break;
        const float bv = bias ? bias[m0 + m] : 0.f;
        float *row = yb + (size_t)(m0 + m) * R3;
        for (int v = v_first + lane; v < v_end; v += 64) {
          const int j = di[v];
          row[v] = (j >= j0 && j < jn) ? otile[m * TILE + (j - j0)] : bv;
        }
      }
    }
    if (gn_partial != nullptr) {
      float *red2 = red + NBLK * NB;
      for (int e = tid; e < NB; e += NT_) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < NBLK; ++j) a += red[j * NB + e];
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
        const double nfill = H2IN ? 0.0 : (double)((v_end - v_first) - (jn - j0));
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
    DIL_STAMP(6);
  }   // work items
}

#endif   // BDM_EXPERIMENTAL

static int sconv_dil_launch(bool h2in, int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                            const void *xconst, float x_inv_scale, const int *in_index, const int *dil_list, const int *dil_index,
                            const int *tile_start, const void *packed_w, const float *inv_scale, const float *bias, float *y,
                            int compact, int gn_cg, double *gn_partial, int *slices_out, int half_tile, int *work_counter,
                            void *stream) {
  BDM_REQUIRE(b >= 0 && cin >= 1 && cout >= 1 && n_max >= 1 && n_dil_max >= 1 && xr && (h2in ? xconst != nullptr : amax != nullptr) &&
                  in_index && dil_list && dil_index && tile_start && inv_scale && y && work_counter,
              "sparse_conv_dil: bad arguments");
  if (r != 8 && r != 16 && r != 32) {
    set_error("sparse_conv_dil: resolution %d unsupported (8, 16, 32)", r);
    return BDM_ERR_UNSUPPORTED;
  }
  BDM_REQUIRE(half_tile_ok(r, half_tile), "sparse_conv_dil: tile %d unsupported (0 = full tiles; 64 / 128 / 256: EXPERIMENTAL=1 builds)", half_tile);
  if (b == 0) return BDM_OK;
  const int c8 = (cin + 7) / 8;
  const int mi = cout > 32 ? 2 : 1;                     // 64 or 32 output channels per workgroup
  int tile, xcap, tiles;
  sconv_dil_geometry(r, half_tile, &tile, &xcap, &tiles);
  const int bm = 32 * mi, nblk = tile / 16, nb = (bm / 16) * 8, ncb = cdiv(cout, bm);
  size_t smem = half_tile ? 16 * (2 * ((size_t)xcap + 128))                             // the row ranges (weights come from global memory)
                          : 16 * ((size_t)OS_PAIRS * 4 * bm + 2 * ((size_t)xcap + 128));
  const size_t smem_out = sizeof(float) * ((size_t)(nblk + 1) * nb + (compact ? 0 : (size_t)bm * tile));
  if (smem_out > smem) smem = smem_out;
  BDM_REQUIRE(!half_tile || smem + 1024 <= 81920, "sparse_conv_dil: the half-tile form needs two workgroups per CU (%zu bytes of LDS)", smem);
  // one persistent workgroup per CU (or per item, when there are fewer)
  static int cus[16] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 16 || cus[dev] == 0) {
    hipDeviceProp_t prop;
    const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    if (dev >= 0 && dev < 16) cus[dev] = n > 0 ? n : 256;
  }
  const int ncu = ((dev >= 0 && dev < 16) ? cus[dev] : 256) * (half_tile ? 2 : 1);   // persistent workgroups: one (two) per CU
  const long long items = (long long)tiles * ncb * b;
  dim3 grid((unsigned)(items < ncu ? items : ncu));
  hipStream_t s = (hipStream_t)stream;
  const int slices = h2in ? tiles + 27 : tiles;
  if (gn_partial != nullptr) {
    BDM_REQUIRE(gn_cg >= 4 && (gn_cg & (gn_cg - 1)) == 0 && bm % gn_cg == 0 && cout % gn_cg == 0,
                "sparse_conv_dil: GroupNorm statistics need a power-of-two channels-per-group dividing %d (got cg=%d)", bm, gn_cg);
    if (slices_out) *slices_out = slices;
  }
#define DIL_LAUNCH_(MT, NT, NW, R, H)                                                                                      \
  do {                                                                                                                     \
    BDM_ALLOW_LDS((sconv_dil_kernel<MT, NT, NW, R, H>), smem);                                                             \
    hipLaunchKernelGGL((sconv_dil_kernel<MT, NT, NW, R, H>), grid, dim3(NW * 64), smem, s, c8, cout, n_max, n_dil_max,     \
                       xcap, (const float4 *)xr, amax, (const float4 *)xconst, x_inv_scale, in_index, dil_list, dil_index, \
                       tile_start, (const float4 *)packed_w, inv_scale, bias, y, compact, gn_cg, gn_partial, slices, b,    \
                       ncb, tiles, work_counter);                                                                          \
  } while (0)
#define DIL_LAUNCH(MT, NT, NW, R) do { if (h2in) DIL_LAUNCH_(MT, NT, NW, R, true); else DIL_LAUNCH_(MT, NT, NW, R, false); } while (0)
#define DILH_LAUNCH_(MT, NT, R, H)                                                                                         \
  do {                                                                                                                     \
    BDM_ALLOW_LDS((sconv_dilh_kernel<MT, NT, R, H>), smem);                                                                \
    hipLaunchKernelGGL((sconv_dilh_kernel<MT, NT, R, H>), grid, dim3(256), smem, s, c8, cout, n_max, n_dil_max,            \
                       (const float4 *)xr, amax, (const float4 *)xconst, x_inv_scale, in_index, dil_list, dil_index,       \
                       tile_start, (const float4 *)packed_w, inv_scale, bias, y, compact, gn_cg, gn_partial, slices, b,    \
                       ncb, tiles, work_counter);                                                                          \
  } while (0)
#define DILH_LAUNCH(MT, NT, R) do { if (h2in) DILH_LAUNCH_(MT, NT, R, true); else DILH_LAUNCH_(MT, NT, R, false); } while (0)
#define DILH_PICK(NT, R) do { if (mi == 2) DILH_LAUNCH(4, NT, R); else DILH_LAUNCH(2, NT, R); } while (0)
#ifdef BDM_EXPERIMENTAL
  if (half_tile) {
    const int nt = half_tile / 64;
    if (r == 32) { if (nt == 4) DILH_PICK(4, 32); else if (nt == 2) DILH_PICK(2, 32); else DILH_PICK(1, 32); }
    else if (r == 16) { if (nt == 4) DILH_PICK(4, 16); else if (nt == 2) DILH_PICK(2, 16); else DILH_PICK(1, 16); }
    else { if (nt == 2) DILH_PICK(2, 8); else DILH_PICK(1, 8); }
    return launch_status("sparse_conv_dil (half tiles)");
  }
#endif
  if (r == 32) { if (mi == 2) DIL_LAUNCH(4, 4, 8, 32); else DIL_LAUNCH(2, 4, 8, 32); }
  else if (r == 16) { if (mi == 2) DIL_LAUNCH(4, 2, 8, 16); else DIL_LAUNCH(2, 2, 8, 16); }
  else { if (mi == 2) DIL_LAUNCH(4, 1, 8, 8); else DIL_LAUNCH(2, 1, 8, 8); }
#undef DIL_LAUNCH
#undef DIL_LAUNCH_
#undef DILH_PICK
#undef DILH_LAUNCH
#undef DILH_LAUNCH_
  return launch_status("sparse_conv_dil");
}

extern "C" int bdm_sparse_conv_dil(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                                   const int *occ_index, const int *dil_list, const int *dil_index, const int *tile_start,
                                   const void *packed_w, const float *inv_scale, const float *bias, float *y,
                                   int compact, int tile, int *work_counter, void *stream) {
  return sconv_dil_launch(false, b, cin, cout, r, n_max, n_dil_max, xr, amax, nullptr, 0.f, occ_index, dil_list, dil_index, tile_start,
                          packed_w, inv_scale, bias, y, compact, 0, nullptr, nullptr, tile, work_counter, stream);
}

extern "C" int bdm_sparse_conv_dil_gn(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                                      const int *occ_index, const int *dil_list, const int *dil_index, const int *tile_start,
                                      const void *packed_w, const float *inv_scale, const float *bias,
                                      float *y, int compact, int groups, void *gn_partial, int *slices_out, int tile,
                                      int *work_counter, void *stream) {
  BDM_REQUIRE(groups >= 1 && cout % groups == 0 && gn_partial != nullptr && slices_out != nullptr, "sparse_conv_dil_gn: bad arguments");
  return sconv_dil_launch(false, b, cin, cout, r, n_max, n_dil_max, xr, amax, nullptr, 0.f, occ_index, dil_list, dil_index, tile_start,
                          packed_w, inv_scale, bias, y, compact, cout / groups, (double *)gn_partial, slices_out, tile, work_counter, stream);
}

// SECOND convolution of a PVConv on the twice-dilated set (H2IN form of the kernel): rows_h2 / xconst / x_inv_scale from
// bdm_group_norm_to_h2_rows, in_index = the FIRST list's ranks (dil_index of bdm_voxel_dilate), lists / tiles of
// bdm_voxel_dilate_again; y (b, n_dil_max, cout) compact rows.  The partials cover the listed voxels only: slices =
// bdm_voxel_dilate_slices(r) + 27, the last 27 are written by bdm_conv3d_class_constants.
extern "C" int bdm_sparse_conv_dil_h2_gn(int b, int cin, int cout, int r, int n_rows_max, int n_dil_max, const void *rows_h2,
                                         const void *xconst, float x_inv_scale, const int *in_index, const int *dil_list,
                                         const int *dil_index, const int *tile_start, const void *packed_w, const float *inv_scale,
                                         const float *bias, float *y, int groups, void *gn_partial, int *slices_out, int tile,
                                         int *work_counter, void *stream) {
  BDM_REQUIRE(groups >= 1 && cout % groups == 0 && gn_partial != nullptr && slices_out != nullptr && x_inv_scale > 0.f,
              "sparse_conv_dil_h2_gn: bad arguments");
  return sconv_dil_launch(true, b, cin, cout, r, n_rows_max, n_dil_max, rows_h2, nullptr, xconst, x_inv_scale, in_index, dil_list,
                          dil_index, tile_start, packed_w, inv_scale, bias, y, 1, cout / groups, (double *)gn_partial, slices_out, tile,
                          work_counter, stream);
}
