// two_proc_repro.hip -- torch-free reproduction of the two-process difference.
//
//   hipcc --offload-arch=gfx950 -O2 -I include tools/coresidency/two_proc_repro.hip -o /tmp/two_proc_repro -L bdm_amd -l:libbdm_hip.so \
//         -Wl,-rpath,$PWD/bdm_amd
//   /tmp/two_proc_repro 300            # alone: every line must report 0 differing repetitions
//   /tmp/two_proc_repro 300 & /tmp/two_proc_repro 300   # two PROCESSES on one GPU at the same time
//   /tmp/two_proc_repro 300 - 1 & /tmp/two_proc_repro 300 - 2   # ... holding DIFFERENT data (seed argument)
//
// Every case launches ONE kernel (or one C-ABI call) REPS times on fixed inputs with a device-wide synchronisation after every
// launch, copies the output back and compares it bit by bit with the first repetition.  Cases:
//   library kernels through the C ABI (no torch, no second stream, plain hipMalloc memory):
//     pointwise_conv_gn (GroupNorm-folded 1x1 GEMM), pointwise_conv, devoxelize_gn_gate_add, conv3d fp16x3
//   trivial kernels defined in this file (nothing of the library):
//     copy        y = x, one float4 per thread, no LDS
//     swish       y = x * rcp(1 + exp2(-x log2 e)) per element (v_exp_f32 / v_rcp_f32)
//     lds_copy    float4 through LDS with a barrier
//     long_copy   grid-stride copy, few workgroups, long-running waves
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "bdm_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define ABI_OK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s: code %d: %s\n", #x, rc_, bdm_last_error()); exit(2); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd() {  // xorshift64*, uniform in [-1, 1)
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (float)((double)((rng_state * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float *dev_random(size_t n, float scale = 1.f, float shift = 0.f) {
  std::vector<float> h(n);
  for (auto &v : h) v = rnd() * scale + shift;
  float *d;
  HIP_OK(hipMalloc(&d, n * sizeof(float)));
  HIP_OK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
template <typename T> static T *dev_alloc(size_t n) {
  T *d;
  HIP_OK(hipMalloc(&d, n * sizeof(T)));
  HIP_OK(hipMemset(d, 0, n * sizeof(T)));
  return d;
}

__global__ void copy_kernel(const float4 *x, float4 *y, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) y[i] = x[i];
}
__global__ void swish_kernel(const float4 *x, float4 *y, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = x[i];
  auto sw = [](float t) { return t * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-t * 1.44269504088896340736f)); };
  y[i] = make_float4(sw(v.x), sw(v.y), sw(v.z), sw(v.w));
}
__global__ void lds_copy_kernel(const float4 *x, float4 *y, size_t n4) {
  __shared__ float4 tile[256];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  tile[threadIdx.x] = i < n4 ? x[i] : make_float4(0, 0, 0, 0);
  __syncthreads();
  if (i < n4) y[i] = tile[threadIdx.x ^ 1];
}
__global__ void long_copy_kernel(const float4 *x, float4 *y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}


// ---- probes: WHAT does a disturbed kernel lose?  Each probe checks one hardware resource of its own workgroup and counts
// violations into bad[0..2] (they need no reference run: any non-zero count is a fault).
__device__ __forceinline__ double slow_chain(double v, int n) {
  for (int k = 0; k < n; ++k) v = sqrt(v + 1.0);  // a long dependent fp64 chain (like the mean / rstd of a folded GroupNorm)
  return v;
}
// (0) barrier: ONE slow lane writes an LDS word, everybody reads it after __syncthreads(); an early release shows the old value
__global__ __launch_bounds__(256) void probe_barrier_kernel(int *bad, int iters) {
  __shared__ int flag;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  int mine = 0;
  for (int it = 1; it <= iters; ++it) {
    if (threadIdx.x == 255) flag = it + (slow_chain((double)it, 48) < 0.0 ? 1 : 0);
    __syncthreads();
    if (flag != it) ++mine;
    __syncthreads();
  }
  if (mine) atomicAdd(bad + 0, mine);
}
// (1) LDS contents: every thread fills its own words, waits, and checks its OWN words again (no barrier involved)
__global__ __launch_bounds__(256) void probe_lds_kernel(int *bad, int iters) {
  __shared__ unsigned buf[6144];  // 24 KB
  int mine = 0;
  for (int it = 1; it <= iters; ++it) {
    for (int i = threadIdx.x; i < 6144; i += 256) buf[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(it * 40503u + blockIdx.x);
    const double d = slow_chain((double)(it + threadIdx.x), 24);
    for (int i = threadIdx.x; i < 6144; i += 256)
      if (buf[i] != ((unsigned)(i * 2654435761u) ^ (unsigned)(it * 40503u + blockIdx.x))) ++mine;
    if (d < 0.0) buf[0] = 1u;
  }
  if (mine) atomicAdd(bad + 1, mine);
}
// (2) registers: 48 live values per lane across a long wait
__global__ __launch_bounds__(256) void probe_regs_kernel(int *bad, int iters, const float *x) {
  int mine = 0;
  for (int it = 1; it <= iters; ++it) {
    float r[48];
#pragma unroll
    for (int j = 0; j < 48; ++j) r[j] = x[(threadIdx.x * 48 + j + it) & 0xFFFF];
    const double d = slow_chain((double)(it + threadIdx.x), 24);
#pragma unroll
    for (int j = 0; j < 48; ++j)
      if (__float_as_uint(r[j]) != __float_as_uint(x[(threadIdx.x * 48 + j + it) & 0xFFFF])) ++mine;
    if (d < 0.0) ++mine;
  }
  if (mine) atomicAdd(bad + 2, mine);
}

// (3) LDS to the brim: workgroups whose dynamic LDS adds up to (nearly) the CU's whole 160 KB; each fills ALL of its allocation,
// waits, and verifies it.  No second kernel, no second process: does the top of the LDS hold?  log: up to 32 records of
// {HW_REG_LDS_ALLOC of the workgroup, first bad word, bad words, value found}.
__global__ __launch_bounds__(256) void probe_brim_kernel(int words, int iters, int *bad, unsigned *log) {
  extern __shared__ unsigned brim[];
  const unsigned alloc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6);  // HW_REG_LDS_ALLOC: base / size of this workgroup
  for (int it = 1; it <= iters; ++it) {
    for (int i = threadIdx.x; i < words; i += 256) brim[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(it * 40503u + blockIdx.x * 7u);
    __syncthreads();
    const double d = slow_chain((double)(it + threadIdx.x), 64);
    __syncthreads();
    int mine = 0, first = -1;
    unsigned seen = 0;
    for (int i = threadIdx.x; i < words; i += 256) {
      const unsigned v = brim[i];
      if (v != ((unsigned)(i * 2654435761u) ^ (unsigned)(it * 40503u + blockIdx.x * 7u))) { if (!mine) { first = i; seen = v; } ++mine; }
    }
    if (d < 0.0) brim[0] = 1u;
    if (mine) {
      const int slot = atomicAdd(bad + 3, 1);
      atomicAdd(bad + 4, mine);
      if (slot < 32) { log[slot * 4] = alloc; log[slot * 4 + 1] = (unsigned)first; log[slot * 4 + 2] = (unsigned)mine; log[slot * 4 + 3] = seen; }
    }
    __syncthreads();
  }
}

// trivial VICTIM with the access pattern of the global-memory devoxelisation and nothing else: eight scattered 4-byte loads per
// (point, channel) from a static grid, a weighted sum, one store.  No LDS, no transcendental, no parameter tables.
__global__ void gather8_kernel(int c, int n, int r, const float *__restrict__ coords, const float *__restrict__ grid, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, ci = blockIdx.y, bi = blockIdx.z;
  if (i >= n) return;
  const float *pc = coords + (size_t)bi * 3 * n;
  const float x = pc[i], y = pc[n + i], z = pc[2 * n + i];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z), x1 = x - xl, y1 = y - yl, z1 = z - zl;
  const int r2 = r * r, base = (int)xl * r2 + (int)yl * r + (int)zl, sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
  const float *g = grid + ((size_t)bi * c + ci) * r2 * r;
  float acc = (1 - x1) * (1 - y1) * (1 - z1) * g[base];
  acc += (1 - x1) * (1 - y1) * z1 * g[base + sz];
  acc += (1 - x1) * y1 * (1 - z1) * g[base + sy];
  acc += (1 - x1) * y1 * z1 * g[base + sy + sz];
  acc += x1 * (1 - y1) * (1 - z1) * g[base + sx];
  acc += x1 * (1 - y1) * z1 * g[base + sx + sz];
  acc += x1 * y1 * (1 - z1) * g[base + sx + sy];
  acc += x1 * y1 * z1 * g[base + sx + sy + sz];
  out[((size_t)bi * c + ci) * n + i] = acc;
}

struct Case {
  const char *name;
  std::function<void()> launch;
  const void *out;
  size_t bytes;
};

// trivial aggressor: streams data through a big dynamic-LDS allocation (the library's convolutions hold up to 135 KB per workgroup)
__global__ void big_lds_kernel(const float4 *x, float4 *y, size_t n4, int lds_items) {
  extern __shared__ float4 big[];
  for (int i = threadIdx.x; i < lds_items; i += blockDim.x) big[i] = x[((size_t)blockIdx.x * lds_items + i) % n4];
  __syncthreads();
  float4 acc = make_float4(0, 0, 0, 0);
  for (int i = threadIdx.x; i < lds_items; i += blockDim.x) { const float4 v = big[(i * 33) % lds_items]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  y[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// trivial aggressors with fp64 / fp32 division in every lane (v_rcp_f64 / v_div_* sequences; the library's sparse feature kernel
// computes 1.0 / (double)count per cell)
__global__ void f64div_kernel(const float *x, float *y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double v = (double)x[i] + 2.5;
    float acc = 0.f;
    for (int k = 0; k < 16; ++k) { acc += (float)(1.0 / v); v += 1.0; }
    y[i] = acc;
  }
}
__global__ void f32div_kernel(const float *x, float *y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = x[i] + 2.5f, acc = 0.f;
    for (int k = 0; k < 16; ++k) { acc += 1.0f / v; v += 1.0f; }
    y[i] = acc;
  }
}

// Trivial stand-ins for what distinguishes the library's sparse_gemm_s3_kernel (the kernel tools/coresidency/two_proc_aggressors2.sh named):
// 48 KB of static LDS, 256 threads, bf16 MFMA, and WHOLE WORKGROUPS THAT RETURN AT ONCE (rows beyond a shape's occupied count).
//   mode bit 0: most workgroups exit before touching anything      bit 1: use the matrix cores      bit 2: only 8 KB of LDS
typedef __attribute__((ext_vector_type(16))) float t_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 t_bf16x8;
template <int LDS_ITEMS>
__global__ __launch_bounds__(256) void lds_exit_kernel(int mode, const int *__restrict__ live, const uint4 *__restrict__ x, float *__restrict__ y) {
  __shared__ uint4 A[LDS_ITEMS], Bm[LDS_ITEMS];
  if ((mode & 1) && (int)blockIdx.y * 128 >= live[blockIdx.z]) return;  // as `if (m_count && m0 >= m_count[bi]) return;`
  const int tid = threadIdx.x;
  t_f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int it = 0; it < 8; ++it) {
    __syncthreads();
    for (int e = tid; e < LDS_ITEMS; e += 256) { A[e] = x[(blockIdx.x * 64 + it * 7 + e) & 0xFFFF]; Bm[e] = x[(blockIdx.y * 32 + it * 5 + e) & 0xFFFF]; }
    __syncthreads();
    const uint4 a = A[(tid * 5 + it) % LDS_ITEMS], b = Bm[(tid * 3 + it) % LDS_ITEMS];
    if (mode & 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const t_bf16x8 *>(&a), *reinterpret_cast<const t_bf16x8 *>(&b), acc, 0, 0, 0);
    else acc[0] += __uint_as_float((a.x ^ b.y) & 0x3F800000u);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  y[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * 256 + blockIdx.x * 256 + tid] = s;
}

// Ablation stand-in for sparse_gemm_s3_kernel: the same grid, LDS footprint, load / store patterns and MFMA count, each part behind a mode
// bit (1: global operand loads, 2: LDS staging + barriers, 4: MFMAs, 8: the output stores) -- which part disturbs the neighbour?
__global__ __launch_bounds__(256) void agg_gemm_kernel(int mode, int M, int G, int N, const uint4 *__restrict__ A, const uint4 *__restrict__ Bw,
                                                       float *__restrict__ Y) {
  constexpr int BM = 128, BN = 128, AI = 12 * BM / 256, BI = 12 * BN / 256;
  __shared__ uint4 As[12 * BM], Bs[12 * BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5, wr = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, bi = blockIdx.z;
  const uint4 *Ab = A + (size_t)bi * G * 3 * M;
  t_f32x16 acc[2][2];
  for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 16; ++r) acc[x][y][r] = (float)(tid + r);
  uint4 ar[AI], br[BI];
  for (int i = 0; i < AI; ++i) ar[i] = make_uint4(tid, i, 1u, 2u);
  for (int i = 0; i < BI; ++i) br[i] = make_uint4(tid, i, 3u, 4u);
  for (int g0 = 0; g0 < G; g0 += 4) {
    if (mode & 1) {
      for (int i = 0; i < AI; ++i) { const int e = tid + i * 256, row = e % BM, gs = e / BM, g = min(g0 + gs / 3, G - 1), sp = gs % 3; ar[i] = Ab[(unsigned)((g * 3 + sp) * M + min(m0 + row, M - 1))]; }
      for (int i = 0; i < BI; ++i) { const int e = tid + i * 256, col = e % BN, gs = e / BN, g = min(g0 + gs / 3, G - 1), sp = gs % 3; br[i] = Bw[(unsigned)((g * 3 + sp) * N + min(n0 + col, N - 1))]; }
    }
    if (mode & 2) {
      __syncthreads();
      for (int i = 0; i < AI; ++i) As[tid + i * 256] = ar[i];
      for (int i = 0; i < BI; ++i) Bs[tid + i * 256] = br[i];
      __syncthreads();
    }
    if (mode & 4) {
      for (int kk = 0; kk < 2; ++kk)
        for (int x = 0; x < 2; ++x)
          for (int y = 0; y < 2; ++y) {
            const uint4 a = (mode & 2) ? As[((2 * kk + lh) * 3) * BM + (wr * 2 + x) * 32 + li] : ar[(kk + x) % AI];
            const uint4 b = (mode & 2) ? Bs[((2 * kk + lh) * 3) * BN + (wc * 2 + y) * 32 + li] : br[(kk + y) % BI];
            for (int t = 0; t < 6; ++t)
              acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const t_bf16x8 *>(&a), *reinterpret_cast<const t_bf16x8 *>(&b), acc[x][y], 0, 0, 0);
          }
    }
  }
  if (mode & 8) {
    float *Yb = Y + (size_t)bi * M * N;
    for (int x = 0; x < 2; ++x)
      for (int y = 0; y < 2; ++y) {
        const int nn = n0 + (wc * 2 + y) * 32 + li;
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + (wr * 2 + x) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < M && nn < N) Yb[(size_t)m * N + nn] = acc[x][y][r];
        }
      }
  } else if (acc[0][0][0] == 12345.678f) Y[tid] = acc[1][1][3];  // keep the work alive
}

// The library's sparse first convolution, kernel by kernel (torch-free set-up of a voxel plan for B = 2 clouds of 1024 points, 16^3)
static hipStream_t g_agg_stream = 0;  // stream of the aggressor's launches (null stream unless --inproc)
static volatile bool g_stop = false;  // --inproc: the victim side is done, the aggressor thread may leave before its time is up

struct SparseSetup {
  int B = 2, n = 1024, r = 16, C = 64, cout = 64, n_max = 1024;
  float *feat, *y, *out;
  int *cnt, *occ_index, *occ_list, *n_occ;
  unsigned char *rowocc;
  void *ws, *xs, *wpk;
  SparseSetup() {
    const int r3 = r * r * r;
    float *coords = dev_random((size_t)B * 3 * n, 0.3f), *norm = dev_alloc<float>((size_t)B * 3 * n);
    int *vox = dev_alloc<int>((size_t)B * 3 * n), *ind = dev_alloc<int>((size_t)B * n);
    cnt = dev_alloc<int>((size_t)B * r3); occ_index = dev_alloc<int>((size_t)B * r3); occ_list = dev_alloc<int>((size_t)B * n_max);
    n_occ = dev_alloc<int>(B); rowocc = dev_alloc<unsigned char>((size_t)B * r * r);
    ws = dev_alloc<unsigned char>(bdm_voxelize_workspace_bytes(B, n, r));
    ABI_OK(bdm_voxel_coords(B, n, r, 0.f, coords, norm, vox, nullptr));
    ABI_OK(bdm_voxelize_plan_full(B, n, r, n_max, vox, ind, cnt, ws, occ_index, occ_list, n_occ, rowocc, nullptr));
    feat = dev_random((size_t)B * C * n);
    xs = dev_alloc<unsigned char>((size_t)B * (C / 8) * 3 * n_max * 16);
    float *w = dev_random((size_t)cout * C * 27, 0.05f);
    wpk = dev_alloc<unsigned short>(bdm_sparse_conv_s3_weight_elems(cout, C));
    ABI_OK(bdm_sparse_conv_pack_weights_s3(cout, C, w, wpk, nullptr));
    y = dev_alloc<float>((size_t)B * n_max * 27 * cout);
    out = dev_alloc<float>((size_t)B * cout * r3);
    features(); gemm(); gather();
    xr = dev_alloc<float>((size_t)B * (C / 8) * n_max * 8); amax = dev_alloc<float>(B);
    wh2 = dev_alloc<unsigned short>(bdm_conv3d_h2_weight_elems(cout, C)); inv_scale = dev_alloc<float>(cout);
    float *scale_ws = dev_alloc<float>(cout);
    ABI_OK(bdm_conv3d_h2_pack_weights(cout, C, w, wh2, scale_ws, inv_scale, nullptr));
    ABI_OK(bdm_sparse_voxel_features_f32(B, C, n, r, n_max, feat, (long long)C * n, n, cnt, ws, occ_list, n_occ, xr, amax, nullptr));
    const int tiles = bdm_voxel_dilate_slices(r);
    dil_list = dev_alloc<int>((size_t)B * r3); dil_index = dev_alloc<int>((size_t)B * r3);
    int *plane_start = dev_alloc<int>((size_t)B * (r + 2));
    tile_start = dev_alloc<int>((size_t)B * tiles * 8); counter = dev_alloc<int>(1);
    yc = dev_alloc<float>((size_t)B * r3 * cout);
    ABI_OK(bdm_voxel_dilate(B, r, r3, cnt, dil_list, dil_index, plane_start, tile_start, nullptr));
    conv_os();
    HIP_OK(hipDeviceSynchronize());
  }
  void features() { ABI_OK(bdm_sparse_voxel_features_s3(B, C, n, r, n_max, feat, (long long)C * n, n, cnt, ws, occ_list, n_occ, xs, (void *)g_agg_stream)); }
  void gemm() { ABI_OK(bdm_sparse_conv_gemm_s3(B, n_max, C, 27 * cout, xs, wpk, n_occ, y, (void *)g_agg_stream)); }
  void gather() { ABI_OK(bdm_sparse_conv_gather(B, cout, r, n_max, y, occ_index, rowocc, nullptr, out, (void *)g_agg_stream)); }
  float *xr, *amax, *inv_scale, *yc;
  int *dil_list, *dil_index, *tile_start, *counter;
  void *wh2;
  // round 4: the compact output-stationary first convolution (sparse_conv_os.hip) -- the kernel that replaced GEMM + gather at the
  // 32^3 / 16^3 levels of the default forward (persistent workgroups: its work counter is zeroed before every launch)
  void conv_os() {
    HIP_OK(hipMemsetAsync(counter, 0, sizeof(int), g_agg_stream));
    ABI_OK(bdm_sparse_conv_dil(B, C, cout, r, n_max, r * r * r, xr, amax, occ_index, dil_list, dil_index, tile_start, wh2, inv_scale, nullptr, yc, 1,
                               counter, (void *)g_agg_stream));
  }
};

// usage: two_proc_repro --aggress <kind> <seconds>   kind: lds128k | lds32k | copy | f64div | f32div | features | gemm_s3 | gather |
//        exit48k (48 KB LDS, most workgroups return at once) | noexit48k | exit48k_mfma | noexit48k_mfma | exit8k
static int aggress(const char *kind, double seconds) {
  const size_t n4 = (size_t)1 << 22;
  float *x = dev_random(n4 * 4), *y = dev_alloc<float>(n4 * 4);
  const int bytes = !strcmp(kind, "lds128k") ? 128 * 1024 : (!strcmp(kind, "lds32k") ? 32 * 1024 : 0);
  if (bytes > 48 * 1024) HIP_OK(hipFuncSetAttribute((const void *)big_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  int *live = dev_alloc<int>(2);
  { const int h[2] = {150, 90}; HIP_OK(hipMemcpy(live, h, sizeof(h), hipMemcpyHostToDevice)); }
  const int exit_mode = !strcmp(kind, "exit48k") ? 1 : !strcmp(kind, "noexit48k") ? 0 : !strcmp(kind, "exit48k_mfma") ? 3 :
                        !strcmp(kind, "noexit48k_mfma") ? 2 : !strcmp(kind, "exit8k") ? 5 : -1;
  SparseSetup *sp = (!strcmp(kind, "features") || !strcmp(kind, "gemm_s3") || !strcmp(kind, "gemm_s3_all") || !strcmp(kind, "gather") || !strcmp(kind, "conv_os")) ? new SparseSetup() : nullptr;
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventRecord(e0, g_agg_stream));
  long iters = 0;
  for (;;) {
    for (int i = 0; i < 50; ++i) {
      if (bytes) hipLaunchKernelGGL(big_lds_kernel, dim3(2048), dim3(512), bytes, 0, (const float4 *)x, (float4 *)y, n4, bytes / 16);
      else if (exit_mode >= 0 && (exit_mode & 4)) hipLaunchKernelGGL(lds_exit_kernel<256>, dim3(14, 8, 2), dim3(256), 0, g_agg_stream, exit_mode, live, (const uint4 *)x, y);
      else if (exit_mode >= 0) hipLaunchKernelGGL(lds_exit_kernel<1536>, dim3(14, 8, 2), dim3(256), 0, g_agg_stream, exit_mode, live, (const uint4 *)x, y);
      else if (!strcmp(kind, "f64div")) hipLaunchKernelGGL(f64div_kernel, dim3(4096), dim3(256), 0, g_agg_stream, x, y, n4 * 4);
      else if (!strcmp(kind, "f32div")) hipLaunchKernelGGL(f32div_kernel, dim3(4096), dim3(256), 0, g_agg_stream, x, y, n4 * 4);
      else if (!strcmp(kind, "features")) sp->features();
      else if (!strcmp(kind, "gemm_s3")) sp->gemm();
      else if (!strcmp(kind, "gather")) sp->gather();
      else if (!strcmp(kind, "conv_os")) sp->conv_os();
      else if (!strcmp(kind, "pw")) {  // the library's fp32-MFMA 1x1 GEMM (64 -> 64 channels, 2 x 32768 columns): another LDS + MFMA kernel
        static float *pw_w = dev_random(64 * 64, 0.15f), *pw_b = dev_random(64, 0.1f);
        ABI_OK(bdm_pointwise_conv(2, 64, 64, 32768, pw_w, 64, x, 64ll * 32768, 32768, pw_b, nullptr, 0, nullptr, 0, 0, y, 64ll * 32768, 32768, 0, 0.f,
                                  (void *)g_agg_stream));
      }
      else if (!strncmp(kind, "agg", 3) && kind[3] >= '0' && kind[3] <= '9') {  // agg<mode>: the ablation stand-in above
        static SparseSetup *sq = new SparseSetup();
        hipLaunchKernelGGL(agg_gemm_kernel, dim3(14, 8, 2), dim3(256), 0, g_agg_stream, atoi(kind + 3), sq->n_max, sq->C / 8, 27 * sq->cout,
                           (const uint4 *)sq->xs, (const uint4 *)sq->wpk, sq->y);
      }
      else if (!strcmp(kind, "gemm_s3_all")) {  // the same GEMM with every row block live (no early exit)
        ABI_OK(bdm_sparse_conv_gemm_s3(sp->B, sp->n_max, sp->C, 27 * sp->cout, sp->xs, sp->wpk, nullptr, sp->y, (void *)g_agg_stream));
      }
      else hipLaunchKernelGGL(copy_kernel, dim3((n4 + 255) / 256), dim3(256), 0, g_agg_stream, (const float4 *)x, (float4 *)y, n4);
    }
    iters += 50;
    HIP_OK(hipEventRecord(e1, g_agg_stream)); HIP_OK(hipEventSynchronize(e1));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    if (ms > seconds * 1e3 || g_stop) break;
  }
  printf("aggressor %s: %ld launches\n", kind, iters);
  return 0;
}

static int aggress_on_stream(const char *kind, double seconds) {
  HIP_OK(hipStreamCreateWithFlags(&g_agg_stream, hipStreamNonBlocking));
  return aggress(kind, seconds);
}

int main(int argc, char **argv) {
  if (argc > 3 && !strcmp(argv[1], "--aggress")) return aggress(argv[2], atof(argv[3]));
  // --inproc <kind> <seconds> [victim args]: the aggressor runs in THIS process (host thread; the null stream does not serialise with
  // it: the victim cases then use a non-blocking stream) -- does the disturbance need a second PROCESS at all?
  std::thread *bg = nullptr;
  if (argc > 3 && !strcmp(argv[1], "--inproc")) {
    static std::string kind = argv[2];
    static double secs = atof(argv[3]);
    bg = new std::thread([] { HIP_OK(hipSetDevice(0)); aggress_on_stream(kind.c_str(), secs); });
    argc -= 3; argv += 3;
  }
  const int reps = argc > 1 ? atoi(argv[1]) : 200;
  const char *only = argc > 2 && strcmp(argv[2], "-") ? argv[2] : nullptr;
  // a different seed per process makes data of the OTHER process recognisable: with equal seeds both processes hold the same
  // bytes at the same virtual addresses, and a read that was served with the neighbour's data would go unnoticed
  if (argc > 3) rng_state ^= 0xD1B54A32D192ED03ull * (uint64_t)atoll(argv[3]);
  std::vector<Case> cases;

  // ---- library: SharedMLP pair at the SA0 shape of the denoiser (B=2, 32 -> 64 -> 64 channels, 1024 x 32 columns)
  const int B = 2, K0 = 32, M = 64, N = 1024 * 32, G = 8;
  float *x0 = dev_random((size_t)B * K0 * N), *w1 = dev_random((size_t)M * K0, 0.2f), *b1 = dev_random(M, 0.1f);
  float *w2 = dev_random((size_t)M * M, 0.15f), *b2 = dev_random(M, 0.1f), *gamma = dev_random(M, 0.2f, 1.f), *beta = dev_random(M, 0.1f);
  float *y1 = dev_alloc<float>((size_t)B * M * N), *y2 = dev_alloc<float>((size_t)B * M * N), *y3 = dev_alloc<float>((size_t)B * M * N);
  const int s1 = bdm_pointwise_conv_gn_slices(B, M, K0, N, G), s2 = bdm_pointwise_conv_gn_slices(B, M, M, N, G);
  double *p1 = dev_alloc<double>((size_t)B * G * s1 * 2), *p2 = dev_alloc<double>((size_t)B * G * s2 * 2);
  ABI_OK(bdm_pointwise_conv_gn(B, M, K0, N, w1, K0, x0, (long long)K0 * N, N, nullptr, 0, 0, 0, b1, y1, (long long)M * N, N, nullptr, 0, 0,
                               nullptr, nullptr, 0.f, G, p1, nullptr, 0, nullptr));
  HIP_OK(hipDeviceSynchronize());
  cases.push_back({"lib pointwise_conv_gn: statistics epilogue only (32 -> 64, 32768 columns)", [&] {
    ABI_OK(bdm_pointwise_conv_gn(B, M, K0, N, w1, K0, x0, (long long)K0 * N, N, nullptr, 0, 0, 0, b1, y3, (long long)M * N, N, nullptr, 0, 0,
                                 nullptr, nullptr, 0.f, G, p2, nullptr, 0, nullptr)); }, y3, (size_t)B * M * N * 4});
  cases.push_back({"lib pointwise_conv_gn: folded GroupNorm input + statistics (64 -> 64, 32768 columns)", [&] {
    ABI_OK(bdm_pointwise_conv_gn(B, M, M, N, w2, M, y1, (long long)M * N, N, nullptr, 0, 0, 0, b2, y2, (long long)M * N, N, p1, s1, G, gamma,
                                 beta, 1e-5f, G, p2, nullptr, 0, nullptr)); }, y2, (size_t)B * M * N * 4});
  cases.push_back({"lib pointwise_conv (64 -> 64, 32768 columns)", [&] {
    ABI_OK(bdm_pointwise_conv(B, M, M, N, w2, M, y1, (long long)M * N, N, b2, nullptr, 0, nullptr, 0, 0, y3, (long long)M * N, N, 0, 0.f,
                              nullptr)); }, y3, (size_t)B * M * N * 4});
  // ---- library: devoxelisation with folded GroupNorm (B=2, 128 channels, 1024 points, 16^3)
  const int C = 128, NP = 1024, R = 16;
  float *coords = dev_random((size_t)B * 3 * NP, 7.4f, 7.5f);  // voxel-grid coordinates in [0.1, 14.9]
  float *grid = dev_random((size_t)B * C * R * R * R), *coef = dev_random((size_t)B * C * 2, 0.3f, 0.8f), *gate = dev_random((size_t)B * C, 0.4f, 0.5f);
  float *add = dev_random((size_t)B * C * NP), *dout = dev_alloc<float>((size_t)B * C * NP);
  cases.push_back({"lib devoxelize_gn_gate_add (128 channels, 1024 points, 16^3)", [&] {
    ABI_OK(bdm_devoxelize_gn_gate_add(B, C, NP, R, coords, grid, coef, gate, add, (long long)C * NP, NP, dout, (long long)C * NP, NP, nullptr)); },
    dout, (size_t)B * C * NP * 4});
  // snapshots of the victim's static inputs: are they still intact at the end (an out-of-bounds WRITE of a neighbour would change them)?
  std::vector<uint32_t> grid0((size_t)B * C * R * R * R), coords0((size_t)B * 3 * NP);
  HIP_OK(hipMemcpy(grid0.data(), grid, grid0.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(coords0.data(), coords, coords0.size() * 4, hipMemcpyDeviceToHost));
  cases.push_back({"trivial gather8 (devoxelisation's 8 scattered loads per output, no LDS, no parameters)", [&] {
    hipLaunchKernelGGL(gather8_kernel, dim3((NP + 255) / 256, C, B), dim3(256), 0, 0, C, NP, R, (const float *)coords, (const float *)grid, dout); },
    dout, (size_t)B * C * NP * 4});
  // ---- library: the three calls back to back, 40 rounds without any synchronisation in between (the shape of a real forward)
  cases.push_back({"lib chain: 40 x [gemm+stats -> folded gemm+stats -> devoxelize] without synchronisation", [&] {
    for (int it = 0; it < 40; ++it) {
      ABI_OK(bdm_pointwise_conv_gn(B, M, K0, N, w1, K0, x0, (long long)K0 * N, N, nullptr, 0, 0, 0, b1, y1, (long long)M * N, N, nullptr, 0, 0,
                                   nullptr, nullptr, 0.f, G, p1, nullptr, 0, nullptr));
      ABI_OK(bdm_pointwise_conv_gn(B, M, M, N, w2, M, y1, (long long)M * N, N, nullptr, 0, 0, 0, b2, y2, (long long)M * N, N, p1, s1, G, gamma,
                                   beta, 1e-5f, G, p2, nullptr, 0, nullptr));
      ABI_OK(bdm_devoxelize_gn_gate_add(B, C, NP, R, coords, grid, coef, gate, add, (long long)C * NP, NP, dout, (long long)C * NP, NP, nullptr));
    } }, y2, (size_t)B * M * N * 4});
  // ---- library: what the SIDE STREAMS of a forward run beside the main stream's sparse GEMMs: the sampler chain
  //      (FPS, ball query, 3-NN search, voxel plan at 32^3) at the bench's batch
  {
    const int SB = 16, SN = 4096, SM = 1024, SU = 32;
    float *sc = dev_random((size_t)SB * 3 * SN, 0.5f), *cen = dev_alloc<float>((size_t)SB * 3 * SM);
    int *fidx = dev_alloc<int>((size_t)SB * SM), *nbr = dev_alloc<int>((size_t)SB * SM * SU);
    ABI_OK(bdm_furthest_point_sampling(SB, SN, SM, sc, fidx, cen, nullptr));
    HIP_OK(hipDeviceSynchronize());
    cases.push_back({"lib sampler: furthest_point_sampling (16 x 4096 -> 1024)", [=] {
      ABI_OK(bdm_furthest_point_sampling(SB, SN, SM, sc, fidx, nullptr, nullptr)); }, fidx, (size_t)SB * SM * 4});
    cases.push_back({"lib sampler: ball_query (16 x 1024 centres, 4096 points, r = 0.1, 32 neighbours)", [=] {
      ABI_OK(bdm_ball_query(SB, SN, SM, 0.1f, SU, cen, sc, nbr, nullptr)); }, nbr, (size_t)SB * SM * SU * 4});
    int *nn_i = dev_alloc<int>((size_t)SB * 3 * SN);
    float *nn_w = dev_alloc<float>((size_t)SB * 3 * SN);
    cases.push_back({"lib sampler: three_nn_search (16 x 4096 points, 1024 centres)", [=] {
      ABI_OK(bdm_three_nn_search(SB, SM, SN, sc, cen, nn_i, nn_w, nullptr)); }, nn_w, (size_t)SB * 3 * SN * 4});
    const int PR = 32, pr3 = PR * PR * PR;
    float *pnorm = dev_alloc<float>((size_t)SB * 3 * SN);
    int *pvox = dev_alloc<int>((size_t)SB * 3 * SN), *pind = dev_alloc<int>((size_t)SB * SN), *pcnt = dev_alloc<int>((size_t)SB * pr3);
    int *pocc = dev_alloc<int>((size_t)SB * pr3), *plist = dev_alloc<int>((size_t)SB * SN), *pn = dev_alloc<int>(SB);
    unsigned char *prow = dev_alloc<unsigned char>((size_t)SB * PR * PR);
    void *pws = dev_alloc<unsigned char>(bdm_voxelize_workspace_bytes(SB, SN, PR));
    ABI_OK(bdm_voxel_coords(SB, SN, PR, 0.f, sc, pnorm, pvox, nullptr));
    HIP_OK(hipDeviceSynchronize());
    cases.push_back({"lib sampler: voxelize_plan_full (16 x 4096 points, 32^3)", [=] {
      ABI_OK(bdm_voxelize_plan_full(SB, SN, PR, SN, pvox, pind, pcnt, pws, pocc, plist, pn, prow, nullptr)); }, pocc, (size_t)SB * pr3 * 4});
  }
  // ---- trivial kernels of this file
  const size_t n4 = (size_t)B * M * N / 4;
  cases.push_back({"trivial copy (float4 per thread)", [&] { hipLaunchKernelGGL(copy_kernel, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float4 *)y1, (float4 *)y3, n4); }, y3, n4 * 16});
  cases.push_back({"trivial swish (v_exp_f32 / v_rcp_f32)", [&] { hipLaunchKernelGGL(swish_kernel, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float4 *)y1, (float4 *)y3, n4); }, y3, n4 * 16});
  cases.push_back({"trivial copy through LDS", [&] { hipLaunchKernelGGL(lds_copy_kernel, dim3((n4 + 255) / 256), dim3(256), 0, 0, (const float4 *)y1, (float4 *)y3, n4); }, y3, n4 * 16});
  cases.push_back({"trivial grid-stride copy (256 long-running workgroups)", [&] { hipLaunchKernelGGL(long_copy_kernel, dim3(256), dim3(256), 0, 0, (const float4 *)y1, (float4 *)y3, n4); }, y3, n4 * 16});

  int total_bad = 0;
  if (!only || strstr("probe", only)) {  // hardware-resource probes (absolute: no reference repetition needed)
    int *bad = dev_alloc<int>(4);
    float *px = dev_random(1 << 16);
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(probe_barrier_kernel, dim3(1024), dim3(256), 0, 0, bad, 8);
      hipLaunchKernelGGL(probe_lds_kernel, dim3(1024), dim3(256), 0, 0, bad, 4);
      hipLaunchKernelGGL(probe_regs_kernel, dim3(1024), dim3(256), 0, 0, bad, 4, px);
      HIP_OK(hipDeviceSynchronize());
    }
    {
      int *bad2 = dev_alloc<int>(8);
      unsigned *log = dev_alloc<unsigned>(128);
      HIP_OK(hipFuncSetAttribute((const void *)probe_brim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      const int sizes_kb[] = {16, 20, 32, 40, 53, 64, 80, 160};
      for (int kb : sizes_kb) {
        HIP_OK(hipMemset(bad2, 0, 8 * sizeof(int)));
        const int bytes = kb * 1024, per_cu = (160 * 1024) / bytes;
        for (int r = 0; r < (reps < 40 ? reps : 40); ++r) {
          hipLaunchKernelGGL(probe_brim_kernel, dim3(256 * per_cu * 2), dim3(256), bytes, 0, bytes / 4, 4, bad2, log);
          HIP_OK(hipDeviceSynchronize());
        }
        int hb[8]; unsigned hl[128];
        HIP_OK(hipMemcpy(hb, bad2, sizeof(hb), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hl, log, sizeof(hl), hipMemcpyDeviceToHost));
        printf("probe brim: %3d KB per workgroup (%d fit a CU): %d workgroup-iterations with changed LDS words, %d words", kb, per_cu, hb[3], hb[4]);
        for (int q = 0; q < hb[3] && q < 4; ++q) printf("  [LDS_ALLOC 0x%08x first word %u count %u saw 0x%08x]", hl[q * 4], hl[q * 4 + 1], hl[q * 4 + 2], hl[q * 4 + 3]);
        printf("\n");
        total_bad += hb[3];
      }
    }
    int h[4];
    HIP_OK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
    printf("probe: barrier released early %d times; LDS words changed under their owner %d times; live registers changed %d times (%d launches each)\n",
           h[0], h[1], h[2], reps);
    total_bad += h[0] + h[1] + h[2];
  }
  for (Case &c : cases) {
    if (only && !strstr(c.name, only)) continue;
    std::vector<uint32_t> ref(c.bytes / 4), cur(c.bytes / 4);
    int bad = 0;
    bool shown = false;
    for (int r = 0; r <= reps; ++r) {
      HIP_OK(hipMemset(const_cast<void *>(c.out), 0xFF, c.bytes));  // a launch that writes nothing is a difference, not a repeat
      c.launch();
      HIP_OK(hipDeviceSynchronize());
      HIP_OK(hipMemcpy(r == 0 ? ref.data() : cur.data(), c.out, c.bytes, hipMemcpyDeviceToHost));
      if (r == 0 || memcmp(ref.data(), cur.data(), c.bytes) == 0) continue;
      ++bad;
      if (!shown) {
        shown = true;
        size_t first = 0, count = 0, last = 0;
        for (size_t i = 0; i < ref.size(); ++i)
          if (ref[i] != cur[i]) { if (!count) first = i; last = i; ++count; }
        float a, b;
        memcpy(&a, &ref[first], 4); memcpy(&b, &cur[first], 4);
        printf("    first differing repetition %d: %zu words differ, word range [%zu, %zu], byte offset %% 128 of the first = %zu; first: %g vs %g\n",
               r, count, first, last, first * 4 % 128, a, b);
      }
    }
    printf("%-92s %4d / %d repetitions differ from the first\n", c.name, bad, reps);
    fflush(stdout);
    total_bad += bad;
  }
  {
    std::vector<uint32_t> g1(grid0.size()), c1(coords0.size());
    HIP_OK(hipMemcpy(g1.data(), grid, g1.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(c1.data(), coords, c1.size() * 4, hipMemcpyDeviceToHost));
    size_t gd = 0, cd = 0;
    for (size_t i = 0; i < g1.size(); ++i) gd += g1[i] != grid0[i];
    for (size_t i = 0; i < c1.size(); ++i) cd += c1[i] != coords0[i];
    printf("static inputs of the devoxelisation / gather8 victims at the end: %zu grid words and %zu coordinate words changed\n", gd, cd);
  }
  if (bg) { g_stop = true; bg->join(); }
  return total_bad ? 1 : 0;
}
