// sampler_ops.hip -- the per-step glue of the coupled DDPM loop as gfx950 kernels:
//   PC^2 scheduler step   (diffusers 0.21.0 DDPMScheduler.step; call sites model/model.py:193,286,563)
//   PVD scheduler step    (pvd/__init__.py:136-224)
//   mean-centring         (main_blending.py:229, model/model.py:530-531)
//   BDM-Blending select   (main_blending.py:326-344)
//   projection conditioning: point rasterisation + feature scatter
//                         (model/projection_model.py:127-157,179-231; pytorch3d PointsRasterizer)
// Elementwise arithmetic mirrors torch's op-by-op rounding (no FMA contraction in this file).
#include "../../include/bdm_hip.h"
#include "common.h"

#pragma clang fp contract(off)

using namespace bdm;

// x_{t-1} = c_x0 * ((x - sqrt(1-abar_t) * eps) / sqrt(abar_t)) + c_x * x  [+ sigma * z]
// (x and out may be the SAME buffer -- the reverse loops step in place -- hence no __restrict__ on them)
__global__ void ddpm_step_kernel(long long n, const float *x, const float *__restrict__ eps,
                                 const float *__restrict__ z, float sqrt_beta_prod, float sqrt_alpha_prod,
                                 float c_x0, float c_x, float sigma, float *out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float xi = x[i];
    const float x0 = (xi - sqrt_beta_prod * eps[i]) / sqrt_alpha_prod;
    float v = c_x0 * x0 + c_x * xi;
    if (z) v = v + sigma * z[i];
    out[i] = v;
  }
}
extern "C" int bdm_ddpm_step(long long n, const float *x, const float *eps, const float *noise,
                             float sqrt_beta_prod, float sqrt_alpha_prod, float coef_x0, float coef_x,
                             float sigma, float *out, void *stream) {
  BDM_REQUIRE(n >= 0, "ddpm_step: bad size");
  if (n == 0) return BDM_OK;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, eps, noise,
                     sqrt_beta_prod, sqrt_alpha_prod, coef_x0, coef_x, sigma, out);
  return launch_status("ddpm_step");
}

// Same arithmetic with the five per-timestep scalars read from DEVICE memory: the form a captured hipGraph replays for
// every timestep (the host refreshes coef[0..4] between replays; sigma == 0 marks t == 0: no noise is added).
__global__ void ddpm_step_dev_kernel(long long n, const float *__restrict__ x, const float *__restrict__ eps,
                                     const float *__restrict__ z, const float *__restrict__ coef, float *__restrict__ out) {
  const float sqrt_beta_prod = coef[0], sqrt_alpha_prod = coef[1], c_x0 = coef[2], c_x = coef[3], sigma = coef[4];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float xi = x[i];
    const float x0 = (xi - sqrt_beta_prod * eps[i]) / sqrt_alpha_prod;
    float v = c_x0 * x0 + c_x * xi;
    if (sigma != 0.f) v = v + sigma * z[i];
    out[i] = v;
  }
}
extern "C" int bdm_ddpm_step_dev(long long n, const float *x, const float *eps, const float *noise, const float *coef,
                                 float *out, void *stream) {
  BDM_REQUIRE(n >= 0 && noise != nullptr && coef != nullptr, "ddpm_step_dev: bad arguments");
  if (n == 0) return BDM_OK;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ddpm_step_dev_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, eps, noise, coef, out);
  return launch_status("ddpm_step_dev");
}

// out = ((c0*x0 + c1*x1) + c2*x2 + c3*x3) / div, left to right, k <= 4 terms: the linear multistep combinations and the
// transfer step of the PNDM scheduler (diffusers 0.21.0 PNDMScheduler.step_prk / step_plms / _get_prev_sample)
__global__ void lincomb_kernel(long long n, int k, float c0, float c1, float c2, float c3, const float *__restrict__ x0,
                               const float *__restrict__ x1, const float *__restrict__ x2, const float *__restrict__ x3,
                               float div, float *__restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float acc = c0 * x0[i];
    if (k > 1) acc = acc + c1 * x1[i];
    if (k > 2) acc = acc + c2 * x2[i];
    if (k > 3) acc = acc + c3 * x3[i];
    out[i] = div == 1.0f ? acc : acc / div;
  }
}
extern "C" int bdm_lincomb(long long n, int k, float c0, const float *x0, float c1, const float *x1, float c2,
                           const float *x2, float c3, const float *x3, float div, float *out, void *stream) {
  BDM_REQUIRE(n >= 0 && k >= 1 && k <= 4 && x0 != nullptr && (k < 2 || x1) && (k < 3 || x2) && (k < 4 || x3) && div != 0.f,
              "lincomb: bad arguments");
  if (n == 0) return BDM_OK;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(lincomb_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, k, c0, c1, c2, c3, x0, x1, x2, x3, div,
                     out);
  return launch_status("lincomb");
}

// x0 = a*x - b*eps ; mean = c1*x0 + c2*x ; out = mean + sigma*z   (sigma = 0 at t == 0)
// (x and out are NOT __restrict__: the replayed prior loop steps in place, out == x -- elementwise, each element read and written by one thread)
__global__ void pvd_step_kernel(long long n, const float *x, const float *__restrict__ eps,
                                const float *__restrict__ z, float a, float b, float c1, float c2, float sigma,
                                float *out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float xi = x[i];
    const float x0 = a * xi - b * eps[i];
    const float mean = c1 * x0 + c2 * xi;
    out[i] = mean + sigma * z[i];
  }
}
extern "C" int bdm_pvd_step(long long n, const float *x, const float *eps, const float *noise,
                            float sqrt_recip_abar, float sqrt_recipm1_abar, float coef1, float coef2,
                            float sigma, float *out, void *stream) {
  BDM_REQUIRE(n >= 0 && noise != nullptr, "pvd_step: bad arguments");
  if (n == 0) return BDM_OK;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pvd_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, eps, noise,
                     sqrt_recip_abar, sqrt_recipm1_abar, coef1, coef2, sigma, out);
  return launch_status("pvd_step");
}

// x (B, N, 3) point-major: x -= mean over points (per shape, per axis), in place
__global__ void center_points_kernel(int n, float *__restrict__ x) {
  __shared__ double sh[3][16];
  __shared__ float mean[3];
  float *xb = x + (size_t)blockIdx.x * n * 3;
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n; i += blockDim.x) { s[0] += xb[3 * i]; s[1] += xb[3 * i + 1]; s[2] += xb[3 * i + 2]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    s[d] = wave_sum(s[d]);
    if (lane == 0) sh[d][wave] = s[d];
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double a = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) a += sh[threadIdx.x][w];
    mean[threadIdx.x] = (float)(a / (double)n);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) xb[i] = xb[i] - mean[i % 3];
}
extern "C" int bdm_center_points(int b, int n, float *x, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1, "center_points: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(center_points_kernel, dim3(b), dim3(1024), 0, (hipStream_t)stream, n, x);
  return launch_status("center_points");
}

// out[p] = mask[p] ? prior[p] : recon[p]   for every point p (3 floats each)
__global__ void blend_kernel(long long npts, const float *__restrict__ recon, const float *__restrict__ prior,
                             const long long *__restrict__ mask, float *__restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < 3 * npts;
       i += (long long)gridDim.x * blockDim.x)
    out[i] = mask[i / 3] ? prior[i] : recon[i];
}
extern "C" int bdm_blend_select(long long num_points, const float *recon, const float *prior,
                                const long long *mask, float *out, void *stream) {
  BDM_REQUIRE(num_points >= 0, "blend_select: bad size");
  if (num_points == 0) return BDM_OK;
  int grid = (int)((3 * num_points + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(blend_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, num_points, recon, prior, mask, out);
  return launch_status("blend_select");
}

// -------------------------------------------------------------------------------------
// Projection conditioning
// -------------------------------------------------------------------------------------
// Points are projected with a PerspectiveCameras model (row-vector convention:
// X_view = X_world R + T; ndc = focal * X_view.xy / X_view.z + principal_point) and rasterised
// with the naive PointsRasterizer rule: pixel (yi, xi) has its centre at
// ndc (x, y) = (1 - (2 xi + 1)/W, 1 - (2 yi + 1)/H)  (+X left, +Y up); a point covers the pixel
// when dx^2 + dy^2 < radius^2 and z >= 0; the pixel keeps the point with the smallest z
// (earliest index on ties).  A point then takes the feature vector of the LAST pixel (row-major)
// it owns -- the sequential semantics of the reference's duplicate-index assignment
// (projection_model.py:152-153); points that own no pixel get zeros.
struct Cam { float r[9]; float t[3]; float f[2]; float p[2]; };

__device__ __forceinline__ void project(const Cam &c, float x, float y, float z, float &u, float &v, float &d) {
  const float xv = x * c.r[0] + y * c.r[3] + z * c.r[6] + c.t[0];
  const float yv = x * c.r[1] + y * c.r[4] + z * c.r[7] + c.t[1];
  const float zv = x * c.r[2] + y * c.r[5] + z * c.r[8] + c.t[2];
  u = c.f[0] * xv / zv + c.p[0];
  v = c.f[1] * yv / zv + c.p[1];
  d = zv;
}
__device__ __forceinline__ Cam load_cam(const float *cams, int bi) {
  Cam c;
  const float *p = cams + (size_t)bi * 16;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.r[i] = p[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) c.t[i] = p[9 + i];
  c.f[0] = p[12]; c.f[1] = p[13]; c.p[0] = p[14]; c.p[1] = p[15];
  return c;
}
// nearest pixel index of an NDC coordinate, clamped in FLOAT to [-4, size+4] before the conversion: points that project
// far outside the image (or at infinity, z -> 0+) would otherwise overflow the int conversion and the window loops
__device__ __forceinline__ int nearest_pixel(float ndc, int size) {
  const float p = rintf(((1.f - ndc) * size - 1.f) * 0.5f);
  return (int)fminf(fmaxf(p, -4.f), (float)size + 4.f);  // NaN -> -4 (fmaxf returns the non-NaN operand)
}
#define RAST_WIN 2  // candidate window: pixels within +-2 of the nearest one (radius < 2 pixel pitches)

__global__ void raster_clear_kernel(long long n, unsigned long long *zbuf) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    zbuf[i] = ~0ull;
}

__global__ void raster_splat_kernel(int n, int H, int W, float radius2, const float *__restrict__ pts,
                                    const float *__restrict__ cams, unsigned long long *__restrict__ zbuf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
  if (i >= n) return;
  const Cam c = load_cam(cams, bi);
  const float *p = pts + ((size_t)bi * n + i) * 3;
  float u, v, d;
  project(c, p[0], p[1], p[2], u, v, d);
  if (!(d >= 0.f)) return;  // behind the camera (or NaN)
  // nearest pixel column/row:  u = 1 - (2 xi + 1)/W  ->  xi = ((1 - u) W - 1) / 2
  const int xc = nearest_pixel(u, W), yc = nearest_pixel(v, H);
  unsigned long long *zb = zbuf + (size_t)bi * H * W;
  const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
  for (int yi = yc - RAST_WIN; yi <= yc + RAST_WIN; ++yi) {
    if (yi < 0 || yi >= H) continue;
    const float yf = 1.f - (2.f * yi + 1.f) / H;
    const float dy = yf - v;
    for (int xi = xc - RAST_WIN; xi <= xc + RAST_WIN; ++xi) {
      if (xi < 0 || xi >= W) continue;
      const float xf = 1.f - (2.f * xi + 1.f) / W;
      const float dx = xf - u;
      if (dx * dx + dy * dy < radius2) atomicMin(&zb[(size_t)yi * W + xi], key);
    }
  }
}

__global__ void raster_owner_kernel(int n, int H, int W, float radius2, const float *__restrict__ pts,
                                    const float *__restrict__ cams, const unsigned long long *__restrict__ zbuf,
                                    int *__restrict__ pix_of_point) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
  if (i >= n) return;
  const Cam c = load_cam(cams, bi);
  const float *p = pts + ((size_t)bi * n + i) * 3;
  float u, v, d;
  project(c, p[0], p[1], p[2], u, v, d);
  int owner = -1;
  if (d >= 0.f) {
    const int xc = nearest_pixel(u, W), yc = nearest_pixel(v, H);
    const unsigned long long *zb = zbuf + (size_t)bi * H * W;
    for (int yi = yc - RAST_WIN; yi <= yc + RAST_WIN; ++yi) {
      if (yi < 0 || yi >= H) continue;
      for (int xi = xc - RAST_WIN; xi <= xc + RAST_WIN; ++xi) {
        if (xi < 0 || xi >= W) continue;
        if ((unsigned)(zb[(size_t)yi * W + xi] & 0xFFFFFFFFull) == (unsigned)i &&
            zb[(size_t)yi * W + xi] != ~0ull)
          owner = yi * W + xi;  // row-major scan: the last owned pixel wins
      }
    }
  }
  pix_of_point[(size_t)bi * n + i] = owner;
}

extern "C" size_t bdm_rasterize_workspace_bytes(int b, int h, int w) { return sizeof(unsigned long long) * (size_t)b * h * w; }

extern "C" int bdm_rasterize_points(int b, int n, int h, int w, float radius, const float *points,
                                    const float *cameras, int *pix_of_point, void *workspace, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && h >= 1 && w >= 1, "rasterize_points: bad sizes");
  BDM_REQUIRE(workspace != nullptr, "rasterize_points: workspace is NULL");
  BDM_REQUIRE(radius * (h > w ? h : w) * 0.5f < (float)RAST_WIN, "rasterize_points: radius %g spans more than %d pixels", radius, RAST_WIN);
  if (b == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  unsigned long long *zbuf = (unsigned long long *)workspace;
  hipLaunchKernelGGL(raster_clear_kernel, dim3(1024), dim3(256), 0, s, (long long)b * h * w, zbuf);
  hipLaunchKernelGGL(raster_splat_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, h, w, radius * radius, points,
                     cameras, zbuf);
  hipLaunchKernelGGL(raster_owner_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, s, n, h, w, radius * radius, points,
                     cameras, zbuf, pix_of_point);
  return launch_status("rasterize_points");
}

// out (B, N, 3 + C) point-major = cat[x_t, feature_image[pix_of_point]]  (zeros when pix < 0);
// feature image stored pixel-major (B, H*W, C): one contiguous C-vector per pixel.
__global__ void condition_gather_kernel(int n, int C, int HW, const float *__restrict__ x_t,
                                        const float *__restrict__ feat, const int *__restrict__ pix,
                                        float *__restrict__ out) {
  const int pt = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63, bi = blockIdx.y;
  if (pt >= n) return;
  const int px = pix[(size_t)bi * n + pt];
  float *o = out + ((size_t)bi * n + pt) * (3 + C);
  if (lane < 3) o[lane] = x_t[((size_t)bi * n + pt) * 3 + lane];
  if (px >= 0) {
    const float *f = feat + ((size_t)bi * HW + px) * C;
    for (int c = lane; c < C; c += 64) o[3 + c] = f[c];
  } else {
    for (int c = lane; c < C; c += 64) o[3 + c] = 0.f;
  }
}
extern "C" int bdm_condition_gather(int b, int n, int c, int hw, const float *x_t, const float *feature_image,
                                    const int *pix_of_point, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && c >= 0 && hw >= 1, "condition_gather: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(condition_gather_kernel, dim3(cdiv(n, 4), b), dim3(256), 0, (hipStream_t)stream, n, c, hw, x_t,
                     feature_image, pix_of_point, out);
  return launch_status("condition_gather");
}

// The same gather written CHANNEL-FIRST, out (B, 3 + C, N): what the denoiser consumes (point_cloud_model.py:65 transposes
// the point-major tensor first thing) -- the 100 MB transpose pass per step disappears.  Workgroup = 64 points; the
// pixel-major feature rows are read as contiguous 256-byte pieces (64 channels of one point per wave step), transposed through
// an LDS tile and written as 256-byte runs along the point axis.
// Cw: feature channels actually written (C, or 0: the coordinate rows only -- every consumer of the feature rows takes its share from
// a hoisted per-pixel map instead, bdm_condition_xyz_cf)
__global__ __launch_bounds__(256) void condition_gather_cf_kernel(int n, int C, int Cw, int HW, const float *__restrict__ x_t,
                                                                  const float *__restrict__ feat,
                                                                  const int *__restrict__ pix, float *__restrict__ out) {
  __shared__ float tile[64][65];
  __shared__ int s_px[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, bi = blockIdx.y, n0 = blockIdx.x * 64;
  float *ob = out + (size_t)bi * (3 + C) * n;
  if (tid < 64) s_px[tid] = (Cw > 0 && n0 + tid < n) ? pix[(size_t)bi * n + n0 + tid] : -1;
  if (tid < 192) {  // x_t (B, N, 3) -> channels 0..2
    const int d = tid >> 6, pt = n0 + lane;
    if (pt < n) ob[(size_t)d * n + pt] = x_t[((size_t)bi * n + pt) * 3 + d];
  }
  __syncthreads();
  for (int c0 = 0; c0 < Cw; c0 += 64) {
    const int c = c0 + lane;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {  // wave w reads the rows of points w*16 .. w*16+15, 64 channels each
      const int pl = wave * 16 + k, px = s_px[pl];
      tile[pl][lane] = (px >= 0 && c < C) ? feat[((size_t)bi * HW + px) * C + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {  // wave w writes channels c0 + w*16 .. +15, 64 points each
      const int cl = wave * 16 + k;
      if (c0 + cl < C && n0 + lane < n) ob[(size_t)(3 + c0 + cl) * n + n0 + lane] = tile[lane][cl];
    }
    __syncthreads();
  }
}
extern "C" int bdm_condition_gather_cf(int b, int n, int c, int hw, const float *x_t, const float *feature_image,
                                       const int *pix_of_point, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && c >= 0 && hw >= 1, "condition_gather_cf: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(condition_gather_cf_kernel, dim3(cdiv(n, 64), b), dim3(256), 0, (hipStream_t)stream, n, c, c, hw, x_t,
                     feature_image, pix_of_point, out);
  return launch_status("condition_gather_cf");
}

// Rows 0..2 (the coordinates) of that tensor only: out (b, 3 + c, n) keeps its layout, rows 3.. are left UNWRITTEN.  For the reverse
// loop when every consumer of the feature rows reads a hoisted per-pixel map instead (ops.Conditioning: 100 MB per step at B = 16 that
// nothing would read); bdm_condition_gather_cf on the same tensor completes it.
extern "C" int bdm_condition_xyz_cf(int b, int n, int c, const float *x_t, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && c >= 0 && x_t && out, "condition_xyz_cf: bad arguments");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(condition_gather_cf_kernel, dim3(cdiv(n, 64), b), dim3(256), 0, (hipStream_t)stream, n, c, 0, 1, x_t,
                     (const float *)nullptr, (const int *)nullptr, out);
  return launch_status("condition_xyz_cf");
}

// -------------------------------------------------------------------------------------
// Quality metrics (evaluation/evaluation_cd.py:111-131, evaluation/evaluation_f1.py:90-110): per-point squared
// distance to the nearest point of the other cloud.  src (b, n, 3), tgt (b, m, 3) point-major -> out (b, n).
// -------------------------------------------------------------------------------------
__global__ void nn_sqdist_kernel(int n, int m, const float *__restrict__ src, const float *__restrict__ tgt,
                                 float *__restrict__ out) {
  __shared__ float st[3 * 1024];
  const int bi = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  const float *s = src + ((size_t)bi * n + (i < n ? i : 0)) * 3;
  const float x = s[0], y = s[1], z = s[2];
  float best = INFINITY;
  for (int base = 0; base < m; base += 1024) {
    const int len = min(1024, m - base);
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * len; t += blockDim.x) st[t] = tgt[((size_t)bi * m + base) * 3 + t];
    __syncthreads();
    for (int k = 0; k < len; ++k) {
      const float dx = x - st[3 * k], dy = y - st[3 * k + 1], dz = z - st[3 * k + 2];
      best = fminf(best, dx * dx + dy * dy + dz * dz);
    }
  }
  if (i < n) out[(size_t)bi * n + i] = best;
}
extern "C" int bdm_nn_sqdist(int b, int n, int m, const float *src, const float *tgt, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && n >= 1 && m >= 1, "nn_sqdist: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(nn_sqdist_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, (hipStream_t)stream, n, m, src, tgt, out);
  return launch_status("nn_sqdist");
}

// DDIM step (diffusers 0.21.0 DDIMScheduler.step, epsilon prediction, clip_sample=False):
//   x0 = (x - sqrt(1-abar_t) eps) / sqrt(abar_t);  out = sqrt(abar_prev) x0 + sqrt(1 - abar_prev - std^2) eps [+ std z]
__global__ void ddim_step_kernel(long long n, const float *__restrict__ x, const float *__restrict__ eps,
                                 const float *__restrict__ z, float sqrt_beta_prod, float sqrt_alpha_prod, float c_x0,
                                 float c_eps, float sigma, float *__restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float e = eps[i];
    const float x0 = (x[i] - sqrt_beta_prod * e) / sqrt_alpha_prod;
    float v = c_x0 * x0 + c_eps * e;
    if (z) v = v + sigma * z[i];
    out[i] = v;
  }
}
extern "C" int bdm_ddim_step(long long n, const float *x, const float *eps, const float *noise, float sqrt_beta_prod,
                             float sqrt_alpha_prod, float coef_x0, float coef_eps, float sigma, float *out, void *stream) {
  BDM_REQUIRE(n >= 0, "ddim_step: bad size");
  if (n == 0) return BDM_OK;
  int grid = (int)((n + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, x, eps, noise, sqrt_beta_prod,
                     sqrt_alpha_prod, coef_x0, coef_eps, sigma, out);
  return launch_status("ddim_step");
}
