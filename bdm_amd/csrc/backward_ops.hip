// backward_ops.hip -- the five gradient operators of the reference plugin `_pvcnn_backend` (training half:
// grouping.cu:58-77, neighbor_interpolate.cu:145-170, trilinear_devox.cu:119-162, sampling.cu:52-66, vox.cu:86-110) and
// the training-mode devoxelisation forward that saves their operands (trilinear_devox.cu:21-105 with is_training = true).
// They serve the Merging fusion-decoder training job (main_merging.py:242-366), which is outside the sampling hot path;
// they exist so that the plugin surface is complete.
//
// The reference scatters with float atomicAdd from one block per shape.  Here a launch covers (shape, channel block)
// pairs on the whole machine; the scatter-adds stay float atomics as in the reference (the order of the additions into one
// cell is timing dependent there too), except avg_voxelize_backward, which is a pure gather.
#include "../../include/bdm_hip.h"
#include "common.h"

#pragma clang fp contract(off)

using namespace bdm;

namespace {

__global__ void gather_grad_kernel(int c, int n, int m, const float *__restrict__ grad_y, const int *__restrict__ idx,
                                   float *__restrict__ grad_x) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const float *gy = grad_y + ((size_t)bi * c + ci) * m;
  float *gx = grad_x + ((size_t)bi * c + ci) * n;
  const int *ix = idx + (size_t)bi * m;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) atomicAdd(gx + ix[j], gy[j]);
}

__global__ void grouping_grad_kernel(int c, int n, int mu, const float *__restrict__ grad_y, const int *__restrict__ idx,
                                     float *__restrict__ grad_x) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const float *gy = grad_y + ((size_t)bi * c + ci) * mu;
  float *gx = grad_x + ((size_t)bi * c + ci) * n;
  const int *ix = idx + (size_t)bi * mu;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < mu; e += gridDim.x * blockDim.x) atomicAdd(gx + ix[e], gy[e]);
}

__global__ void three_nn_grad_kernel(int c, int n, int m, const float *__restrict__ grad_y, const int *__restrict__ idx,
                                     const float *__restrict__ w, float *__restrict__ grad_x) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const float *gy = grad_y + ((size_t)bi * c + ci) * n;
  float *gx = grad_x + ((size_t)bi * c + ci) * m;
  const int *ix = idx + (size_t)bi * 3 * n;
  const float *wb = w + (size_t)bi * 3 * n;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const float g = gy[j];
    atomicAdd(gx + ix[j], g * wb[j]);
    atomicAdd(gx + ix[j + n], g * wb[j + n]);
    atomicAdd(gx + ix[j + 2 * n], g * wb[j + 2 * n]);
  }
}

__global__ void devox_grad_kernel(int c, int n, int r3, const int *__restrict__ inds, const float *__restrict__ wgts,
                                  const float *__restrict__ grad_y, float *__restrict__ grad_x) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const float *gy = grad_y + ((size_t)bi * c + ci) * n;
  float *gx = grad_x + ((size_t)bi * c + ci) * r3;
  const int *ix = inds + (size_t)bi * 8 * n;
  const float *wb = wgts + (size_t)bi * 8 * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float g = gy[i];
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(gx + ix[i + k * n], wb[i + k * n] * g);
  }
}

// grad_x[c][i] = grad_y[c][ind[i]] * (1 / cnt[ind[i]])  -- each point is written once: a gather, no atomics needed
__global__ void vox_grad_kernel(int c, int n, int r3, const int *__restrict__ ind, const int *__restrict__ cnt,
                                const float *__restrict__ grad_y, float *__restrict__ grad_x) {
  const int bi = blockIdx.z;
  const int *id = ind + (size_t)bi * n, *cn = cnt + (size_t)bi * r3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int pos = id[i], cur = cn[pos];
    const float inv = cur > 0 ? (float)(1.0 / (double)(float)cur) : 0.f;  // vox.cu:102: 1.0 / float(cnt), narrowed
    for (int ci = blockIdx.y; ci < c; ci += gridDim.y)
      grad_x[((size_t)bi * c + ci) * n + i] = cur > 0 ? grad_y[((size_t)bi * c + ci) * r3 + pos] * inv : 0.f;
  }
}

// corner indices and weights of the trilinear gather, in the order 000, 001, 010, 011, 100, 101, 110, 111 (z fastest);
// the +1 neighbour on an axis is addressed only when the fractional part is > 0 (trilinear_devox.cu:64-75)
__global__ void devox_train_kernel(int c, int n, int r, const float *__restrict__ coords, const float *__restrict__ grid,
                                   float *__restrict__ out, int *__restrict__ inds, float *__restrict__ wgts) {
  const int bi = blockIdx.y, r2 = r * r, r3 = r2 * r;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *cb = coords + (size_t)bi * 3 * n;
  const float x = cb[i], y = cb[i + n], z = cb[i + 2 * n];
  const float xl = floorf(x), yl = floorf(y), zl = floorf(z);
  const float x1 = x - xl, y1 = y - yl, z1 = z - zl, x0 = 1.0f - x1, y0 = 1.0f - y1, z0 = 1.0f - z1;
  const float w[8] = {x0 * y0 * z0, x0 * y0 * z1, x0 * y1 * z0, x0 * y1 * z1, x1 * y0 * z0, x1 * y0 * z1, x1 * y1 * z0, x1 * y1 * z1};
  const int sx = x1 > 0 ? r2 : 0, sy = y1 > 0 ? r : 0, sz = z1 > 0 ? 1 : 0;
  const int i000 = (int)xl * r2 + (int)yl * r + (int)zl;
  const int id[8] = {i000, i000 + sz, i000 + sy, i000 + sy + sz, i000 + sx, i000 + sx + sz, i000 + sx + sy, i000 + sx + sy + sz};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    inds[((size_t)bi * 8 + k) * n + i] = id[k];
    wgts[((size_t)bi * 8 + k) * n + i] = w[k];
  }
  for (int ci = 0; ci < c; ++ci) {
    const float *g = grid + ((size_t)bi * c + ci) * r3;
    float acc = w[0] * g[id[0]];
#pragma unroll
    for (int k = 1; k < 8; ++k) acc = acc + w[k] * g[id[k]];
    out[((size_t)bi * c + ci) * n + i] = acc;
  }
}

inline int zero(void *p, size_t bytes, hipStream_t s, const char *what) {
  if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess) {
    set_error("%s: hipMemsetAsync failed", what);
    return BDM_ERR_LAUNCH;
  }
  return BDM_OK;
}

}  // namespace

extern "C" int bdm_gather_features_backward(int b, int c, int n, int m, const float *grad_y, const int *indices,
                                            float *grad_x, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0, "gather_features_backward: bad sizes");
  if (b == 0 || c == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = zero(grad_x, sizeof(float) * (size_t)b * c * n, s, "gather_features_backward")) return rc;
  if (m == 0) return BDM_OK;
  hipLaunchKernelGGL(gather_grad_kernel, dim3(cdiv(m, 256), c, b), dim3(256), 0, s, c, n, m, grad_y, indices, grad_x);
  return launch_status("gather_features_backward");
}

extern "C" int bdm_grouping_backward(int b, int c, int n, int m, int u, const float *grad_y, const int *indices,
                                     float *grad_x, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 1 && m >= 0 && u >= 0, "grouping_backward: bad sizes");
  if (b == 0 || c == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = zero(grad_x, sizeof(float) * (size_t)b * c * n, s, "grouping_backward")) return rc;
  if (m * u == 0) return BDM_OK;
  hipLaunchKernelGGL(grouping_grad_kernel, dim3(cdiv(m * u, 256), c, b), dim3(256), 0, s, c, n, m * u, grad_y, indices, grad_x);
  return launch_status("grouping_backward");
}

extern "C" int bdm_three_nn_interpolate_backward(int b, int c, int n, int m, const float *grad_y, const int *indices,
                                                 const float *weights, float *grad_x, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && m >= 1, "three_nn_interpolate_backward: bad sizes");
  if (b == 0 || c == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = zero(grad_x, sizeof(float) * (size_t)b * c * m, s, "three_nn_interpolate_backward")) return rc;
  if (n == 0) return BDM_OK;
  hipLaunchKernelGGL(three_nn_grad_kernel, dim3(cdiv(n, 256), c, b), dim3(256), 0, s, c, n, m, grad_y, indices, weights, grad_x);
  return launch_status("three_nn_interpolate_backward");
}

extern "C" int bdm_trilinear_devoxelize_backward(int b, int c, int n, int r, const int *inds, const float *wgts,
                                                 const float *grad_y, float *grad_x, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && r >= 1, "trilinear_devoxelize_backward: bad sizes");
  if (b == 0 || c == 0) return BDM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int r3 = r * r * r;
  if (int rc = zero(grad_x, sizeof(float) * (size_t)b * c * r3, s, "trilinear_devoxelize_backward")) return rc;
  if (n == 0) return BDM_OK;
  hipLaunchKernelGGL(devox_grad_kernel, dim3(cdiv(n, 256), c, b), dim3(256), 0, s, c, n, r3, inds, wgts, grad_y, grad_x);
  return launch_status("trilinear_devoxelize_backward");
}

extern "C" int bdm_avg_voxelize_backward(int b, int c, int n, int r, const int *ind, const int *cnt, const float *grad_y,
                                         float *grad_x, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && r >= 1, "avg_voxelize_backward: bad sizes");
  if (b == 0 || c == 0 || n == 0) return BDM_OK;
  hipLaunchKernelGGL(vox_grad_kernel, dim3(cdiv(n, 256), c < 32 ? c : 32, b), dim3(256), 0, (hipStream_t)stream, c, n, r * r * r,
                     ind, cnt, grad_y, grad_x);
  return launch_status("avg_voxelize_backward");
}

extern "C" int bdm_trilinear_devoxelize_forward_training(int b, int c, int n, int r, const float *coords, const float *grid,
                                                         float *out, int *inds, float *wgts, void *stream) {
  BDM_REQUIRE(b >= 0 && c >= 0 && n >= 0 && r >= 1 && inds != nullptr && wgts != nullptr, "trilinear_devoxelize_forward_training: bad arguments");
  if (b == 0 || n == 0) return BDM_OK;
  hipLaunchKernelGGL(devox_train_kernel, dim3(cdiv(n, 256), b), dim3(256), 0, (hipStream_t)stream, c, n, r, coords, grid, out,
                     inds, wgts);
  return launch_status("trilinear_devoxelize_forward_training");
}
