/*
 * bdm_hip.h -- C ABI of libbdm_hip.so, the MI355X (gfx950) implementation of BDM's
 * coupled-diffusion sampling hot path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*; tensors are contiguous,
 *     channel-first (B, C, N), fp32 / int32 -- the layout of the reference plugin;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on that stream (no allocation, no synchronisation: safe to capture
 *     into a hipGraph);
 *   - outputs and workspaces are caller-allocated; the callee never allocates;
 *   - the return value is 0 on success, non-zero otherwise (1 bad argument, 2 launch
 *     failure, 3 unsupported configuration); bdm_last_error() returns a thread-local
 *     message.  Nothing in the library calls exit() (the reference's CUDA_CHECK_ERRORS
 *     does: experiments/model/pvcnn/modules/functional/src/cuda_utils.cuh:28-37).
 *
 * Section 1 replaces, one entry point per function, the forward half of the reference's
 * pybind11 plugin `_pvcnn_backend`
 *   (experiments/model/pvcnn/modules/functional/src/bindings.cpp:10-37; identical copy
 *    in experiments/pvd/modules/functional/src/).
 * Sections 2-4 replace the stock-PyTorch operators and the Python loops the reference
 * runs around that plugin inside the per-step denoiser forward and the DDPM loop
 * (reference file:line given per function).
 */
#ifndef BDM_HIP_H
#define BDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *bdm_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int bdm_abi_version(void);

/* ------------------------------------------------------------------------------------
 * 1. `_pvcnn_backend` forward operators
 * ---------------------------------------------------------------------------------- */

/* furthest_point_sampling (sampling.cpp:43-58, sampling.cu:86-167).
 * coords (b,3,n) -> indices (b,m) int32.  indices[.,0] = 0; ties between equal maxima
 * resolve to the smallest (k mod 512), then the smallest k, as the reference's 512-thread
 * reduction does.  centers_out (b,3,m) may be NULL; when given it receives
 * gather(coords, indices), i.e. functional/sampling.py:37-48 in one launch. */
int bdm_furthest_point_sampling(int b, int n, int m, const float *coords, int *indices,
                                float *centers_out, void *stream);

/* gather_features_forward (sampling.cpp:6-23, sampling.cu:17-31).
 * features (b,c,n), indices (b,m) -> out (b,c,m). */
int bdm_gather_features_forward(int b, int c, int n, int m, const float *features,
                                const int *indices, float *out, void *stream);

/* ball_query (ball_query.cpp:6-30, ball_query.cu:19-50).
 * centers (b,3,m), points (b,3,n) -> neighbors (b,m,u) int32: first u points (ascending
 * index) with d2 < radius*radius (strict), padded with the first hit, all zero when no hit. */
int bdm_ball_query(int b, int n, int m, float radius, int u, const float *centers,
                   const float *points, int *neighbors, void *stream);

/* grouping_forward (grouping.cpp:6-24, grouping.cu:18-36).
 * features (b,c,n), indices (b,m,u) -> out (b,c,m,u). */
int bdm_grouping_forward(int b, int c, int n, int m, int u, const float *features,
                         const int *indices, float *out, void *stream);

/* three_nearest_neighbors_interpolate_forward (neighbor_interpolate.cpp:6-40,
 * neighbor_interpolate.cu:20-129).  points (b,3,n), centers (b,3,m), features (b,c,m)
 * -> out (b,c,n), indices (b,3,n) int32, weights (b,3,n). */
int bdm_three_nn_interpolate_forward(int b, int c, int m, int n, const float *points,
                                     const float *centers, const float *features, float *out,
                                     int *indices, float *weights, void *stream);
/* The two halves of the above, so that one search can serve several feature tensors
 * (the reference repeats the search for t_emb: modules/pointnet.py:107-108). */
int bdm_three_nn_search(int b, int m, int n, const float *points, const float *centers,
                        int *indices, float *weights, void *stream);
/* features (b,c,m) with row stride ld_f and batch stride bs_f (elements);
 * out likewise (ld_o, bs_o): lets the caller write straight into a concat buffer. */
int bdm_three_nn_apply(int b, int c, int m, int n, const float *features, long long bs_f, int ld_f,
                       const int *indices, const float *weights, float *out, long long bs_o, int ld_o,
                       void *stream);

/* avg_voxelize_forward (vox.cpp:17-43, vox.cu:18-72).
 * features (b,c,n), coords (b,3,n) int32 in [0,r) -> out (b,c,r^3), ind (b,n), cnt (b,r^3).
 * Deterministic: every voxel sums its points in ascending point index (the reference's
 * float atomicAdd leaves the order to thread timing).  workspace: bdm_voxelize_workspace_bytes. */
size_t bdm_voxelize_workspace_bytes(int b, int n, int r);
int bdm_avg_voxelize_forward(int b, int c, int n, int r, const float *features, const int *coords,
                             float *out, int *ind, int *cnt, void *workspace, void *stream);

/* trilinear_devoxelize_forward, inference form (trilinear_devox.cpp:18-55,
 * trilinear_devox.cu:21-105).  coords (b,3,n) float in [0,r-1], grid (b,c,r^3) -> out (b,c,n). */
int bdm_trilinear_devoxelize_forward(int b, int c, int n, int r, const float *coords,
                                     const float *grid, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BDM_HIP_H */
