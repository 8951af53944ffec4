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
/* PointNetFPModule input assembly in one launch (pointnet.py:104-112): out0 = cat[interpolate(fa), fs] (c_a + c_s rows),
 * out1 = interpolate(ft) (c_t rows), with the (indices, weights) of bdm_three_nn_search.  Same arithmetic as
 * bdm_three_nn_apply (three products, two adds, no contraction). */
int bdm_fp_assemble(int b, int m, int n, const int *indices, const float *weights, int c_a, const float *fa, long long bs_a,
                    int ld_a, int c_s, const float *fs, long long bs_s, int ld_s, int c_t, const float *ft, long long bs_t,
                    int ld_t, float *out0, long long bs_0, int ld_0, float *out1, long long bs_1, int ld_1, void *stream);
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

/* --- the training half of the plugin (bindings.cpp:10-37): gradient operators and the training-mode devoxelisation.
 * Outside the sampling hot path (they serve training_bdm_merging, main_merging.py:242-366); present so that the plugin
 * surface is complete.  grad_x is fully written by the callee (zero-filled, then accumulated).  Scatter-adds are float
 * atomics as in the reference (grouping.cu:58-77, neighbor_interpolate.cu:145-170, trilinear_devox.cu:119-162,
 * sampling.cu:52-66); avg_voxelize_backward (vox.cu:86-110) is a gather. */
int bdm_gather_features_backward(int b, int c, int n, int m, const float *grad_y, const int *indices,
                                 float *grad_x, void *stream);
int bdm_grouping_backward(int b, int c, int n, int m, int u, const float *grad_y, const int *indices,
                          float *grad_x, void *stream);
int bdm_three_nn_interpolate_backward(int b, int c, int n, int m, const float *grad_y, const int *indices,
                                      const float *weights, float *grad_x, void *stream);
int bdm_trilinear_devoxelize_backward(int b, int c, int n, int r, const int *inds, const float *wgts,
                                      const float *grad_y, float *grad_x, void *stream);
int bdm_avg_voxelize_backward(int b, int c, int n, int r, const int *ind, const int *cnt, const float *grad_y,
                              float *grad_x, void *stream);
/* trilinear_devoxelize_forward with is_training = true: also returns inds (b,8,n) int32 and wgts (b,8,n), corner order
 * 000, 001, 010, 011, 100, 101, 110, 111 (trilinear_devox.cu:21-105). */
int bdm_trilinear_devoxelize_forward_training(int b, int c, int n, int r, const float *coords, const float *grid,
                                              float *out, int *inds, float *wgts, void *stream);

/* ------------------------------------------------------------------------------------
 * 2. Dense per-point / per-voxel operators of one denoiser forward
 *    (stock nn.Modules in the reference; hand-written gfx950 kernels here)
 *    Strided operands: a (B, C, L) tensor is addressed as base + b*bs + c*ld + l (elements),
 *    so callers can read from / write into channel slices of concat buffers
 *    (torch.cat in pvcnn.py:104,120 and pointnet.py:110) without copies.
 * ---------------------------------------------------------------------------------- */

/* Conv1d / Conv2d with kernel 1 (+ bias [+ per-shape bias] [+ LeakyReLU]):
 *   y[b] (m x n) = act(w (m x k, row stride ldw) * x[b] (k x n) + bias[m] + batch_bias[b][m]) + residual[b]
 * Replaces nn.Conv1d/nn.Conv2d(k=1) in modules/shared_mlp.py:25-30, the q/k/v/out projections of
 * modules/pvconv.py:21-31, pvcnn_fuse.py:111-123 and the classifier (pvcnn.py:62-69).
 * act: 0 none, 2 LeakyReLU(slope), 3 exact GELU; bias, batch_bias and residual may be NULL (residual: the
 * `proj(x) + skip` additions of pvcnn_fuse.py:203-212).  f32-input MFMA, fp32 accumulate. */
int bdm_pointwise_conv(int b, int m, int k, int n, const float *w, int ldw, const float *x,
                       long long bs_x, int ld_x, const float *bias, const float *batch_bias,
                       int ld_bb, const float *residual, long long bs_r, int ld_r, float *y,
                       long long bs_y, int ld_y, int act, float slope, void *stream);

/* nn.GroupNorm(groups, c) over (b, c, l) with optional residual added first and optional Swish
 * (shared_mlp.py:27-29; pvconv.py:78-86; Attention: norm(h + x) then Swish, pvconv.py:59-61).
 * act: 0 none, 1 Swish.  In-place (y == x) is allowed.  Deterministic two-pass reduction. */
size_t bdm_group_norm_workspace_bytes(int b, int groups);
int bdm_group_norm(int b, int c, int l, int groups, const float *x, long long bs_x, int ld_x,
                   const float *residual, long long bs_r, int ld_r, const float *gamma,
                   const float *beta, float eps, int act, float *y, long long bs_y, int ld_y,
                   void *workspace, void *stream);

/* features.max(dim=-1) over the neighbour axis (pointnet.py:86): x (b,c,m,u) -> y (b,c,m) strided. */
/* SharedMLP = [Conv k=1 -> GroupNorm(8) -> Swish]* (modules/shared_mlp.py:25-30) without a pass per GroupNorm.
 *   bdm_pointwise_conv_gn: y = W x' + bias with
 *     x' = x, or (in_partial != NULL) x' = Swish(GroupNorm(x)) applied while the operand is staged, the statistics of x taken
 *          from the slice partials in_partial (b, in_groups, in_slices, 2 doubles) its producer left (in_groups <= 8, k <= 1024);
 *     amax != NULL: amax[s * ceil(m / amax_rows) + i] (b * ceil(m / amax_rows) slots, ZERO on entry) receives max |y| over rows
 *          [i * amax_rows, (i + 1) * amax_rows) of shape s (amax_rows % 32 == 0): the per-shape scales of the fp16x3
 *          attention, bdm_attention_core_h2;
 *     x2 != NULL: the operand is torch.cat([x (k1 rows), x2 (k - k1 rows)], dim=1) read in place (pointnet.py:108-110, the skip
 *          features of an FP module are never copied next to the interpolated ones);
 *     and (out_partial != NULL) the (sum, sum of squares) of y per (shape, group of m / out_groups channels) written as
 *          bdm_pointwise_conv_gn_slices(b, m, k, n, out_groups) slices per (shape, group) into out_partial
 *          (b, out_groups, slices, 2 doubles); channels per group must be a power of two >= 4.
 *   bdm_max_over_neighbors_gn: max over the neighbour axis of Swish(GroupNorm(x)), statistics from such partials
 *     (pointnet.py:86 after the last MLP layer).
 * Deterministic (fixed summation orders, no float atomics). */
int bdm_pointwise_conv_gn_slices(int b, int m, int k, int n, int groups);
int bdm_pointwise_conv_gn(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x, int ld_x,
                          const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y, long long bs_y,
                          int ld_y, const void *in_partial, int in_slices, int in_groups, const float *in_gamma,
                          const float *in_beta, float in_eps, int out_groups, void *out_partial, float *amax, int amax_rows,
                          void *stream);
/* bdm_pointwise_conv_gn with a per-element addend: y = W x' + bias + add (add (b, m, n) strided as y), statistics and amax taken
 * over y including the addend -- the hoisted share of a layer whose other input columns were applied to the conditioning image once
 * per trajectory (bdm_sparse_conv_rows_from_map, ops.Conditioning).  Not for the skinny shapes. */
int bdm_pointwise_conv_gn_add(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x, int ld_x,
                              const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y, long long bs_y,
                              int ld_y, const void *in_partial, int in_slices, int in_groups, const float *in_gamma,
                              const float *in_beta, float in_eps, int out_groups, void *out_partial, float *amax, int amax_rows,
                              const float *add, long long bs_add, int ld_add, void *stream);
/* The same with a per-SHAPE bias batch_bias (b, ld_bb >= m): y = (W x' + bias) + batch_bias[shape] (+ add; may be NULL), statistics and
 * amax over that y.  The share of a layer whose remaining input columns are constant along a shape's points -- the time embedding
 * concatenated to the features (pvcnn.py:88, pointnet.py:104-112): W . [x ; t 1^T] = W_x . x + (W_t . t) 1^T. */
int bdm_pointwise_conv_gn_bb(int b, int m, int k, int n, const float *w, int ldw, const float *x, long long bs_x, int ld_x,
                             const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y, long long bs_y,
                             int ld_y, const void *in_partial, int in_slices, int in_groups, const float *in_gamma,
                             const float *in_beta, float in_eps, int out_groups, void *out_partial, float *amax, int amax_rows,
                             const float *batch_bias, int ld_bb, const float *add, long long bs_add, int ld_add, void *stream);

#ifdef BDM_EXPERIMENTAL
/* bf16x6 form of the two entry points above (csrc/experimental/pointwise_s3.hip; measured: no faster, the GEMMs are staging-bound): the weights are split ONCE into exact bf16 triples
 * (bdm_pointwise_s3_pack_weights: packed = bdm_pointwise_s3_weight_elems(m, k) 16-bit elements), the activations while they are
 * staged; six partial products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulate: fp32-grade (no scale, no range limit)
 * at 2.7x the matrix rate of the fp32-MFMA kernel.  Same arguments and semantics otherwise; statistics slices are always
 * ceil(n / 128) * max(1, (m / groups) / 32) (the canonical layout).  Not for the skinny shapes (n <= 64, k >= 128: K-split kernel). */
size_t bdm_pointwise_s3_weight_elems(int m, int k);
int bdm_pointwise_s3_pack_weights(int m, int k, const float *w, int ldw, void *packed, void *stream);
int bdm_pointwise_conv_s3(int b, int m, int k, int n, const void *packed_w, const float *x, long long bs_x, int ld_x,
                          const float *bias, const float *batch_bias, int ld_bb, const float *residual, long long bs_r,
                          int ld_r, float *y, long long bs_y, int ld_y, int act, float slope, void *stream);
int bdm_pointwise_conv_gn_s3(int b, int m, int k, int n, const void *packed_w, const float *x, long long bs_x, int ld_x,
                             const float *x2, long long bs_x2, int ld_x2, int k1, const float *bias, float *y, long long bs_y,
                             int ld_y, const void *in_partial, int in_slices, int in_groups, const float *in_gamma,
                             const float *in_beta, float in_eps, int out_groups, void *out_partial, float *amax,
                             int amax_rows, void *stream);
#endif /* BDM_EXPERIMENTAL */
int bdm_max_over_neighbors_gn(int b, int c, int m, int u, const float *x, const void *in_partial, int in_slices, int groups,
                              const float *gamma, const float *beta, float eps, float *y, long long bs_y, int ld_y,
                              void *stream);
int bdm_max_over_neighbors(int b, int c, int m, int u, const float *x, float *y, long long bs_y,
                           int ld_y, void *stream);

/* BallQuery.forward's grouped tensor (modules/ball_query.py:16-30) in one launch:
 * out (b, 3+c, m, u) = cat[ grouping(coords, idx) - centers[..., None], grouping(features, idx) ].
 * workspace: NULL -> channel-first gather: for 256 <= n <= 8192 points a workgroup stages three channel rows of ALL the shape's points in
 * LDS (LDS-DMA, global_load_lds_dwordx4, when n % 256 == 0 and the rows are 16-byte aligned) and gathers from there (round 6), otherwise
 * one scattered load per element; else >= bdm_sa_group_workspace_bytes(b, c, n) bytes (16-byte aligned): [coords ; features] are first
 * repacked point-major so the gather reads 16-byte runs.  Same values in every form (a gather). */
size_t bdm_sa_group_workspace_bytes(int b, int c, int n);
int bdm_sa_group(int b, int c, int n, int m, int u, const float *coords, const float *centers,
                 const float *features, long long bs_f, int ld_f, const int *indices, float *out,
                 void *workspace, void *stream);

/* PointNetSAModule.forward (modules/pointnet.py:80-90) at the first level, WITHOUT the grouped tensor or either layer's output in
 * memory: grouping ([xyz - centre ; features], 1 <= c <= 32 feature rows), two SharedMLP layers (shared_mlp.py:11-37: Conv2d k=1 ->
 * GroupNorm(8) -> Swish; widths m1 = 32, m2 = 64), max over the u = 32 neighbours -> out (b, m2, m).  Four launches: the features
 * repacked point-major into `rows` (bdm_sa_mlp2_fused_rows_bytes(b, c, n) bytes, 16-byte aligned), then three passes that each
 * re-gather the neighbours' rows and recompute what they need (pass 1: GroupNorm-1 statistics; pass 2: GroupNorm-2 statistics;
 * pass 3: the output).  w1 (m1, 3 + c), w2 (m2, m1) as the convolutions hold them.  partial1 / partial2: fp64 scratch,
 * b * 8 * bdm_sa_mlp2_fused_slices(b, m) * 2 elements each.  Deterministic. */
size_t bdm_sa_mlp2_fused_rows_bytes(int b, int c, int n);
int bdm_sa_mlp2_fused_slices(int b, int m);
int bdm_sa_mlp2_fused(int b, int c, int n, int m, int u, int m1, int m2, const float *coords, const float *features,
                      long long bs_f, int ld_f, const float *centers, const int *indices, const float *w1, const float *b1,
                      const float *g1w, const float *g1b, float eps1, const float *w2, const float *b2, const float *g2w,
                      const float *g2b, float eps2, int groups, void *rows, void *partial1, void *partial2, float *out,
                      long long bs_o, int ld_o, void *stream);

/* y[b][ci][:] = v[b][ci]  -- t_emb[:, :, None].expand(-1, -1, N) (pvcnn.py:88) written into a concat slice. */
int bdm_broadcast_rows(int b, int c, int l, const float *v, int ld_v, float *y, long long bs_y,
                       int ld_y, void *stream);
/* strided row copy (building torch.cat operands in place) */
/* torch.cat([x0, x1], dim=1) in one launch.  A part with ld == 0 is a point-invariant column: element (shape, channel) at
 * x[shape * bs + channel], broadcast along l (the time embedding before the first SA level). */
int bdm_concat2_rows(int b, int l, int c0, const float *x0, long long bs_0, int ld_0, int c1, const float *x1, long long bs_1,
                     int ld_1, float *y, long long bs_y, int ld_y, void *stream);
int bdm_copy_rows(int b, int c, int l, const float *x, long long bs_x, int ld_x, float *y,
                  long long bs_y, int ld_y, void *stream);
/* (b, rows, cols) -> (b, cols, rows): PointCloudModel.forward's transposes (point_cloud_model.py:65). */
int bdm_transpose(int b, int rows, int cols, const float *x, float *y, void *stream);

/* get_timestep_embedding + embedf (pvcnn_utils.py:171-185, pvcnn.py:72-76,87-88):
 * t (b,) float -> out (b, dim) = Linear(LeakyReLU_0.1(Linear([sin, cos](t * freq)))). */
int bdm_time_embedding(int b, int dim, const float *t, const float *w0, const float *b0,
                       const float *w2, const float *b2, float *out, void *stream);
/* the same from int64 timesteps (what the schedulers hand over): t.float() inside the kernel */
int bdm_time_embedding_i64(int b, int dim, const long long *t, const float *w0, const float *b0,
                           const float *w2, const float *b2, float *out, void *stream);

/* Voxelization.forward's coordinate maths (modules/voxelization.py:16-25, normalize=True):
 * coords (b,3,n) -> norm_coords (b,3,n) float in [0, r-1], vox_coords (b,3,n) int32 (round half even). */
int bdm_voxel_coords(int b, int n, int r, float eps, const float *coords, float *norm_coords,
                     int *vox_coords, void *stream);

/* SE3d gate (modules/se.py:8-19, ReLU variant): gate (b,c) = sigmoid(w2 relu(w1 mean_l x)).
 * x (b,c,l) contiguous; mean_ws (b,c) scratch.  counters: NULL -> two launches (row means, then the FC layers); else b
 * ints that are ZERO on entry (and are left zero) -> one launch, the last workgroup of each shape runs the FC layers. */
int bdm_se_gate(int b, int c, int hidden, int l, const float *x, const float *w1, const float *w2,
                float *mean_ws, float *gate, int *counters, void *stream);

/* PVConv tail (pvconv.py:95-96): out = trilinear_devoxelize(grid * gate[:, :, None]) + add.
 * gate and add may be NULL. */
int bdm_devoxelize_gate_add(int b, int c, int n, int r, const float *coords, const float *grid,
                            const float *gate, const float *add, long long bs_a, int ld_a,
                            float *out, long long bs_o, int ld_o, void *stream);

/* Attention core (pvconv.py:46-55): out[c][i] = sum_j v[c][j] softmax_j(sum_c' q[c'][i] k[c'][j]),
 * no 1/sqrt(c) scale.  q, k, v share strides.  l <= 64: one workgroup per shape (global attention);
 * otherwise a flash-style MFMA kernel (c <= 64; the 16^3-token voxel attention).
 * workspace: NULL -> fp32-input MFMA kernel (only in `make EXPERIMENTAL=1` builds; BDM_ERR_UNSUPPORTED otherwise); else >= bdm_attention_workspace_bytes(b, c, l) bytes -> the bf16x6
 * kernel (q, k, v pre-split into exact bf16 triples, six partial products per fp32 product; fp32-grade accuracy). */
size_t bdm_attention_workspace_bytes(int b, int c, int l);
int bdm_attention_core(int b, int c, int l, const float *q, const float *k, const float *v,
                       long long bs_qkv, int ld_qkv, float *out, long long bs_o, int ld_o,
                       void *workspace, void *stream);
/* fp16x3 form (half the matrix work of the bf16x6 kernel behind bdm_attention_core): needs amax[3 s + 0..2] = max |q|, |k|, |v| of
 * shape s (the projection GEMM leaves them: bdm_pointwise_conv_gn amax with amax_rows = c) and a workspace of bdm_attention_h2_workspace_bytes;
 * 64 < l, c <= 64 */
size_t bdm_attention_h2_workspace_bytes(int b, int c, int l, int key_slices);
/* key_slices = key ranges per (shape, query tile), 1 .. 8: each range leaves an unnormalised partial result, merged in ascending key order
 * (deterministic).  The CALLER chooses the count and passes it to both functions (no environment look-up in the library: the workspace and
 * the launch cannot disagree, and the launch checks workspace_bytes).  bdm_attention_h2_key_slices(b, l) is the library's recommendation
 * for b shapes of l positions -- 1 when the (shape, query tile) items fill the chip, 2 or 4 for few shapes; a shape's bits depend on the
 * count, so a caller that wants them independent of the launch's batch passes a fixed one (host side: BDM_ATTN_KSPLIT, read once). */
int bdm_attention_h2_key_slices(int b, int l);
int bdm_attention_core_h2(int b, int c, int l, const float *q, const float *k, const float *v, long long bs_qkv, int ld_qkv,
                          const float *amax, float *out, long long bs_o, int ld_o, void *workspace, size_t workspace_bytes,
                          int key_slices, void *stream);

#ifdef BDM_EXPERIMENTAL  /* fp32-input MFMA convolution family (csrc/experimental/conv3d.hip): `make EXPERIMENTAL=1` */
/* nn.Conv3d(cin, cout, 3, padding=1) on (b, cin, r, r, r), r in {8, 16, 32} (pvconv.py:75-85).
 * packed_w = bdm_conv3d_pack_weights(w) : [27][cin][cout] from the module's (cout, cin, 3, 3, 3). */
int bdm_conv3d_pack_weights(int cout, int cin, const float *w, float *packed, void *stream);
int bdm_conv3d_3x3x3(int b, int cin, int cout, int r, const float *x, const float *packed_w,
                     const float *bias, float *y, void *stream);
/* Same convolution for an input that is a freshly voxelised point cloud (first Conv3d of a PVConv): rowocc
 * (b, r*r) uint8 flags the (x, y) grid rows holding at least one point (bdm_voxel_row_occupancy from the
 * voxeliser's cnt); work on all-zero operand rows is skipped.  Results are bit-identical to the dense call. */
int bdm_conv3d_3x3x3_sparse(int b, int cin, int cout, int r, const float *x, const float *packed_w,
                            const float *bias, const unsigned char *rowocc, float *y, void *stream);
#endif /* BDM_EXPERIMENTAL */
/* rowocc (b, r*r) uint8: grid row (x, y) holds an occupied cell (from the voxeliser's cnt) */
int bdm_voxel_row_occupancy(int b, int r, const int *cnt, unsigned char *rowocc, void *stream);

/* --- the same convolution at fp32 accuracy on the BF16 matrix cores ("bf16x6", conv3d_s3.hip) ---
 * Every fp32 operand is split exactly into three bf16 terms and the six leading partial products are accumulated in
 * fp32 (dropped terms <= 2^-23 of the product).  Activations travel in the S3 layout (b, ceil(c/8), 3, r^3, 8) bf16,
 * produced directly by the voxeliser and by the GroupNorm(+Swish) between the two convolutions of a PVConv. */
size_t bdm_conv3d_s3_weight_elems(int cout, int cin);   /* number of bf16 elements of the packed weights */
int bdm_conv3d_s3_pack_weights(int cout, int cin, const float *w, void *packed, void *stream);
int bdm_conv3d_3x3x3_s3(int b, int cin, int cout, int r, const void *x_s3, const void *packed_w,
                        const float *bias, float *y, void *stream);
/* x (b, c, v) fp32 contiguous -> S3 of GroupNorm(groups)(x) [Swish when act == 1]; groups == 0: plain conversion.
 * workspace: bdm_group_norm_workspace_bytes. */
int bdm_group_norm_to_s3(int b, int c, int v, int groups, const float *x, const float *gamma,
                         const float *beta, float eps, int act, void *out_s3, void *workspace, void *stream);
int bdm_group_norm_stats(int b, int c, int l, int groups, const float *x, long long bs_x, void *workspace,
                         int *slices_out, void *stream);
/* avg_voxelize_forward writing S3 (features strided: bs_f, ld_f); same deterministic summation order. */
int bdm_voxelize_plan(int b, int n, int r, const int *coords, int *ind, int *cnt, void *workspace, void *stream);
/* the plan plus bdm_voxel_compact and bdm_voxel_row_occupancy of the same shape in ONE launch (same values) */
int bdm_voxelize_plan_full(int b, int n, int r, int n_max, const int *coords, int *ind, int *cnt, void *workspace,
                           int *occ_index, int *occ_list, int *n_occ, unsigned char *rowocc, void *stream);
int bdm_avg_voxelize_s3(int b, int c, int n, int r, const float *features, long long bs_f, int ld_f,
                        const int *coords, void *out_s3, int *ind, int *cnt, void *workspace, void *stream);

/* fp16x3 form of the dense convolution (default for the second convolution of every PVConv): operands stored as two
 * fp16 terms (hi, lo) after exact power-of-two scaling, three partial products per fp32 product on
 * v_mfma_f32_16x16x32_f16 -- half the matrix work of bf16x6 at the same accuracy (csrc/conv3d_h2.hip).
 *   packed_w : bdm_conv3d_h2_weight_elems(cout, cin) fp16; inv_scale (cout floats) = 2^-e[co] for the epilogue;
 *              scale_ws (cout floats) is scratch of the pack.
 *   x_h2     : (b, ceil(c/8), 2, v, 8) fp16 = act_scale * swish(group_norm(x)) split in two, from
 *              bdm_group_norm_to_h2 (groups = 0: plain split of act_scale * x).  act_scale is a power of two chosen by
 *              the caller so that the scaled values sit high in fp16's range (|act_scale * x| saturates at 65504);
 *              the convolution receives x_inv_scale = 1 / act_scale. */
size_t bdm_conv3d_h2_weight_elems(int cout, int cin);
int bdm_conv3d_h2_pack_weights(int cout, int cin, const float *w, void *packed, float *scale_ws, float *inv_scale,
                               void *stream);
/* saturated (may be NULL): one device word that is OR-ed with 1 when any scaled value left fp16's range (|act_scale * y| >
 * 65504, or NaN) and was clamped by the split -- the accuracy guard of the fp16x3 path: the host polls the word once per
 * trajectory and re-routes that layer to the bf16x6 kernels (bdm_amd/ops.py: poll_h2_saturation). */
int bdm_group_norm_to_h2(int b, int c, int v, int groups, const float *x, const float *gamma, const float *beta,
                         float eps, int act, float act_scale, void *out_h2, void *workspace,
                         unsigned int *saturated, void *stream);
/* bdm_group_norm_to_h2 with the statistics given as `slices` slice partials per (shape, group) (b, groups, slices, 2 doubles)
 * instead of a statistics pass over x; channels per group >= 4 */
int bdm_group_norm_to_h2_stats(int b, int c, int v, int groups, const float *x, const float *gamma, const float *beta,
                               float eps, int act, float act_scale, void *out_h2, const void *partial, int slices,
                               unsigned int *saturated, void *stream);
/* bdm_group_norm_to_h2_stats for the COMPACT output of the first convolution (bdm_sparse_conv_dil, compact = 1): xc (b, n_dil_max, c)
 * one row per entry of the dilated voxel list, dil_index (b, v) row of voxel v or -1; every voxel outside the list has the value
 * bias[channel] (NULL: 0).  Same result as densifying xc first; the dense fp32 grid is never written or read. */
int bdm_group_norm_to_h2_stats_compact(int b, int c, int v, int groups, const float *xc, int n_dil_max, const int *dil_index,
                                       const float *bias, const float *gamma, const float *beta, float eps, int act,
                                       float act_scale, void *out_h2, const void *partial, int slices,
                                       unsigned int *saturated, void *stream);
int bdm_conv3d_3x3x3_h2(int b, int cin, int cout, int r, const void *x_h2, float x_inv_scale, const void *packed_w,
                        const float *inv_scale, const float *bias, float *y, void *stream);

/* GroupNorm-folded tail of a PVConv without attention (pvconv.py:84-96): the second convolution leaves its output raw and
 * the GroupNorm(groups) statistics of that output as slice partials (layout / size: bdm_group_norm_workspace_bytes;
 * *slices_out slices per (shape, group)); the consumers then normalise + Swish on the fly instead of rewriting the grid:
 *   bdm_se_gate_gn            per (shape, channel): coef (b, c, 2) = (gamma rstd, beta - mean gamma rstd), and the SE gate
 *                             sigmoid(w2 relu(w1 mean_l swish(coef.x x + coef.y))) (w1 == NULL: only coef is produced)
 *   bdm_devoxelize_gn_gate_add out = trilinear_devoxelize(swish(coef.x grid + coef.y) * gate) + add */
int bdm_conv3d_3x3x3_h2_gn(int b, int cin, int cout, int r, const void *x_h2, float x_inv_scale, const void *packed_w,
                           const float *inv_scale, const float *bias, float *y, int groups, void *gn_workspace,
                           int *slices_out, void *stream);
int bdm_se_gate_gn(int b, int c, int hidden, int l, int groups, const float *x, const void *gn_workspace, int slices,
                   const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *mean_ws,
                   float *coef, float *gate, void *stream);
int bdm_devoxelize_gn_gate_add(int b, int c, int n, int r, const float *coords, const float *grid, const float *coef,
                               const float *gate, const float *add, long long bs_a, int ld_a, float *out,
                               long long bs_o, int ld_o, void *stream);
/* Point branch of the PVConv folded in as well: bdm_se_gate_gn_pf also turns the slice partials that the branch's 1x1
 * convolution left (bdm_pointwise_conv_gn; pf_groups groups over the same c channels, pf_n points per row) into affine forms
 * pf_coef (b, c, 2), and bdm_devoxelize_gn_gate_add_pf adds Swish(pf_coef.x add + pf_coef.y) with add the RAW convolution
 * output -- the branch's GroupNorm needs no launch of its own. */
int bdm_se_gate_gn_pf(int b, int c, int hidden, int l, int groups, const float *x, const void *gn_workspace, int slices,
                      const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *mean_ws,
                      float *coef, float *gate, const void *pf_partial, int pf_slices, int pf_groups, int pf_n,
                      const float *pf_gamma, const float *pf_beta, float pf_eps, float *pf_coef, void *stream);
int bdm_devoxelize_gn_gate_add_pf(int b, int c, int n, int r, const float *coords, const float *grid, const float *coef,
                                  const float *gate, const float *add, long long bs_a, int ld_a, const float *add_coef,
                                  float *out, long long bs_o, int ld_o, void *stream);
/* the same with the SE gate computed inside the kernel from se_mean (b, c) = the channel means bdm_se_gate_gn(w1 = NULL) left:
 * one launch less per PVConv, bit-identical gate (same summation order as the separate FC kernel). hidden <= 64. */
int bdm_devoxelize_gn_se_add(int b, int c, int n, int r, const float *coords, const float *grid, const float *coef,
                             const float *se_mean, int hidden, const float *w1, const float *w2, const float *add,
                             long long bs_a, int ld_a, float *out, long long bs_o, int ld_o, void *stream);

/* ---- PVConv tail on the SMALL voxel grids (8^3 levels; csrc/pvconv_small.hip) -- per-shape workgroups, no hand-off between them ----
 * bdm_pvconv_tail_small: SE gate (both FC layers of se.py:8-19, from se_mean (b, c) = bdm_se_gate_gn(w1 = NULL)'s channel means)
 * + Swish(GroupNorm-2(grid)) * gate + trilinear devoxelisation at the n points + Swish(GroupNorm(point branch)) (add_coef; NULL: `add`
 * is added as it is) -> out (pvconv.py:91-97), one workgroup per (shape, 8 channels); bit-identical to bdm_se_gate_gn_pf +
 * bdm_devoxelize_gn_gate_add_pf.  With xh != NULL the same workgroups also form the NEXT PVConv's first-convolution operand on the same
 * voxel plan (cnt / plan_workspace / occ_list / n_occ / n_max of bdm_voxelize_plan_full): per occupied cell the mean of `out` over its
 * points (vox.cu:18-72 on the plan's ordered lists: the values of bdm_sparse_voxel_features_f32), times the power of two x_scale,
 * split into (hi, lo) fp16 records xh (b, c/8, 2, n_max) -- what bdm_sparse_split_h2 writes -- and amax_out (b) such that
 * bdm_sparse_conv_gemm_h2 derives x_scale from it; *saturated |= 1 when a scaled value leaves fp16's range. */
int bdm_pvconv_tail_small(int b, int c, int n, int r, int hidden, const float *coords, const float *grid, const float *coef,
                          const float *se_mean, const float *w1, const float *w2, const float *add, long long bs_a, int ld_a,
                          const float *add_coef, float *out, long long bs_o, int ld_o, const int *cnt, const void *plan_workspace,
                          const int *occ_list, const int *n_occ, int n_max, float x_scale, void *xh, float *amax_out, int *saturated,
                          void *stream);
#ifdef BDM_EXPERIMENTAL  /* csrc/experimental/sparse_gather_h2_small.hip: measured not faster */
/* bdm_sparse_conv_gather_h2_small: bdm_sparse_conv_gather_gn + bdm_group_norm_to_h2_stats in one launch, one workgroup per (shape,
 * GroupNorm group): y (b, n_max, 27, cout) = the GEMM's output, out_h2 (b, cout/8, 2, r^3) records = act_scale *
 * Swish(GroupNorm(groups)(conv1 output)) as two fp16 terms (the second convolution's operand); the dense fp32 output of the first
 * convolution is not written.  Needs 8 | cout / groups <= 64. */
int bdm_sparse_conv_gather_h2_small(int b, int cout, int r, int n_max, const float *y, const int *occ_index, const float *bias,
                                    int groups, const float *gamma, const float *beta, float eps, float act_scale, void *out_h2,
                                    unsigned int *saturated, void *stream);
#endif /* BDM_EXPERIMENTAL */

/* bf16x6 form of the two steps above (default): operands pre-split into exact bf16 triples ("S3" records of 8
 * channels x 16 bytes), GEMM on v_mfma_f32_32x32x16_bf16 with six partial products per fp32 product.
 *   xs (b, ceil(c/8), 3, n_max) records; ws (ceil(cin/8), 3, 27*cout) records = bdm_sparse_conv_s3_weight_elems bf16. */
int bdm_sparse_voxel_features_s3(int b, int c, int n, int r, int n_max, const float *features, long long bs_f,
                                 int ld_f, const int *cnt, const void *plan_workspace, const int *occ_list,
                                 const int *n_occ, void *xs, void *stream);
size_t bdm_sparse_conv_s3_weight_elems(int cout, int cin);
int bdm_sparse_conv_pack_weights_s3(int cout, int cin, const float *w, void *ws, void *stream);
int bdm_sparse_conv_gemm_s3(int b, int n_max, int cin, int n27, const void *xs, const void *ws, const int *n_occ,
                            float *y, void *stream);
/* + col_bias (b, n27; batch stride bs_cb floats): y[b][k][tap * cout + co] += col_bias[b][tap * cout + co] on every row -- the share of input channels that are
 * CONSTANT over a shape's occupied cells: the time embedding the reference concatenates to the features before the first PVConv of
 * set-abstraction levels 1.. (pvcnn.py:103; avg_voxelize of a per-shape constant is that constant on every occupied cell, vox.cu:18-72),
 * col_bias[b][tap][co] = sum_c W[co][cin + c][tap] t[b][c].  The gather adds a row once per OCCUPIED neighbour: exactly the
 * zero-padded convolution of the concatenated grid, with the GEMM's K and the feature pass reduced to the real feature channels. */
int bdm_sparse_conv_gemm_s3_cb(int b, int n_max, int cin, int n27, const void *xs, const void *ws, const int *n_occ,
                               const float *col_bias, long long bs_cb, float *y, void *stream);

/* Hoisted form of bdm_sparse_voxel_features* + bdm_sparse_conv_gemm* for an input whose feature channels are a gather of a per-image
 * map (the projection conditioning x_in[i] = [xyz_i, F[pix_i]], projection_model.py:179-231): with hmap (b, hw, n27) = F . Wf^T
 * computed once per trajectory (n27 = 27 * cout columns ordered tap * cout + co) and wx (n27, 3) the coordinate columns of the
 * weights,  y (b, n_max, n27)[k] = mean over the points i of occupied cell k (ascending i) of (hmap[pix_i] (0 where pix_i < 0) +
 * wx . xyz_i): the rows bdm_sparse_conv_gather consumes.  xyz (b, 3, n), pix (b, n) from bdm_rasterize_points. */
int bdm_sparse_conv_rows_from_map(int b, int n, int r, int n_max, int n27, int hw, const float *hmap, const int *pix,
                                  const float *xyz, const float *wx, const int *cnt, const void *plan_workspace,
                                  const int *occ_list, const int *n_occ, float *y, void *stream);

/* --- first convolution of a PVConv on the occupied voxels only (sparse_conv.hip) ---
 * out = Conv3d(avg_voxelize(features)) without materialising the dense grid:
 *   bdm_voxel_compact          cnt (b, r^3) -> occ_index (b, r^3) [-1 = empty], occ_list (b, n_max), n_occ (b)
 *   bdm_sparse_voxel_features  per-voxel mean features of the occupied cells, xc (b, c, n_max) (zeros beyond n_occ);
 *                              plan_workspace = the workspace bdm_voxelize_plan filled
 *   bdm_sparse_conv_gemm       y (b, n_max, 27*cout) = xc^T . wt,   wt = bdm_sparse_conv_pack_weights(w) (cin, 27*cout)
 *   bdm_sparse_conv_gather     out (b, cout, r^3) = bias + sum over taps (fixed order) of the occupied neighbours' y rows
 * Deterministic; equals bdm_conv3d_3x3x3(avg_voxelize) up to fp32 summation order. */
int bdm_voxel_compact(int b, int r, int n_max, const int *cnt, int *occ_index, int *occ_list, int *n_occ,
                      void *stream);
int bdm_sparse_voxel_features(int b, int c, int n, int r, int n_max, const float *features, long long bs_f,
                              int ld_f, const int *cnt, const void *plan_workspace, const int *occ_list,
                              const int *n_occ, float *xc, void *stream);
int bdm_sparse_conv_pack_weights(int cout, int cin, const float *w, float *wt, void *stream);
int bdm_sparse_conv_gemm(int b, int n_max, int cin, int n27, const float *xc, const float *wt,
                         const int *n_occ, float *y, void *stream);
int bdm_sparse_conv_gather(int b, int cout, int r, int n_max, const float *y, const int *occ_index,
                           const unsigned char *rowocc, const float *bias, float *out, void *stream);
/* the same gather + the GroupNorm(groups) statistics of its output as r*r slice partials per (shape, group):
 * gn_partial (b, groups, r*r, 2 doubles); consumed by bdm_group_norm_to_h2_stats */
int bdm_sparse_conv_gather_gn(int b, int cout, int r, int n_max, const float *y, const int *occ_index,
                              const unsigned char *rowocc, const float *bias, float *out, int groups, void *gn_partial,
                              void *stream);

/* --- fp32 feature records of the occupied cells (sparse_conv_h2.hip), and the same first convolution in ONE kernel, no
 *     (n_occ x 27*cout) intermediate (experimental: correct and deterministic but slower than GEMM + gather) ---
 *   bdm_sparse_voxel_features_f32  occupied cells' mean features as fp32 records xr (b, ceil(c/8), n_max) x 8 channels,
 *                                  rows >= n_occ zero; amax[i] (b slots, ZERO on entry) receives max |value| of shape i
 *   bdm_sparse_conv_fused_pack_weights  (cout, cin, 3,3,3) fp32 -> [ceil(cin/8)][27][2][cout] records of 8 fp16
 *                                  (hi / lo of w * 2^e[co]); inv_scale[co] = 2^-e[co]; scale_ws (cout floats) scratch
 *   bdm_sparse_conv_fused          out (b, cout, r^3) = bias + conv: workgroup = (shape, slab of output x-planes, 32 output
 *                                  channels) with the slab's accumulators in LDS; occupied cells (sorted by voxel index:
 *                                  occ_list of bdm_voxelize_plan_full) are multiplied on the matrix cores (fp16x3, activation
 *                                  scale derived on the device from amax) and scattered tap by tap, phases separated by
 *                                  barriers: deterministic.  r in {8, 16, 32}. */
int bdm_sparse_voxel_features_f32(int b, int c, int n, int r, int n_max, const float *features, long long bs_f,
                                  int ld_f, const int *cnt, const void *plan_workspace, const int *occ_list,
                                  const int *n_occ, void *xr, float *amax, void *stream);
#ifdef BDM_EXPERIMENTAL  /* one-kernel form (csrc/experimental/sparse_conv_fused.hip): `make EXPERIMENTAL=1` */
size_t bdm_sparse_conv_fused_weight_elems(int cout, int cin);
int bdm_sparse_conv_fused_pack_weights(int cout, int cin, const float *w, void *packed, float *scale_ws,
                                       float *inv_scale, void *stream);
int bdm_sparse_conv_fused(int b, int cin, int cout, int r, int n_max, const void *xr, const float *amax,
                          const void *packed_w, const float *inv_scale, const int *occ_list, const int *n_occ,
                          const float *bias, float *out, void *stream);
#endif /* BDM_EXPERIMENTAL */

/* fp16x3 form of bdm_sparse_conv_gemm_s3 (the default GEMM of the first convolution): half the matrix work.  xr / amax from
 * bdm_sparse_voxel_features_f32, split once into (hi, lo) fp16 records xh (b, ceil(cin/8), 2, n_max) by bdm_sparse_split_h2;
 * weights [ceil(cin/8)][2][27*cout] fp16 records with per-output-channel scale
 * (bdm_sparse_conv_pack_weights_h2; inv_scale (cout floats)); y (b, n_max, 27*cout) fp32 as the other GEMM forms. */
size_t bdm_sparse_conv_h2_weight_elems(int cout, int cin);
int bdm_sparse_conv_pack_weights_h2(int cout, int cin, const float *w, void *packed, float *scale_ws,
                                    float *inv_scale, void *stream);
int bdm_sparse_split_h2(int b, int cin, int n_max, const void *xr, const float *amax, void *xh, void *stream);
int bdm_sparse_conv_gemm_h2(int b, int n_max, int cin, int cout, const void *xh, const float *amax, const void *packed_w,
                            const float *inv_scale, const int *n_occ, float *y, void *stream);
/* + the per-shape column addend of bdm_sparse_conv_gemm_s3_cb (col_bias (b, 27 * cout)) */
int bdm_sparse_conv_gemm_h2_cb(int b, int n_max, int cin, int cout, const void *xh, const float *amax, const void *packed_w,
                               const float *inv_scale, const int *n_occ, const float *col_bias, long long bs_cb, float *y, void *stream);

/* The same first convolution as ONE output-stationary implicit GEMM with tap skipping (sparse_conv_os.hip, round 4; the default
 * wherever the input is not the hoisted conditioning map): no (n_occ x 27*cout) intermediate, no gather, no operand-split pass.  Only
 * the voxels that can differ from the bias -- the once-dilated occupied set, listed in voxel order -- are computed, in tiles of
 * consecutive list entries whose occupied neighbours are ONE contiguous range of compact rows (staged in LDS per 8-channel chunk).
 *   bdm_voxel_dilate        cnt (b, r^3) -> dil_list (b, n_dil_max) voxel ids of the dilated set in ascending order (n_dil_max = r^3
 *                           always suffices), dil_index (b, r^3) rank in that list or -1, plane_start (b, r + 2): occupied cells in
 *                           x-planes < x, tile_start (b, bdm_voxel_dilate_slices(r, tile), 16): per tile [first entry, end entry, first voxel of
 *                           the linear range it owns, its end, first compact row of its input range, rows, 0, live tiles of the
 *                           shape].  Depends on (coords, r) only: part of the voxel plan of a level.  r in {8, 16, 32}.
 *   bdm_sparse_conv_dil     xr / amax: bdm_sparse_voxel_features_f32 (fp32 records (b, ceil(cin/8), n_max) x 8 channels + per-shape
 *                           max |value|); occ_index (b, r^3) compact row of every cell or -1; packed_w / inv_scale:
 *                           bdm_conv3d_h2_pack_weights (the dense fp16x3 convolution's weight image).  fp16x3 arithmetic, activation
 *                           scale per shape.  compact = 1: y (b, n_dil_max, cout), one row per list entry (every other voxel of the
 *                           grid equals bias: bdm_group_norm_to_h2_stats_compact consumes this form); compact = 0: y (b, cout, r^3),
 *                           every cell written.
 *                           work_counter: one int, ZERO on entry (left non-zero): the persistent workgroups (one per CU) pull their
 *                           (tile, channel block, shape) items from it.
 *   bdm_sparse_conv_dil_gn  also leaves GroupNorm(groups) partials of the DENSE output (bias voxels included) in gn_partial
 *                           (b, groups, tiles, 2 doubles), *slices_out = tiles = bdm_voxel_dilate_slices(r, tile).
 * `tile` (round 6) selects the tile form of a plan and of the convolutions that run on it (the two must agree):
 *   0            FULL tiles (<= 512 / 256 / 128 entries at r = 32 / 16 / 8, one row range of <= 3 r^2 rows, one workgroup per CU);
 *   64/128/256   HALF tiles (EXPERIMENTAL=1 builds only: measured not faster, profiles/r06_sparse_dil_half_tiles.txt; the default library
 *                refuses them): that many entries, four waves and 80 KB of LDS per workgroup, TWO workgroups per CU; a tile
 *                inside one x-plane lists three row ranges ((x-1, x, x+1) x y-rows y0-1 .. y1+1), a tile across planes one range of
 *                whole planes; never more than 1376 rows (<= 3 tile + 12 r inside a plane; across planes only when they fit).
 * A tile record is 16 ints: [first entry, end entry, first voxel of the owned linear range, its end, range 0 first row, range 0 rows,
 * (first x-plane << 8) | last, live tiles of the shape, range 1 first row, rows, range 2 first row, rows, tile, 0, 0, 0]. */
int bdm_voxel_dilate_slices(int r, int tile);
int bdm_voxel_dilate(int b, int r, int n_dil_max, const int *cnt, int *dil_list, int *dil_index, int *plane_start, int *tile_start,
                     int tile, void *stream);
int bdm_sparse_conv_dil(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                        const int *occ_index, const int *dil_list, const int *dil_index, const int *tile_start,
                        const void *packed_w, const float *inv_scale, const float *bias, float *y, int compact, int tile,
                        int *work_counter, void *stream);
int bdm_sparse_conv_dil_gn(int b, int cin, int cout, int r, int n_max, int n_dil_max, const void *xr, const float *amax,
                           const int *occ_index, const int *dil_list, const int *dil_index, const int *tile_start,
                           const void *packed_w, const float *inv_scale, const float *bias, float *y, int compact, int groups,
                           void *gn_partial, int *slices_out, int tile, int *work_counter, void *stream);

/* --- the whole voxel branch of a PVConv without dense grids (pvconv_compact.hip, sparse_conv_os.hip; pvconv.py:74-97) ---
 * After GroupNorm + Swish the first convolution's output is a per-channel CONSTANT outside the once-dilated set D1, so the second
 * convolution differs from one of 27 per-class constants (class = 9 cx + 3 cy + cz, c = 0 / 1 / 2: first plane / interior / last
 * plane of the axis -- the zero padding removes taps there) only on the twice-dilated set D2, and the devoxelisation only reads D1.
 *   bdm_voxel_dilate_again        D2 from D1's ranks (dil_index of bdm_voxel_dilate): list, ranks, tiles (input ranges = rows of D1),
 *                                 class_count (b, 27): voxels outside D2 per class
 *   bdm_group_norm_to_h2_rows     GroupNorm + Swish + fp16 split of the FIRST convolution's output on the rows of D1: x = compact rows
 *                                 (b, n_rows_max, c) [dense_in = 0] or the grid (b, c, v) read at dil_list [dense_in = 1]; statistics
 *                                 from `partial` (b, groups, slices, 2).  -> rows_h2 (b, ceil(c/8), 2, n_rows_max) records of 8 fp16,
 *                                 const_h2 (b, ceil(c/8), 2) records and const_f32 (b, c): the value outside D1 (from bias)
 *   bdm_sparse_conv_dil_h2_gn     the SECOND convolution on D2's tiles from those rows: y (b, n_dil_max, cout) compact rows; GroupNorm
 *                                 partials (b, groups, slices, 2), *slices_out = bdm_voxel_dilate_slices(r, tile) + 27 (the tiles'; the last 27
 *                                 are left for bdm_conv3d_class_constants); work_counter as bdm_sparse_conv_dil
 *   bdm_conv3d_class_weight_sums  (cout, cin, 3,3,3) -> wsum (27, cin, cout) doubles: per class, the sum of the taps inside the grid
 *   bdm_conv3d_class_constants    class_vals (b, 27, cout) = bias + wsum[k] . const_f32[b] (fp64, rounded once) and slices
 *                                 slice0 .. slice0 + 26 of gn_partial: class_count x (sum, sum of squares) per group
 *   bdm_se_gate_gn_rows(_pf)      bdm_se_gate_gn(_pf) from the rows of D2 + counts x class constants: coef (b, c, 2), mean_ws (b, c),
 *                                 gate (b, c) (w1 = NULL: no FC layers); part_ws: bdm_se_gate_gn_rows_workspace_elems(b, c) floats
 *   bdm_devoxelize_gn_gate_add_rows(_pf)  bdm_devoxelize_gn_gate_add(_pf) reading the 8 corner rows through D2's ranks */
int bdm_voxel_dilate_again(int b, int r, int n_dil_max, const int *dil_index_in, int *dil_list, int *dil_index, int *plane_start,
                           int *tile_start, int *class_count, int tile, void *stream);
int bdm_group_norm_to_h2_rows(int b, int c, int v, int groups, const float *x, int dense_in, int n_rows_max, const int *dil_list,
                              const int *tile_start, int tiles_max, const float *bias, const float *gamma, const float *beta,
                              float eps, int act, float act_scale, void *rows_h2, void *const_h2, float *const_f32,
                              const void *partial, int slices, unsigned int *saturated, void *stream);
int bdm_sparse_conv_dil_h2_gn(int b, int cin, int cout, int r, int n_rows_max, int n_dil_max, const void *rows_h2, const void *xconst,
                              float x_inv_scale, const int *in_index, const int *dil_list, const int *dil_index,
                              const int *tile_start, const void *packed_w, const float *inv_scale, const float *bias, float *y,
                              int groups, void *gn_partial, int *slices_out, int tile, int *work_counter, void *stream);
size_t bdm_conv3d_class_weight_elems(int cout, int cin);
int bdm_conv3d_class_weight_sums(int cout, int cin, const float *w, void *wsum, void *stream);
int bdm_conv3d_class_constants(int b, int cin, int cout, const void *wsum, const float *bias, const float *const_f32,
                               const int *class_count, float *class_vals, int groups, void *gn_partial, int slices, int slice0,
                               void *stream);
size_t bdm_se_gate_gn_rows_workspace_elems(int b, int c);
int bdm_se_gate_gn_rows(int b, int c, int hidden, int v, int groups, const float *rows, int n_rows_max, const int *tile_start,
                        int tiles_max, const float *class_vals, const int *class_count, const void *gn_partial, int slices,
                        const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *part_ws,
                        float *mean_ws, float *coef, float *gate, void *stream);
int bdm_se_gate_gn_rows_pf(int b, int c, int hidden, int v, int groups, const float *rows, int n_rows_max, const int *tile_start,
                           int tiles_max, const float *class_vals, const int *class_count, const void *gn_partial, int slices,
                           const float *gamma, const float *beta, float eps, const float *w1, const float *w2, float *part_ws,
                           float *mean_ws, float *coef, float *gate, const void *pf_partial, int pf_slices, int pf_groups, int pf_n,
                           const float *pf_gamma, const float *pf_beta, float pf_eps, float *pf_coef, void *stream);
int bdm_devoxelize_gn_gate_add_rows(int b, int c, int n, int r, const float *coords, const float *rows, int n_rows_max,
                                    const int *dil_index, const float *class_vals, const float *coef, const float *gate,
                                    const float *add, long long bs_a, int ld_a, float *out, long long bs_o, int ld_o, void *stream);
int bdm_devoxelize_gn_gate_add_rows_pf(int b, int c, int n, int r, const float *coords, const float *rows, int n_rows_max,
                                       const int *dil_index, const float *class_vals, const float *coef, const float *gate,
                                       const float *add, long long bs_a, int ld_a, const float *add_coef, float *out, long long bs_o,
                                       int ld_o, void *stream);

/* ------------------------------------------------------------------------------------
 * 3. Per-step glue of the coupled DDPM loop
 * ---------------------------------------------------------------------------------- */

/* PC^2 scheduler step: diffusers 0.21.0 DDPMScheduler.step (epsilon prediction, fixed_small
 * variance, clip_sample=False), call sites experiments/model/model.py:193,286,563.
 *   x0   = (x - sqrt_beta_prod * eps) / sqrt_alpha_prod
 *   out  = coef_x0 * x0 + coef_x * x  [+ sigma * noise   when noise != NULL, i.e. t > 0]
 * The five scalars are the scheduler's per-timestep float32 coefficients (host side).  `out` may be `x` (elementwise: the reverse
 * loops step in place); the same holds for bdm_ddpm_step_philox / bdm_pvd_step_philox. */
int bdm_ddpm_step(long long n, const float *x, const float *eps, const float *noise,
                  float sqrt_beta_prod, float sqrt_alpha_prod, float coef_x0, float coef_x,
                  float sigma, float *out, void *stream);
/* The same step with {sqrt_beta_prod, sqrt_alpha_prod, coef_x0, coef_x, sigma} read from device memory (coef[5]), so
 * that ONE captured hipGraph of a reverse step can be replayed for every timestep; sigma == 0 (t == 0) adds no noise.
 * out may alias x. */
int bdm_ddpm_step_dev(long long n, const float *x, const float *eps, const float *noise, const float *coef,
                      float *out, void *stream);

/* DDIM step (diffusers 0.21.0 DDIMScheduler.step; the reference's schedulers_map['ddim'], model/model.py:60):
 *   x0 = (x - sqrt_beta_prod * eps) / sqrt_alpha_prod;  out = coef_x0 * x0 + coef_eps * eps [+ sigma * noise when eta > 0] */
int bdm_ddim_step(long long n, const float *x, const float *eps, const float *noise, float sqrt_beta_prod,
                  float sqrt_alpha_prod, float coef_x0, float coef_eps, float sigma, float *out, void *stream);

/* out = (c0*x0 [+ c1*x1 [+ c2*x2 [+ c3*x3]]]) / div over n floats, k <= 4 terms, evaluated left to right: the linear
 * multistep combinations and the transfer step of the PNDM scheduler (diffusers 0.21.0 PNDMScheduler, the reference's
 * schedulers_map['pndm'], model/model.py:61).  out may alias any input. */
int bdm_lincomb(long long n, int k, float c0, const float *x0, float c1, const float *x1, float c2,
                const float *x2, float c3, const float *x3, float div, float *out, void *stream);

/* PVD scheduler step: GaussianDiffusion.p_sample (experiments/pvd/__init__.py:136-224):
 *   x0 = sqrt_recip_abar * x - sqrt_recipm1_abar * eps;  mean = coef1 * x0 + coef2 * x;
 *   out = mean + sigma * noise      (sigma = 0 at t == 0; noise is always drawn, as the reference does).  `out` may be `x` (elementwise). */
int bdm_pvd_step(long long n, const float *x, const float *eps, const float *noise,
                 float sqrt_recip_abar, float sqrt_recipm1_abar, float coef1, float coef2,
                 float sigma, float *out, void *stream);

/* Per-shape counter-based random streams (rng_ops.hip).  The reference draws the initial cloud, the DDPM / PVD noise and
 * the blend masks from ONE per-process generator seeded seed + rank (training_utils.py:373-378; draw sites
 * main_blending.py:228,330-338, model/model.py:286, pvd/__init__.py:213,232), so a shape's sample depends on its rank and
 * batch slot.  Here element e of draw `draw` of purpose `purpose` of the shape with key keys[s] is
 *   Philox4x32-10(key = keys[s], counter = (e / 4, 0, draw, purpose))[e % 4]
 * (keys[s] is derived on the host from (run seed, GLOBAL shape index)): rank-count- and batch-invariant, one launch per
 * batch.  Normals: Box-Muller on 24-bit uniforms.  out (b, per_shape) float / int64 (bit 0 of each word: Bernoulli 1/2,
 * the `torch.randint(0, 2, ...)` of main_blending.py:330-338). */
int bdm_philox_normal(int b, long long per_shape, const unsigned long long *keys, unsigned int draw,
                      unsigned int purpose, float *out, void *stream);
int bdm_philox_bits(int b, long long per_shape, const unsigned long long *keys, unsigned int draw,
                    unsigned int purpose, long long *out, void *stream);
/* bdm_ddpm_step / bdm_pvd_step with the noise of bdm_philox_normal(b, per_shape, keys, draw, purpose) generated inside the
 * kernel (never written to memory); same bits as the two-launch form.  DDPM: sigma == 0 (t == 0) uses no draw. */
int bdm_ddpm_step_philox(int b, long long per_shape, const float *x, const float *eps,
                         const unsigned long long *keys, unsigned int draw, unsigned int purpose,
                         float sqrt_beta_prod, float sqrt_alpha_prod, float coef_x0, float coef_x, float sigma,
                         float *out, void *stream);
int bdm_pvd_step_philox(int b, long long per_shape, const float *x, const float *eps,
                        const unsigned long long *keys, unsigned int draw, unsigned int purpose,
                        float sqrt_recip_abar, float sqrt_recipm1_abar, float coef1, float coef2, float sigma,
                        float *out, void *stream);

/* x (b, n, 3) point-major: subtract the per-shape mean over points, in place
 * (main_blending.py:229; model/model.py:530-531). */
int bdm_center_points(int b, int n, float *x, void *stream);

/* BDM-Blending per-point select (main_blending.py:326-344):
 * out[p] = mask[p] ? prior[p] : recon[p] for num_points = B*N points of 3 floats; mask int64. */
int bdm_blend_select(long long num_points, const float *recon, const float *prior,
                     const long long *mask, float *out, void *stream);

/* Projection conditioning, per-step part (model/projection_model.py:127-157; pytorch3d
 * PerspectiveCameras + naive PointsRasterizer, radius in NDC, one point per pixel).
 * points (b,n,3) point-major world coordinates; cameras (b,16) = R row-major (9), T (3),
 * focal (2), principal point (2).  pix_of_point (b,n): flat index h*W+w of the LAST pixel
 * (row-major) owned by the point, or -1. */
size_t bdm_rasterize_workspace_bytes(int b, int h, int w);
int bdm_rasterize_points(int b, int n, int h, int w, float radius, const float *points,
                         const float *cameras, int *pix_of_point, void *workspace, void *stream);
/* get_input_with_conditioning's output (projection_model.py:179-231):
 * out (b, n, 3+c) = cat[x_t, feature_image[pix_of_point]] with zeros for points owning no pixel;
 * feature_image is stored pixel-major (b, h*w, c). */
int bdm_condition_gather(int b, int n, int c, int hw, const float *x_t, const float *feature_image,
                         const int *pix_of_point, float *out, void *stream);
/* the same, written channel-first: out (b, 3 + c, n) -- the layout the denoiser consumes (no transpose pass) */
int bdm_condition_gather_cf(int b, int n, int c, int hw, const float *x_t, const float *feature_image,
                            const int *pix_of_point, float *out, void *stream);
/* rows 0..2 (the coordinates) of that tensor only; rows 3.. stay UNWRITTEN until bdm_condition_gather_cf completes the same tensor.  For
 * the reverse loop when every consumer of the feature rows reads a hoisted per-pixel map instead (100 MB per step at b = 16). */
int bdm_condition_xyz_cf(int b, int n, int c, const float *x_t, float *out, void *stream);

/* Quality metrics of the evaluation scripts (evaluation/evaluation_cd.py:111-131, evaluation_f1.py:90-110):
 * out (b, n) = min over the m target points of the squared distance; src (b,n,3), tgt (b,m,3) point-major. */
int bdm_nn_sqdist(int b, int n, int m, const float *src, const float *tgt, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * 4. Image encoder of the projection conditioning (ViT-S/16), once per image batch
 *    (experiments/model/feature_model.py:85-132; timm VisionTransformer; hoisted out of the per-step loop)
 *    Tokens are channel-first (b, d, t): Linear layers = bdm_pointwise_conv (bias / GELU / residual fused),
 *    attention = bdm_attention_core per head.
 * ---------------------------------------------------------------------------------- */
/* (b,3,h,w) in [0,1] -> ImageNet-normalised patches (b, 3*p*p, t), row k = c*p*p + py*p + px; mean3/std3 are HOST arrays */
int bdm_vit_patchify(int b, int h, int w, int patch, const float *host_mean3, const float *host_std3, const float *img,
                     float *out, void *stream);
/* out (b, d, t+1): column 0 = cls + pos[0], column 1+i = patches[:, :, i] + pos[1+i]; pos token-major (t+1, d) */
int bdm_vit_assemble_tokens(int b, int d, int t, const float *patches, const float *cls, const float *pos,
                            float *out, void *stream);
/* nn.LayerNorm(d) over the channel axis of (b, d, t) */
int bdm_layer_norm_channels(int b, int d, int t, const float *x, const float *gamma, const float *beta, float eps,
                            float *y, void *stream);
/* conditioning image (b, h*w, 3+d) pixel-major = cat[(img - colors_mean)/colors_std, bilinear(tokens[:, :, 1:] as grid x grid)]
 * with align_corners=False (projection_model.py:110-125 + feature_model.py:107-119) */
int bdm_vit_conditioning_image(int b, int d, int grid, int h, int w, float colors_mean, float colors_std,
                               const float *tokens, const float *img, float *out, void *stream);


/* ------------------------------------------------------------------------------------
 * 5. Step executor: one recorded reverse step replayed by ONE call
 *    Replaces the per-step Python of the reference's reverse loop (experiments/model/model.py:275-287: conditioning ->
 *    denoiser -> scheduler.step, ~230 dependent launches) once its arguments are static: the host framework records the
 *    step's C-ABI calls (function name + arguments, 8 bytes per argument: integers and pointers as they are, float / double
 *    arguments as the bit pattern of a double), its memsets / device copies and its stream / event edges, and replays the
 *    list from C.  No allocation, no synchronisation, nothing Python per launch.  The tape does NOT own the buffers whose
 *    addresses it holds: the recorder keeps them alive (bdm_amd/tape.py).
 * ---------------------------------------------------------------------------------- */
void *bdm_tape_create(void);
void bdm_tape_destroy(void *tape);
int bdm_tape_length(const void *tape);
/* append `function` (an `int bdm_*(...)` entry point of this header, by name) with n_args 8-byte argument slots */
int bdm_tape_append_call(void *tape, const char *function, const unsigned long long *args, int n_args);
int bdm_tape_append_memset(void *tape, void *dst, int byte_value, size_t bytes, void *stream);
int bdm_tape_append_memcpy(void *tape, void *dst, const void *src, size_t bytes, void *stream);
/* `waiter` waits for everything enqueued on `other` up to this point of the replay (torch's Stream.wait_stream) */
int bdm_tape_append_wait_stream(void *tape, void *waiter, void *other);
int bdm_tape_append_event_record(void *tape, void *event, void *stream);
int bdm_tape_append_event_wait(void *tape, void *stream, void *event);
/* replay entries [first, first + count) (count < 0: to the end); returns 0 or the status of the first failing entry, whose
 * index bdm_tape_failed_entry reports (-1: none) */
int bdm_tape_replay(void *tape, int first, int count);
int bdm_tape_failed_entry(const void *tape);
/* test hook of the argument marshalling (no GPU work): writes its arguments, converted to double, to out16 (host memory) */
int bdm_tape_echo(int a, long long b, float c, const void *d, unsigned int e, float f, int g, void *out16);

#ifdef __cplusplus
}
#endif
#endif /* BDM_HIP_H */
