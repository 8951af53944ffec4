// pointwise_common.h -- argument block shared by the 1x1-convolution GEMM kernels (dense_ops.hip: fp32 MFMA; pointwise_s3.hip: bf16x6)
#pragma once
struct PwGn {
  const double *in_partial;  // (b, in_G, in_S, 2) or NULL
  int in_S, in_G;
  const float *in_gamma, *in_beta;
  float in_eps;
  double *out_partial;       // (b, M / out_cg, S_out, 2) or NULL;  S_out = ceil(N / 128) * max(1, out_cg / 32): canonical, tile-independent
  int out_cg;
  // second source of the K axis: rows k >= k1 of the operand come from x2 (torch.cat([x, x2], dim=1) without the copy)
  const float *x2;
  long long bsx2;
  int ldx2, k1;
  // max |y| per (shape, block of `amax_rows` output rows) (amax_rows % 32 == 0): slot [shape * amax_slots + block], bit patterns
  // of non-negative floats combined with an integer atomicMax (order independent).  Zero on entry.  Feeds the fp16x3 attention's
  // scales; per SHAPE so that a shape's result does not depend on its batch-mates.
  unsigned *amax;
  int amax_rows, amax_slots;
};

// tile of the 1x1 GEMMs for a (b, m, k, n) problem: (32 mi) x (128 ni), K chunk bk (dense_ops.hip)
void bdm_pw_tile(int b, int m, int k, int n, int *mi, int *ni, int *bk);
