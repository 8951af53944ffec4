// s3_split.h -- the exact three-way bf16 split ("S3") shared by the bf16x6 kernels (conv3d_s3.hip, sparse_conv.hip).
//   x = hi + mid + lo  with 8 + 8 + 8 mantissa bits: hi and mid by truncation (the remainders are exact in fp32),
//   lo rounded to nearest even.  A record is 8 consecutive channels of one position = 8 bf16 = 16 bytes.
#pragma once
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
  const unsigned u = __float_as_uint(x);
  h = (unsigned short)(u >> 16);
  const float r1 = x - __uint_as_float(u & 0xFFFF0000u);  // exact
  const unsigned u1 = __float_as_uint(r1);
  m = (unsigned short)(u1 >> 16);
  const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);  // exact
  const unsigned u2 = __float_as_uint(r2);
  l = (unsigned short)((u2 + 0x7FFFu + ((u2 >> 16) & 1u)) >> 16);  // round to nearest even (r2 is finite)
}

__device__ __forceinline__ void store_s3(unsigned short *base, size_t rec, size_t split_stride, const float v[8]) {
  // base points at split 0 of this (shape, channel group); records are 8 bf16 = 16 bytes
  unsigned short h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3(v[j], h[j], m[j], l[j]);
  uint4 ph, pm, pl;
  ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16); ph.z = h[4] | (h[5] << 16); ph.w = h[6] | (h[7] << 16);
  pm.x = m[0] | (m[1] << 16); pm.y = m[2] | (m[3] << 16); pm.z = m[4] | (m[5] << 16); pm.w = m[6] | (m[7] << 16);
  pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16); pl.z = l[4] | (l[5] << 16); pl.w = l[6] | (l[7] << 16);
  uint4 *o = reinterpret_cast<uint4 *>(base);
  o[rec] = ph;
  o[split_stride + rec] = pm;
  o[2 * split_stride + rec] = pl;
}
