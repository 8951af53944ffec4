// vit_ops.hip -- the image encoder of the projection conditioning (ViT-S/16, MSN) on the HIP path.
// Reference: experiments/model/feature_model.py:85-132 (timm 0.9.7 VisionTransformer, global_pool='', num_classes=0)
// and experiments/model/projection_model.py:110-125 (get_local_conditioning).  Runs ONCE per image batch (hoisted out
// of the per-step loop).  Tokens are kept channel-first (B, D, T) so every Linear is the pointwise MFMA GEMM of
// dense_ops.hip (with fused bias / GELU / residual) and attention is the flash kernel (one call per head).
// The final kernel fuses: drop CLS, bilinear 14x14 -> HxW upsampling (align_corners=False), colour normalisation and
// the transposition to the pixel-major (B, H*W, 3 + D) conditioning image the per-step gather reads.
#include "../../include/bdm_hip.h"
#include "common.h"

using namespace bdm;

// img (B,3,H,W) in [0,1] -> patches (B, 3*p*p, T): k = c*p*p + py*p + px (the Conv2d weight's flattening), ImageNet-normalised
__global__ void vit_patchify_kernel(int H, int W, int p, float m0, float m1, float m2, float s0, float s1, float s2,
                                    const float *__restrict__ img, float *__restrict__ out) {
  const int tw = W / p, T = (H / p) * tw, K = 3 * p * p;
  const int bi = blockIdx.z, k = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int c = k / (p * p), py = (k / p) % p, px = k % p, ty = t / tw, tx = t % tw;
  const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
  const float v = img[(((size_t)bi * 3 + c) * H + ty * p + py) * W + tx * p + px];
  out[((size_t)bi * K + k) * T + t] = (v - mean) / sd;
}
extern "C" int bdm_vit_patchify(int b, int h, int w, int patch, const float *mean3, const float *std3, const float *img,
                                float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && h % patch == 0 && w % patch == 0, "vit_patchify: image %dx%d not divisible by patch %d", h, w, patch);
  if (b == 0) return BDM_OK;
  const int T = (h / patch) * (w / patch);
  hipLaunchKernelGGL(vit_patchify_kernel, dim3(cdiv(T, 64), 3 * patch * patch, b), dim3(64), 0, (hipStream_t)stream, h, w, patch,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], img, out);
  return launch_status("vit_patchify");
}

// tokens (B, D, T+1): column 0 = cls + pos[0], column 1+t = patch_embed[t] + pos[1+t];  pos is token-major (T+1, D)
__global__ void vit_assemble_kernel(int D, int T, const float *__restrict__ patches, const float *__restrict__ cls,
                                    const float *__restrict__ pos, float *__restrict__ out) {
  const int bi = blockIdx.z, d = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > T) return;
  const float v = t == 0 ? cls[d] : patches[((size_t)bi * D + d) * T + t - 1];
  out[((size_t)bi * D + d) * (T + 1) + t] = v + pos[(size_t)t * D + d];
}
extern "C" int bdm_vit_assemble_tokens(int b, int d, int t, const float *patches, const float *cls, const float *pos,
                                       float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && d >= 1 && t >= 1, "vit_assemble_tokens: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(cdiv(t + 1, 64), d, b), dim3(64), 0, (hipStream_t)stream, d, t, patches, cls, pos, out);
  return launch_status("vit_assemble_tokens");
}

// nn.LayerNorm(D) over the channel axis of channel-first tokens (B, D, T): one thread per token
__global__ void layer_norm_channels_kernel(int D, int T, const float *__restrict__ x, const float *__restrict__ gamma,
                                           const float *__restrict__ beta, float eps, float *__restrict__ y) {
  const int bi = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float *xb = x + (size_t)bi * D * T + t;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s += xb[(size_t)d * T];
  const float mean = s / D;
  float q = 0.f;
  for (int d = 0; d < D; ++d) { const float a = xb[(size_t)d * T] - mean; q += a * a; }
  const float rstd = 1.0f / sqrtf(q / D + eps);
  float *yb = y + (size_t)bi * D * T + t;
  for (int d = 0; d < D; ++d) yb[(size_t)d * T] = (xb[(size_t)d * T] - mean) * rstd * gamma[d] + beta[d];
}
extern "C" int bdm_layer_norm_channels(int b, int d, int t, const float *x, const float *gamma, const float *beta, float eps,
                                       float *y, void *stream) {
  BDM_REQUIRE(b >= 0 && d >= 1 && t >= 1, "layer_norm_channels: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(layer_norm_channels_kernel, dim3(cdiv(t, 64), b), dim3(64), 0, (hipStream_t)stream, d, t, x, gamma, beta,
                     eps, y);
  return launch_status("layer_norm_channels");
}

// conditioning image (B, H*W, 3 + D) pixel-major = cat[(rgb - cmean)/cstd, bilinear_upsample(tokens[:, :, 1:])]
// (feature_model.py:107-119: F.interpolate(..., mode='bilinear', align_corners=False); projection_model.py:112-115)
__global__ void vit_cond_image_kernel(int D, int g, int H, int W, float cmean, float cstd, const float *__restrict__ tokens,
                                      const float *__restrict__ img, float *__restrict__ out) {
  const int bi = blockIdx.y, pix = blockIdx.x, y = pix / W, x = pix % W;
  const int T1 = g * g + 1;
  const float sy = fmaxf((y + 0.5f) * ((float)g / H) - 0.5f, 0.f), sx = fmaxf((x + 0.5f) * ((float)g / W) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < g - 1 ? 1 : 0), x1 = x0 + (x0 < g - 1 ? 1 : 0);
  const float ly1 = sy - y0, ly0 = 1.f - ly1, lx1 = sx - x0, lx0 = 1.f - lx1;
  float *o = out + ((size_t)bi * H * W + pix) * (3 + D);
  if (threadIdx.x < 3) o[threadIdx.x] = (img[(((size_t)bi * 3 + threadIdx.x) * H + y) * W + x] - cmean) / cstd;
  const float *tb = tokens + (size_t)bi * D * T1 + 1;  // skip the CLS column
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float *tr = tb + (size_t)d * T1;
    const float v00 = tr[y0 * g + x0], v01 = tr[y0 * g + x1], v10 = tr[y1 * g + x0], v11 = tr[y1 * g + x1];
    o[3 + d] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
}
extern "C" int bdm_vit_conditioning_image(int b, int d, int grid, int h, int w, float colors_mean, float colors_std,
                                          const float *tokens, const float *img, float *out, void *stream) {
  BDM_REQUIRE(b >= 0 && d >= 1 && grid >= 1 && h >= 1 && w >= 1, "vit_conditioning_image: bad sizes");
  if (b == 0) return BDM_OK;
  hipLaunchKernelGGL(vit_cond_image_kernel, dim3(h * w, b), dim3(128), 0, (hipStream_t)stream, d, grid, h, w, colors_mean,
                     colors_std, tokens, img, out);
  return launch_status("vit_conditioning_image");
}
