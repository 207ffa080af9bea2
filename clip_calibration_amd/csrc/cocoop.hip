// CoCoOp's instance-conditioned prompts (reference trainers/classification/cocoop.py:154-199) -- SURVEY §8(f) row f-4.
//
//   cocoop_ctx_kernel        bias[b] = W2 relu(W1 f[b] + b1) + b2 ; ctx_shifted[b] = ctx + bias[b]   (cocoop.py:154-161)
//   cocoop_prompts_kernel    prompts[(b,c), l, :] = 1 <= l <= n_ctx ? ctx_shifted[b, l-1, :] : base[c, l, :]   (:163-171)
//   per_image_logits_kernel  logits[b,c] = scale * <img_n[b], txt[b,c] / ||txt[b,c]||>                  (:193-199)
//
// The text tower between the last two is the ordinary clipmi_text_encoder on B*C prompts -- that is where the time goes
// (5.96 GFLOP per prompt); these three are HBM-bound glue: algorithmic bytes = 2*L*D per prompt written (+ read by the
// tower) and 4*E per (image, class) pair read.
#include "common.h"

namespace clipmi {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One workgroup (4 waves) per image.  hidden = E/16 is 32 for ViT-B/16: each wave owns hidden units wave, wave+4, ...
__global__ __launch_bounds__(256) void cocoop_ctx_kernel(const float* __restrict__ img, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, const float* __restrict__ ctx,
                                                         float* __restrict__ out, int E, int H, int D, int n_ctx) {
  extern __shared__ float hid[];   // [H]
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* f = img + (int64_t)b * E;
  for (int h = wave; h < H; h += 4) {
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += w1[(int64_t)h * E + e] * f[e];
    s = wave_sum(s);
    if (lane == 0) hid[h] = fmaxf(s + b1[h], 0.f);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = b2[d];
    for (int h = 0; h < H; ++h) s += w2[(int64_t)d * H + h] * hid[h];
    for (int t = 0; t < n_ctx; ++t) out[((int64_t)b * n_ctx + t) * D + d] = ctx[(int64_t)t * D + d] + s;
  }
}

// 8 halves (16 B) per thread.
template <typename TB>
__global__ __launch_bounds__(256) void cocoop_prompts_kernel(const TB* __restrict__ base, const float* __restrict__ ctx,
                                                             half_t* __restrict__ out, int C, int L, int D, int n_ctx,
                                                             int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int dv = D >> 3;
  const int d = (int)(i % dv) << 3;
  const int64_t row = i / dv;            // (b*C + c)*L + l
  const int l = (int)(row % L);
  const int64_t bc = row / L;
  const int c = (int)(bc % C);
  const int64_t b = bc / C;
  f16x8 v;
  if (l >= 1 && l <= n_ctx) {
    const float* s = ctx + ((int64_t)b * n_ctx + (l - 1)) * D + d;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)s[e];
  } else {
    const TB* s = base + ((int64_t)c * L + l) * D + d;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)s[e];
  }
  *reinterpret_cast<f16x8*>(out + row * D + d) = v;
}

// One wave per (image, class) pair.
__global__ __launch_bounds__(256) void per_image_logits_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                               float scale, float* __restrict__ logits,
                                                               float* __restrict__ txt_n_last, int B, int C, int E) {
  const int lane = threadIdx.x & 63;
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (int64_t)B * C) return;
  const int b = (int)(pair / C);
  const float* t = txt + pair * E;
  const float* f = img + (int64_t)b * E;
  float dot = 0.f, ss = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float tv = t[e];
    dot += tv * f[e];
    ss += tv * tv;
  }
  dot = wave_sum(dot);
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss);
  if (lane == 0) logits[pair] = scale * dot * inv;
  if (txt_n_last && b == B - 1) {   // the reference returns the LAST image's text features (cocoop.py:199)
    const int c = (int)(pair % C);
    for (int e = lane; e < E; e += 64) txt_n_last[(int64_t)c * E + e] = t[e] * inv;
  }
}

// CLIP-Adapter (reference trainers/classification/clip_adapter.py:138-172): out[b] = ratio * relu(W2 relu(W1 f[b])) + (1 - ratio) * f[b],
// both Linears without bias.  One workgroup per image; same shape of work as the CoCoOp meta-net.
__global__ __launch_bounds__(256) void adapter_blend_kernel(const float* __restrict__ f, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, float ratio, float* __restrict__ out,
                                                            int E, int H) {
  extern __shared__ float hid[];   // [H]
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* x = f + (int64_t)b * E;
  for (int h = wave; h < H; h += 4) {
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s += w1[(int64_t)h * E + e] * x[e];
    s = wave_sum(s);
    if (lane == 0) hid[h] = fmaxf(s, 0.f);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += 256) {
    float s = 0.f;
    for (int h = 0; h < H; ++h) s += w2[(int64_t)e * H + h] * hid[h];
    out[(int64_t)b * E + e] = ratio * fmaxf(s, 0.f) + (1.f - ratio) * x[e];
  }
}

// out = a + alpha * b  (TaskRes: base text features + alpha * learned residual, taskres.py:105-106)
__global__ __launch_bounds__(256) void scale_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float alpha,
                                                        float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = a[i] + alpha * b[i];
}

}  // namespace

int launch_adapter_blend(const float* f, const float* w1, const float* w2, float ratio, float* out, int B, int E, int H, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(f && w1 && w2 && out, CLIPMI_ERR_ARG, "adapter_blend: null pointer");
  CLIPMI_REQUIRE(B > 0 && E > 0 && H > 0 && H <= 8192, CLIPMI_ERR_SHAPE, "adapter_blend: B=%d E=%d H=%d", B, E, H);
  hipLaunchKernelGGL(adapter_blend_kernel, dim3(B), dim3(256), H * sizeof(float), s, f, w1, w2, ratio, out, E, H);
  return check_launch("adapter_blend_kernel");
}

int launch_scale_add(const float* a, const float* b, float alpha, float* out, int64_t n, hipStream_t s) {
  if (n == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(a && b && out && n > 0, CLIPMI_ERR_ARG, "scale_add: null pointer");
  hipLaunchKernelGGL(scale_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, alpha, out, n);
  return check_launch("scale_add_kernel");
}

int launch_cocoop_ctx(const float* img_n, const float* w1, const float* b1, const float* w2, const float* b2, const float* ctx,
                      float* ctx_shifted, int B, int E, int H, int D, int n_ctx, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(img_n && w1 && b1 && w2 && b2 && ctx && ctx_shifted, CLIPMI_ERR_ARG, "cocoop_ctx: null pointer");
  CLIPMI_REQUIRE(B > 0 && E > 0 && H > 0 && H <= 4096 && D > 0 && n_ctx > 0, CLIPMI_ERR_SHAPE,
                 "cocoop_ctx: B=%d E=%d H=%d D=%d n_ctx=%d unsupported", B, E, H, D, n_ctx);
  hipLaunchKernelGGL(cocoop_ctx_kernel, dim3(B), dim3(256), H * sizeof(float), s, img_n, w1, b1, w2, b2, ctx, ctx_shifted, E, H, D, n_ctx);
  return check_launch("cocoop_ctx_kernel");
}

int launch_cocoop_prompts(const void* base, int base_dtype, const float* ctx_shifted, half_t* prompts, int nb, int C, int L, int D,
                          int n_ctx, hipStream_t s) {
  if (nb == 0 || C == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(base && ctx_shifted && prompts, CLIPMI_ERR_ARG, "cocoop_prompts: null pointer");
  CLIPMI_REQUIRE(nb > 0 && C > 0 && L > 1 && D > 0 && D % 8 == 0 && n_ctx > 0 && n_ctx < L - 1, CLIPMI_ERR_SHAPE,
                 "cocoop_prompts: nb=%d C=%d L=%d D=%d n_ctx=%d unsupported (D %% 8 == 0, 0 < n_ctx < L-1)", nb, C, L, D, n_ctx);
  CLIPMI_REQUIRE((uintptr_t)prompts % 16 == 0, CLIPMI_ERR_ARG, "cocoop_prompts: prompts must be 16-byte aligned");
  const int64_t total = (int64_t)nb * C * L * (D / 8);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (base_dtype == CLIPMI_F32)
    hipLaunchKernelGGL(cocoop_prompts_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)base, ctx_shifted, prompts, C, L, D, n_ctx, total);
  else if (base_dtype == CLIPMI_F16)
    hipLaunchKernelGGL(cocoop_prompts_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)base, ctx_shifted, prompts, C, L, D, n_ctx, total);
  else {
    set_error("cocoop_prompts: bad base dtype %d", base_dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("cocoop_prompts_kernel");
}

int launch_logits_per_image(const float* img_n, const float* txt, float scale, const float* dac_conf, float* logits, float* conf,
                            int32_t* pred, float* txt_n_last, int B, int C, int E, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(img_n && txt && logits, CLIPMI_ERR_ARG, "logits_per_image: null pointer (img_n, txt and logits are required)");
  CLIPMI_REQUIRE(B > 0 && C > 0 && E > 0, CLIPMI_ERR_SHAPE, "logits_per_image: B=%d C=%d E=%d", B, C, E);
  const int64_t pairs = (int64_t)B * C;
  hipLaunchKernelGGL(per_image_logits_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 0, s, img_n, txt, scale, logits, txt_n_last, B, C, E);
  int rc = check_launch("per_image_logits_kernel");
  if (rc != CLIPMI_OK) return rc;
  if (dac_conf || conf || pred) rc = launch_calibrate_rows(logits, dac_conf, conf, pred, B, C, s);
  return rc;
}

}  // namespace clipmi
