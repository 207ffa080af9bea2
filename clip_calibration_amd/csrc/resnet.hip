// ModifiedResNet image tower (reference clip/model.py:10-150) -- SURVEY §8(f) row f-4.  Activations are NHWC fp16, so a
// 1x1 convolution IS a GEMM on the [B*H*W, C] rows (clipmi_gemm_f16 with the folded BatchNorm as bias and a ReLU / fp16
// residual epilogue); the 3x3 convolutions go through an im2col whose K order (ky, kx, c) makes every tap one
// contiguous C-vector copy.  Everything here is HBM-bound glue around those GEMMs.
//
//   im2col3x3_nchw_kernel   stem conv1 (3 -> width/2, stride 2, pad 1) straight from the fp32 NCHW image; K = c*9 + ky*3 + kx
//   im2col3x3_nhwc_kernel   every other 3x3 (stride 1, pad 1): col[(b,y,x), (ky*3+kx)*C + c] = x[b, y+ky-1, x+kx-1, c]
//   avgpool_nhwc_kernel     nn.AvgPool2d(k) (the anti-aliased stride, clip/model.py:23,33-37,119)
//   attnpool_tokens_kernel  AttentionPool2d's token build (clip/model.py:69-71): [mean | HW tokens] + positional embedding
//   attnpool_kernel         its single-query attention (clip/model.py:72-90): one query (the mean token) per image and head
#include "common.h"

namespace clipmi {
namespace {

// 8 consecutive K columns (16 B) per thread
template <typename TI>
__global__ __launch_bounds__(256) void im2col3x3_nchw_kernel(const TI* __restrict__ img, half_t* __restrict__ col, int B, int Cin,
                                                             int H, int W, int Ho, int Wo, int stride, int Kpad, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one 8-column group
  if (i >= total) return;
  const int kv = Kpad >> 3;
  const int k0 = (int)(i % kv) << 3;
  const int64_t row = i / kv;
  const int xo = (int)(row % Wo), yo = (int)((row / Wo) % Ho);
  const int64_t b = row / ((int64_t)Wo * Ho);
  f16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    float val = 0.f;
    if (k < Cin * 9) {
      const int c = k / 9, t = k - c * 9, ky = t / 3, kx = t - ky * 3;
      const int y = yo * stride + ky - 1, x = xo * stride + kx - 1;
      if (y >= 0 && y < H && x >= 0 && x < W) val = (float)img[((b * Cin + c) * H + y) * W + x];
    }
    v[e] = (half_t)val;
  }
  *reinterpret_cast<f16x8*>(col + row * Kpad + k0) = v;
}

// 8 channels (16 B) per thread; C % 8 == 0
__global__ __launch_bounds__(256) void im2col3x3_nhwc_kernel(const half_t* __restrict__ x, half_t* __restrict__ col, int C, int H,
                                                             int W, int Kpad, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int kv = Kpad >> 3;
  const int k = (int)(i % kv) << 3;
  const int64_t row = i / kv;          // (b*H + y)*W + x
  f16x8 v = f16x8{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  if (k < 9 * C) {
    const int tap = k / C, c = k - tap * C, ky = tap / 3, kx = tap - ky * 3;
    const int xo = (int)(row % W), yo = (int)((row / W) % H);
    const int64_t b = row / ((int64_t)W * H);
    const int y = yo + ky - 1, xx = xo + kx - 1;
    if (y >= 0 && y < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f16x8*>(x + ((b * H + y) * W + xx) * C + c);
  }
  *reinterpret_cast<f16x8*>(col + row * Kpad + k) = v;
}

// 8 channels (16 B) per thread when C % 8 == 0
template <int VEC>
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int C, int H, int W,
                                                           int k, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one group of VEC output channels
  if (i >= total) return;
  const int Ho = H / k, Wo = W / k, cv = C / VEC;
  const int c = (int)(i % cv) * VEC;
  const int xo = (int)((i / cv) % Wo), yo = (int)((i / ((int64_t)cv * Wo)) % Ho);
  const int64_t b = i / ((int64_t)cv * Wo * Ho);
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  for (int dy = 0; dy < k; ++dy)
    for (int dx = 0; dx < k; ++dx) {
      const half_t* p = x + ((b * H + yo * k + dy) * W + xo * k + dx) * C + c;
      if constexpr (VEC == 8) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
      } else {
        s[0] += (float)p[0];
      }
    }
  const float inv = 1.0f / (float)(k * k);
  half_t* q = y + ((b * Ho + yo) * Wo + xo) * C + c;
  if constexpr (VEC == 8) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(s[e] * inv);
    *reinterpret_cast<f16x8*>(q) = o;
  } else {
    q[0] = (half_t)(s[0] * inv);
  }
}

// tokens[b, 0, :] = mean_hw x[b, hw, :] + pos[0, :]; tokens[b, 1 + hw, :] = x[b, hw, :] + pos[1 + hw, :]
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const half_t* __restrict__ x, const float* __restrict__ pos,
                                                              half_t* __restrict__ tok, int HW, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int t = 0; t < HW; ++t) {
      const float v = (float)x[((int64_t)b * HW + t) * C + c];
      s += v;
      tok[((int64_t)b * (HW + 1) + 1 + t) * C + c] = (half_t)(v + pos[(int64_t)(1 + t) * C + c]);
    }
    tok[(int64_t)b * (HW + 1) * C + c] = (half_t)(s / (float)HW + pos[c]);
  }
}

// One wave per (image, head): q [B, C] (the projected mean token), kv [B*T, 2C] (k | v), head_dim 64 -> out [B, C].
__global__ __launch_bounds__(256) void attnpool_kernel(const half_t* __restrict__ q, const half_t* __restrict__ kv, half_t* __restrict__ out,
                                                       int B, int T, int heads) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= B * heads) return;
  const int b = item / heads, h = item - b * heads;
  const int C = heads * 64;
  const float qd = (float)q[(int64_t)b * C + h * 64 + lane] * 0.125f;     // 1/sqrt(64)
  float m = -1.0e30f, l = 0.f, o = 0.f;
  for (int t = 0; t < T; ++t) {
    const half_t* row = kv + ((int64_t)b * T + t) * 2 * C + h * 64;
    float s = qd * (float)row[lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mn = fmaxf(m, s);
    const float a = __expf(m - mn), p = __expf(s - mn);
    l = l * a + p;
    o = o * a + p * (float)row[C + lane];
    m = mn;
  }
  out[(int64_t)b * C + h * 64 + lane] = (half_t)(o / l);
}

}  // namespace

int launch_im2col3x3_nchw(const void* image, int dtype, half_t* col, int B, int Cin, int H, int W, int stride, int Kpad, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(image && col, CLIPMI_ERR_ARG, "im2col3x3_nchw: null pointer");
  CLIPMI_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && Kpad >= Cin * 9 && Kpad % 64 == 0, CLIPMI_ERR_SHAPE,
                 "im2col3x3_nchw: B=%d Cin=%d H=%d W=%d stride=%d Kpad=%d", B, Cin, H, W, stride, Kpad);
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  const int64_t total = (int64_t)B * Ho * Wo * (Kpad / 8);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == CLIPMI_F32)
    hipLaunchKernelGGL(im2col3x3_nchw_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)image, col, B, Cin, H, W, Ho, Wo, stride, Kpad, total);
  else if (dtype == CLIPMI_F16)
    hipLaunchKernelGGL(im2col3x3_nchw_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)image, col, B, Cin, H, W, Ho, Wo, stride, Kpad, total);
  else {
    set_error("im2col3x3_nchw: bad dtype %d", dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("im2col3x3_nchw_kernel");
}

int launch_im2col3x3_nhwc(const half_t* x, half_t* col, int B, int H, int W, int C, int Kpad, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x && col, CLIPMI_ERR_ARG, "im2col3x3_nhwc: null pointer");
  CLIPMI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && Kpad >= 9 * C && Kpad % 64 == 0, CLIPMI_ERR_SHAPE,
                 "im2col3x3_nhwc: B=%d H=%d W=%d C=%d Kpad=%d (C %% 8 == 0, Kpad %% 64 == 0)", B, H, W, C, Kpad);
  CLIPMI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)col % 16 == 0, CLIPMI_ERR_ARG, "im2col3x3_nhwc: unaligned pointer");
  const int64_t total = (int64_t)B * H * W * (Kpad / 8);
  hipLaunchKernelGGL(im2col3x3_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, col, C, H, W, Kpad, total);
  return check_launch("im2col3x3_nhwc_kernel");
}

int launch_avgpool_nhwc(const half_t* x, half_t* y, int B, int H, int W, int C, int k, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x && y, CLIPMI_ERR_ARG, "avgpool: null pointer");
  CLIPMI_REQUIRE(B > 0 && C > 0 && k >= 1 && H % k == 0 && W % k == 0, CLIPMI_ERR_SHAPE, "avgpool: B=%d H=%d W=%d C=%d k=%d", B, H, W, C, k);
  if (C % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0) {
    const int64_t total = (int64_t)B * (H / k) * (W / k) * (C / 8);
    hipLaunchKernelGGL(avgpool_nhwc_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, C, H, W, k, total);
  } else {
    const int64_t total = (int64_t)B * (H / k) * (W / k) * C;
    hipLaunchKernelGGL(avgpool_nhwc_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, C, H, W, k, total);
  }
  return check_launch("avgpool_nhwc_kernel");
}

int launch_attnpool_tokens(const half_t* x, const float* pos, half_t* tokens, int B, int HW, int C, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x && pos && tokens && B > 0 && HW > 0 && C > 0, CLIPMI_ERR_ARG, "attnpool_tokens: bad argument");
  hipLaunchKernelGGL(attnpool_tokens_kernel, dim3(B), dim3(256), 0, s, x, pos, tokens, HW, C);
  return check_launch("attnpool_tokens_kernel");
}

int launch_attnpool(const half_t* q, const half_t* kv, half_t* out, int B, int T, int heads, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(q && kv && out && B > 0 && T > 0 && heads > 0, CLIPMI_ERR_ARG, "attnpool: bad argument");
  const int64_t items = (int64_t)B * heads;
  hipLaunchKernelGGL(attnpool_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, q, kv, out, B, T, heads);
  return check_launch("attnpool_kernel");
}

}  // namespace clipmi
