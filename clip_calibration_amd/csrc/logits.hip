// Tail of the hot path (SURVEY a-10..a-12), fp32 end to end so that logits match the fp32 oracle given the same features:
//
//   cosine_logits_kernel   logits[B,C] = scale * img_n @ txt_n^T        (zsclip.py:100-101; coop.py:215-217)
//                          exact-f32 matrix cores (v_mfma_f32_16x16x4_f32), operands straight from global/L2
//   row_calibrate_kernel   pred = argmax_c ; logits[i,:] *= dac_conf[pred]  (distanse_aware_calibration.py:49-58)
//                          conf = max_c softmax(logits[i,:])               (vl_calibrator.py:91; vl_evaluator.py:68,83)
//   ece_accumulate_kernel  per-bin (count, sum conf, sum correct)          (tools/metrics.py:90-130)
//
// HBM-bound: algorithmic bytes = 4*(B*E + C*E) read + 4*B*C written (+ 4*B*C re-read/re-written from L2 when DAC is on).
#include "common.h"

namespace clipmi {
namespace {

// One wave computes a 16(m) x 64(n) tile; a workgroup of 4 waves covers 16 x 256.  Lane l holds, for a 16-wide k step,
// A[m = l&15][k0 + 4*(l>>4) .. +3] and B[n = l&15][same k] as one 16-byte load each; the four elements feed four
// 16x16x4 MFMAs (any bijection of k onto (l>>4, element) is valid as long as A and B use the same one).
__global__ __launch_bounds__(256) void cosine_logits_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                            float scale, float* __restrict__ logits, int B, int C, int E) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int n0 = blockIdx.y * 256 + wave * 64;
  if (n0 >= C) return;  // wave-uniform
  const int r = lane & 15, g = lane >> 4;
  const int m = m0 + r < B ? m0 + r : B - 1;
  const float* ap = img + (int64_t)m * E + g * 4;
  const float* bp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + t * 16 + r < C ? n0 + t * 16 + r : C - 1;
    bp[t] = txt + (int64_t)n * E + g * 4;
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < E; k0 += 16) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ap + k0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bp[t] + k0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[t], 0, 0, 0);
    }
  }
  // D layout: col = lane&15 (n), row = (lane>>4)*4 + reg (m)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + t * 16 + r;
    if (n >= C) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int mm = m0 + g * 4 + e;
      if (mm < B) logits[(int64_t)mm * C + n] = scale * acc[t][e];
    }
  }
}

// One wave per image row.
// probs == nullptr: DAC scales the row in place (predict of distanse_aware_calibration.py); probs != nullptr: the row is
// left alone and softmax(row * f) is written to probs (which may be the same buffer).  No __restrict__ on these two.
__global__ __launch_bounds__(256) void row_calibrate_kernel(float* logits, const float* __restrict__ dac,
                                                            float* __restrict__ conf, int32_t* __restrict__ pred, int B, int C,
                                                            float* probs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float* lr = logits + (int64_t)row * C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = lr[c];
    if (v > best) { best = v; bi = c; }   // first occurrence wins inside a lane (ascending c)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (bi == 0x7fffffff) bi = 0;  // all-NaN row
  const float f = dac ? dac[bi] : 1.0f;
  const float mx = best * f;
  float se = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = lr[c] * f;
    if (dac && !probs) lr[c] = v;
    se += __expf(v - mx);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
  if (probs) {
    float* pr = probs + (int64_t)row * C;
    const float inv = 1.0f / se;
    for (int c = lane; c < C; c += 64) pr[c] = __expf(lr[c] * f - mx) * inv;   // same lane reads then writes element c
  }
  if (lane == 0) {
    if (conf) conf[row] = 1.0f / se;
    if (pred) pred[row] = bi;
  }
}

// bins: double [3][n_bins+1] = count | sum_conf | sum_correct ; bin index = np.digitize(conf, linspace(0,1,n_bins+1)) - 1
__global__ __launch_bounds__(256) void ece_accumulate_kernel(const float* __restrict__ conf, const int32_t* __restrict__ pred,
                                                             const int64_t* __restrict__ labels, int n, double* __restrict__ bins,
                                                             int n_bins) {
  extern __shared__ double sh[];  // 3 * (n_bins + 1)
  const int nb1 = n_bins + 1;
  for (int i = threadIdx.x; i < 3 * nb1; i += 256) sh[i] = 0.0;
  __syncthreads();
  const double step = 1.0 / (double)n_bins;  // np.linspace: edge_i = i * step, last edge = 1.0 exactly
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double x = (double)conf[i];
    int b = (int)floor(x * n_bins);
    b = b < 0 ? 0 : (b > n_bins ? n_bins : b);
    auto edge = [&](int k) { return k >= n_bins ? 1.0 : (double)k * step; };
    while (b < n_bins && edge(b + 1) <= x) ++b;
    while (b > 0 && edge(b) > x) --b;
    if (x >= 1.0) b = n_bins;
    atomicAdd(&sh[b], 1.0);
    atomicAdd(&sh[nb1 + b], x);
    atomicAdd(&sh[2 * nb1 + b], (labels[i] == (int64_t)pred[i]) ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * nb1; i += 256)
    if (sh[i] != 0.0) atomicAdd(&bins[i], sh[i]);
}

}  // namespace

int launch_logits(const float* img_n, const float* txt_n, float scale, const float* dac_conf, float* logits, float* conf,
                  int32_t* pred, int B, int C, int E, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(img_n && txt_n && logits, CLIPMI_ERR_ARG, "logits: null pointer (img_n, txt_n and logits are required)");
  CLIPMI_REQUIRE(B > 0 && C > 0 && E > 0 && E % 16 == 0, CLIPMI_ERR_SHAPE, "logits: B=%d C=%d E=%d unsupported (E %% 16 == 0)", B, C, E);
  CLIPMI_REQUIRE((uintptr_t)img_n % 16 == 0 && (uintptr_t)txt_n % 16 == 0, CLIPMI_ERR_ARG, "logits: features must be 16-byte aligned");
  hipLaunchKernelGGL(cosine_logits_kernel, dim3((B + 15) / 16, (C + 255) / 256), dim3(256), 0, s, img_n, txt_n, scale, logits, B, C, E);
  int rc = check_launch("cosine_logits_kernel");
  if (rc != CLIPMI_OK) return rc;
  if (dac_conf || conf || pred) rc = launch_calibrate_rows(logits, dac_conf, conf, pred, B, C, s);
  return rc;
}

int launch_calibrate_rows(float* logits, const float* dac_conf, float* conf, int32_t* pred, int B, int C, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(logits, CLIPMI_ERR_ARG, "calibrate_rows: null logits");
  CLIPMI_REQUIRE(B > 0 && C > 0, CLIPMI_ERR_SHAPE, "calibrate_rows: B=%d C=%d", B, C);
  hipLaunchKernelGGL(row_calibrate_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, dac_conf, conf, pred, B, C, (float*)nullptr);
  return check_launch("row_calibrate_kernel");
}

int launch_softmax_rows(const float* logits, const float* dac_conf, float* probs, float* conf, int32_t* pred, int B, int C,
                        hipStream_t s) {
  if (B == 0) return CLIPMI_OK;   // an empty batch has no storage to point at
  CLIPMI_REQUIRE(logits && probs, CLIPMI_ERR_ARG, "softmax_rows: null pointer (logits and probs are required)");
  CLIPMI_REQUIRE(B > 0 && C > 0, CLIPMI_ERR_SHAPE, "softmax_rows: B=%d C=%d", B, C);
  hipLaunchKernelGGL(row_calibrate_kernel, dim3((B + 3) / 4), dim3(256), 0, s, const_cast<float*>(logits), dac_conf, conf, pred, B, C, probs);
  return check_launch("row_calibrate_kernel");
}

int launch_ece_accumulate(const float* conf, const int32_t* pred, const int64_t* labels, int n, double* bins, int n_bins,
                          hipStream_t s) {
  if (n == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(conf && pred && labels && bins, CLIPMI_ERR_ARG, "ece: null pointer");
  CLIPMI_REQUIRE(n_bins > 0 && n_bins <= 1024, CLIPMI_ERR_SHAPE, "ece: n_bins=%d unsupported", n_bins);
  int grid = (n + 255) / 256;
  grid = grid > 1024 ? 1024 : grid;
  hipLaunchKernelGGL(ece_accumulate_kernel, dim3(grid), dim3(256), 3 * (n_bins + 1) * sizeof(double), s, conf, pred, labels, n,
                     bins, n_bins);
  return check_launch("ece_accumulate_kernel");
}

}  // namespace clipmi
