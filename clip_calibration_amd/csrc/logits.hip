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

// The row pass, one wave per image row (lane = threadIdx & 63):
// probs == nullptr: DAC scales the row in place (predict of distanse_aware_calibration.py); probs != nullptr: the row is
// left alone and softmax(row * f) is written to probs (which may be the same buffer).  No __restrict__ on these two.
__device__ __forceinline__ void calibrate_row(float* lr, const float* __restrict__ dac, int C, float* pr, int lane, float& conf_out, int& pred_out) {
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
    if (dac && !pr) lr[c] = v;
    se += __expf(v - mx);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
  if (pr) {
    const float inv = 1.0f / se;
    for (int c = lane; c < C; c += 64) pr[c] = __expf(lr[c] * f - mx) * inv;   // same lane reads then writes element c
  }
  conf_out = 1.0f / se;
  pred_out = bi;
}

__global__ __launch_bounds__(256) void row_calibrate_kernel(float* logits, const float* __restrict__ dac,
                                                            float* __restrict__ conf, int32_t* __restrict__ pred, int B, int C,
                                                            float* probs) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float cf;
  int pi;
  calibrate_row(logits + (int64_t)row * C, dac, C, probs ? probs + (int64_t)row * C : nullptr, lane, cf, pi);
  if (lane == 0) {
    if (conf) conf[row] = cf;
    if (pred) pred[row] = pi;
  }
}

// bin index = np.digitize(conf, linspace(0, 1, n_bins + 1)) - 1, with conf == 1.0 in bin n_bins (tools/metrics.py:90-130)
__device__ __forceinline__ int ece_bin(double x, int n_bins) {
  const double step = 1.0 / (double)n_bins;  // np.linspace: edge_i = i * step, last edge = 1.0 exactly
  int b = (int)floor(x * n_bins);
  b = b < 0 ? 0 : (b > n_bins ? n_bins : b);
  auto edge = [&](int k) { return k >= n_bins ? 1.0 : (double)k * step; };
  while (b < n_bins && edge(b + 1) <= x) ++b;
  while (b > 0 && edge(b) > x) --b;
  if (x >= 1.0) b = n_bins;
  return b;
}

// bins: double [3][n_bins+1] = count | sum_conf | sum_correct ; bin index = np.digitize(conf, linspace(0,1,n_bins+1)) - 1
__global__ __launch_bounds__(256) void ece_accumulate_kernel(const float* __restrict__ conf, const int32_t* __restrict__ pred,
                                                             const int64_t* __restrict__ labels, int n, double* __restrict__ bins,
                                                             int n_bins) {
  extern __shared__ double sh[];  // 3 * (n_bins + 1)
  const int nb1 = n_bins + 1;
  for (int i = threadIdx.x; i < 3 * nb1; i += 256) sh[i] = 0.0;
  __syncthreads();
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double x = (double)conf[i];
    const int b = ece_bin(x, n_bins);
    atomicAdd(&sh[b], 1.0);
    atomicAdd(&sh[nb1 + b], x);
    atomicAdd(&sh[2 * nb1 + b], (labels[i] == (int64_t)pred[i]) ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * nb1; i += 256)
    if (sh[i] != 0.0) atomicAdd(&bins[i], sh[i]);
}


// ---------------------------------------------------------------------------------------------------------------
// The fused tail (north star: "fused normalise + matmul + DAC-temperature kernel"): ONE launch for
//   img_n = img / ||img||                      (zsclip.py:99)
//   logits = scale * img_n @ txt_n^T           (zsclip.py:100-101)
//   pred = argmax; logits[i,:] *= dac[pred]    (distanse_aware_calibration.py:49-58)
//   conf = max softmax(logits[i,:])            (vl_calibrator.py:91, vl_evaluator.py:68,83)
//   ECE bin accumulation                       (tools/metrics.py:90-130)
// Grid: (B/16) x (C/64) workgroups of 4 waves -- 256 workgroups at B = 256, C = 1000.  A workgroup normalises its 16
// image rows into LDS (the arithmetic of l2norm_kernel, element for element: the row norm is a lane-strided sum and a
// butterfly), each wave then runs a 16 x 16 tile of exact-f32 MFMAs (v_mfma_f32_16x16x4_f32, the k order of
// cosine_logits_kernel) with its A fragments from LDS and its 16 text rows straight from L2, and stores its logits.
// The workgroups of one 16-row block then draw a ticket; the LAST of them to arrive (agent-scope release before the
// ticket, acquire after it: cdna_hip_programming.md Guideline 16, counter form) runs the row pass -- argmax, DAC
// factor, softmax top-1, ECE bins -- over the block's 16 complete rows, one wave per row, with the arithmetic of
// row_calibrate_kernel, and resets the ticket counter.  Results are bit-identical to the three-kernel path.
// counters: ceil(B/16) int32, zero before the first launch, left zero by every launch.
// ---------------------------------------------------------------------------------------------------------------
template <bool NORMALIZE, typename TI>
__global__ __launch_bounds__(256) void fused_tail_kernel(const TI* __restrict__ img, const float* __restrict__ txt, float scale,
                                                         const float* __restrict__ dac, float* logits, float* __restrict__ img_n_out,
                                                         float* __restrict__ conf, int32_t* __restrict__ pred,
                                                         const int64_t* __restrict__ labels, double* __restrict__ bins, int n_bins,
                                                         int* counters, int B, int C, int E) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 16 rows of E floats, row stride E*4 + 16 bytes; then one int
  const int rs = E * 4 + 16;
  int* flag = reinterpret_cast<int*>(smem + 16 * rs);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * 16;
  const int n0 = blockIdx.y * 64 + wave * 16;

  // ---- phase 0: the 16 image rows -> LDS (normalised), 4 rows per wave
  for (int rr = 0; rr < 4; ++rr) {
    const int rl = wave * 4 + rr;
    const int row = m0 + rl < B ? m0 + rl : B - 1;
    const TI* x = img + (int64_t)row * E;
    float inv = 1.0f;
    if constexpr (NORMALIZE) {
      float ss = 0.f;
      for (int e = lane; e < E; e += 64) {
        const float v = (float)x[e];
        ss += v * v;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
      inv = 1.0f / sqrtf(ss);
    }
    float* dst = reinterpret_cast<float*>(smem + rl * rs);
    const bool keep = img_n_out != nullptr && blockIdx.y == 0 && m0 + rl < B;   // wave-uniform
    for (int e = lane; e < E; e += 64) {
      const float v = NORMALIZE ? (float)x[e] * inv : (float)x[e];
      dst[e] = v;
      if (keep) img_n_out[(int64_t)row * E + e] = v;
    }
  }
  __syncthreads();

  // ---- phase 1: 16 x 16 tile per wave
  const int r = lane & 15, g = lane >> 4;
  if (n0 < C) {   // wave-uniform
    const int n = n0 + r < C ? n0 + r : C - 1;
    const float* bp = txt + (int64_t)n * E + g * 4;
    const char* ap = smem + r * rs + g * 16;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < E; k0 += 64) {   // E % 64 == 0: four 16-wide steps with their loads issued together
      f32x4 b[4], a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const f32x4*>(bp + k0 + u * 16);
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(ap + (k0 + u * 16) * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
    // D layout: col = lane&15 (n), row = (lane>>4)*4 + reg (m)
    if (n0 + r < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int mm = m0 + g * 4 + e;
        if (mm < B) logits[(int64_t)mm * C + n0 + r] = scale * acc[e];
      }
    }
  }
  if (!dac && !conf && !pred && !bins) return;   // logits only (kernel argument: uniform)

  // ---- phase 2: ticket; the last workgroup of this row block owns the row pass
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep: the compiler may drop the fence's own wait (Guideline 16, pitfall 12)
    const int ticket = __hip_atomic_fetch_add(counters + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == (int)gridDim.y - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(counters + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  for (int rr = 0; rr < 4; ++rr) {
    const int row = m0 + wave * 4 + rr;
    if (row >= B) break;   // wave-uniform
    float cf;
    int pi;
    calibrate_row(logits + (int64_t)row * C, dac, C, nullptr, lane, cf, pi);
    if (lane == 0) {
      if (conf) conf[row] = cf;
      if (pred) pred[row] = pi;
      if (bins) {
        const double x = (double)cf;
        const int b = ece_bin(x, n_bins), nb1 = n_bins + 1;
        atomicAdd(&bins[b], 1.0);
        atomicAdd(&bins[nb1 + b], x);
        atomicAdd(&bins[2 * nb1 + b], (labels[row] == (int64_t)pi) ? 1.0 : 0.0);
      }
    }
  }
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

size_t fused_tail_workspace_bytes(int B) { return align256((size_t)((B + 15) / 16) * sizeof(int)); }

int launch_fused_tail(const void* img_, int img_dtype, int normalize, const float* txt_n, float scale, const float* dac_conf, float* logits,
                      float* img_n_out, float* conf, int32_t* pred, const int64_t* labels, double* bins, int n_bins, void* workspace,
                      size_t workspace_bytes, int B, int C, int E, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(img_ && txt_n && logits, CLIPMI_ERR_ARG, "fused_tail: null pointer (img, txt_n and logits are required)");
  CLIPMI_REQUIRE(img_dtype == CLIPMI_F32 || img_dtype == CLIPMI_F16, CLIPMI_ERR_ARG, "fused_tail: bad image-feature dtype %d", img_dtype);
  const float* img = static_cast<const float*>(img_);
  CLIPMI_REQUIRE(B > 0 && C > 0 && E > 0 && E % 16 == 0, CLIPMI_ERR_SHAPE, "fused_tail: B=%d C=%d E=%d unsupported (E %% 16 == 0)", B, C, E);
  CLIPMI_REQUIRE((uintptr_t)img % 16 == 0 && (uintptr_t)txt_n % 16 == 0, CLIPMI_ERR_ARG, "fused_tail: features must be 16-byte aligned");
  CLIPMI_REQUIRE(!bins || (labels && n_bins > 0 && n_bins <= 1024), CLIPMI_ERR_ARG, "fused_tail: ECE bins need labels and 1 <= n_bins <= 1024");
  CLIPMI_REQUIRE(normalize || !img_n_out, CLIPMI_ERR_ARG, "fused_tail: img_n_out only with normalize");
  const int lds = 16 * (E * 4 + 16) + 16;
  const bool fits = E % 64 == 0 && lds <= 160 * 1024;
  if (options().tail_unfused.load(std::memory_order_relaxed) == 1 || !fits) {
    // the same arithmetic as separate launches (A/B aid; also shapes the fused kernel does not take)
    int rc;
    const float* in = img;
    if (normalize) {
      CLIPMI_REQUIRE(img_n_out, CLIPMI_ERR_ARG, "fused_tail: the unfused path needs img_n_out to hold the normalised features");
      if ((rc = launch_l2_normalize(img_, img_dtype, img_n_out, B, E, s))) return rc;
      in = img_n_out;
    } else {
      CLIPMI_REQUIRE(img_dtype == CLIPMI_F32, CLIPMI_ERR_ARG, "fused_tail: the unfused path takes fp32 normalised features");
    }
    if ((rc = launch_logits(in, txt_n, scale, dac_conf, logits, conf, pred, B, C, E, s))) return rc;
    if (bins) {
      CLIPMI_REQUIRE(conf && pred, CLIPMI_ERR_ARG, "fused_tail: the unfused path needs conf and pred buffers for the ECE bins");
      return launch_ece_accumulate(conf, pred, labels, B, bins, n_bins, s);
    }
    return CLIPMI_OK;
  }
  CLIPMI_REQUIRE(workspace && workspace_bytes >= fused_tail_workspace_bytes(B), CLIPMI_ERR_WORKSPACE,
                 "fused_tail: workspace too small (%zu < %zu)", workspace_bytes, fused_tail_workspace_bytes(B));
  const dim3 grid((B + 15) / 16, (C + 63) / 64);
  CLIPMI_REQUIRE(grid.y <= 65535, CLIPMI_ERR_SHAPE, "fused_tail: too many classes");
  int* counters = static_cast<int*>(workspace);
  auto go = [&](auto kernel, DeviceOnce& once, auto* typed) {
    ensure_dynamic_lds(kernel, lds, once);
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, typed, txt_n, scale, dac_conf, logits, img_n_out, conf, pred, labels, bins, n_bins,
                       counters, B, C, E);
  };
  static DeviceOnce once[4];
  const half_t* img16 = static_cast<const half_t*>(img_);
  if (img_dtype == CLIPMI_F32) {
    if (normalize) go(fused_tail_kernel<true, float>, once[0], img);
    else go(fused_tail_kernel<false, float>, once[1], img);
  } else {
    if (normalize) go(fused_tail_kernel<true, half_t>, once[2], img16);
    else go(fused_tail_kernel<false, half_t>, once[3], img16);
  }
  return check_launch("fused_tail_kernel");
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
