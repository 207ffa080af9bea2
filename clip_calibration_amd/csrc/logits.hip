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
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 16 rows of E floats, row stride E*4 + 16 bytes; then flag + wave partials
  const int rs = E * 4 + 16;
  int* flag = reinterpret_cast<int*>(smem + 16 * rs);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * 16;
  const int n0 = blockIdx.y * 64 + wave * 16;

  // ---- phase 0: the 16 image rows -> LDS (normalised), 4 rows per wave, all four rows' loads in flight together (row after
  //      row this phase alone was eight dependent round trips to L2).  Arithmetic of l2norm_kernel: lane-strided sum, butterfly.
  if (E <= 1024) {   // uniform: a lane's <= 16 elements of each row stay in registers between the norm and the scaling
    float xv[4][16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int rl = wave * 4 + rr;
        const int row = m0 + rl < B ? m0 + rl : B - 1;
        xv[rr][t] = lane + 64 * t < E ? (float)img[(int64_t)row * E + lane + 64 * t] : 0.f;
      }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int rl = wave * 4 + rr;
      const int row = m0 + rl < B ? m0 + rl : B - 1;
      float inv = 1.0f;
      if constexpr (NORMALIZE) {
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (lane + 64 * t < E) ss += xv[rr][t] * xv[rr][t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        inv = 1.0f / sqrtf(ss);
      }
      float* dst = reinterpret_cast<float*>(smem + rl * rs);
      const bool keep = img_n_out != nullptr && blockIdx.y == 0 && m0 + rl < B;   // wave-uniform
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (lane + 64 * t < E) {
          const float v = NORMALIZE ? xv[rr][t] * inv : xv[rr][t];
          dst[lane + 64 * t] = v;
          if (keep) img_n_out[(int64_t)row * E + lane + 64 * t] = v;
        }
      }
    }
  } else {
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
  }
  __syncthreads();

  // ---- phase 1: 16 x 16 tile per wave
  const int r = lane & 15, g = lane >> 4;
  f32x4 acc_out = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n0 < C) {   // wave-uniform
    const int n = n0 + r < C ? n0 + r : C - 1;
    const float* bp = txt + (int64_t)n * E + g * 4;
    const char* ap = smem + r * rs + g * 16;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // E % 64 == 0: 64-wide steps, the text fragments of step k+1 are loaded (from L2) while the 16 MFMAs of step k run --
    // without the explicit second register set the loop is one L2 round trip per step
    f32x4 bn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) bn[u] = *reinterpret_cast<const f32x4*>(bp + u * 16);
    for (int k0 = 0; k0 < E; k0 += 64) {
      f32x4 b[4], a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = bn[u];
      if (k0 + 64 < E) {
#pragma unroll
        for (int u = 0; u < 4; ++u) bn[u] = *reinterpret_cast<const f32x4*>(bp + k0 + 64 + u * 16);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(ap + (k0 + u * 16) * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc, 0, 0, 0);
    }
    acc_out = acc;
  }
  // ---- the workgroup's 16 x 64 logits tile.  Rows 16-byte aligned (C % 4 == 0): through an LDS tile, ONE 16-byte
  //      write-through (sc1) store per thread -- whole 64-byte row pieces, and no release fence is needed for the hand-off
  //      below (cdna_hip_programming.md Guideline 16, R1).  Otherwise: 4-byte stores from the accumulator layout + a release.
  const bool wide = (C & 3) == 0;   // uniform
  if (wide) {
    __syncthreads();                // every wave is done reading the image rows: the tile takes their place
    float* tile = reinterpret_cast<float*>(smem);   // [16][64 + 4] floats
    // D layout: col = lane&15 (n), row = (lane>>4)*4 + reg (m)
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[(g * 4 + e) * 68 + wave * 16 + r] = scale * acc_out[e];
    __syncthreads();
    const int trow = threadIdx.x >> 4, tq = threadIdx.x & 15;
    const int mm = m0 + trow, nn = blockIdx.y * 64 + tq * 4;
    if (mm < B && nn < C) {         // C % 4 == 0: a 4-column piece is entirely inside or outside
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + trow * 68 + tq * 4);
      const __amdgpu_buffer_rsrc_t lrs = make_rsrc(logits, (int64_t)B * C * 4);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), lrs, (int)(((int64_t)mm * C + nn) * 4), 0, 16 /* sc1 */);
    }
  } else if (n0 < C && n0 + r < C) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int mm = m0 + g * 4 + e;
      if (mm < B) logits[(int64_t)mm * C + n0 + r] = scale * acc_out[e];
    }
  }
  if (!dac && !conf && !pred && !bins) return;   // logits only (kernel argument: uniform)

  // ---- phase 2: ticket; the last workgroup of this row block owns the row pass over the logits the others stored
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!wide) {   // plain stores: write this XCD's L2 back before the ticket (write-through stores need no release)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep: the compiler may drop the fence's own wait (Guideline 16, pitfall 12)
    }
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
  // ECE bins of this row block: summed in LDS first (the image-row region is free by now), then at most 3 * (n_bins + 1)
  // global atomics per row block -- one set of three per ROW on a dozen hot addresses serialised the whole grid
  // (B = 2048: 41 us of the launch).
  double* sbins = reinterpret_cast<double*>(smem + 8192);
  if (bins) {
    for (int t = threadIdx.x; t < 3 * (n_bins + 1); t += 256) sbins[t] = 0.0;
    __syncthreads();
  }
  // the wave's four rows side by side: calibrate_row's arithmetic per row (same lane-strided order, same butterflies), with
  // every load of the pass in flight at once.  C <= 1024: a lane's 16 elements of each row stay in registers between the
  // argmax and the softmax pass (one trip to L2 / HBM instead of two chains of dependent round trips -- done one row after
  // the other with rolled loops this pass made the fused launch slower than the three launches it replaces: 43 us against
  // 25 + 9 + 5 in rocprof).  Larger C: the same arithmetic with the two passes reading memory.
  const int row0 = m0 + wave * 4;
  float* lr[4];
  float best[4], fac[4], mx[4], se[4];
  int bi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = row0 + q < B ? row0 + q : B - 1;   // clamped rows recompute row B-1 and are not written
    lr[q] = logits + (int64_t)row * C;
    best[q] = -INFINITY;
    bi[q] = 0x7fffffff;
    se[q] = 0.f;
  }
  auto finish_argmax = [&](int q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best[q], o, 64);
      const int oi = __shfl_xor(bi[q], o, 64);
      if (ob > best[q] || (ob == best[q] && oi < bi[q])) { best[q] = ob; bi[q] = oi; }
    }
    if (bi[q] == 0x7fffffff) bi[q] = 0;
    fac[q] = dac ? dac[bi[q]] : 1.0f;
    mx[q] = best[q] * fac[q];
  };
  if (C <= 1024) {   // uniform
    float v[4][16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q][t] = lane + 64 * t < C ? lr[q][lane + 64 * t] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int t = 0; t < 16; ++t)
        if (lane + 64 * t < C && v[q][t] > best[q]) { best[q] = v[q][t]; bi[q] = lane + 64 * t; }
      finish_argmax(q);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (lane + 64 * t < C) {
          const float x = v[q][t] * fac[q];
          if (dac && row0 + q < B) lr[q][lane + 64 * t] = x;
          se[q] += __expf(x - mx[q]);
        }
      }
    }
  } else {
    for (int c = lane; c < C; c += 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = lr[q][c];
        if (x > best[q]) { best[q] = x; bi[q] = c; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) finish_argmax(q);
    for (int c = lane; c < C; c += 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x = lr[q][c] * fac[q];
        if (dac && row0 + q < B) lr[q][c] = x;
        se[q] += __expf(x - mx[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se[q] += __shfl_xor(se[q], o, 64);
    const int row = row0 + q;
    if (lane == 0 && row < B) {
      const float cf = 1.0f / se[q];
      if (conf) conf[row] = cf;
      if (pred) pred[row] = bi[q];
      if (bins) {
        const double x = (double)cf;
        const int b = ece_bin(x, n_bins), nb1 = n_bins + 1;
        atomicAdd(&sbins[b], 1.0);
        atomicAdd(&sbins[nb1 + b], x);
        atomicAdd(&sbins[2 * nb1 + b], (labels[row] == (int64_t)bi[q]) ? 1.0 : 0.0);
      }
    }
  }
  if (bins) {
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * (n_bins + 1); t += 256)
      if (sbins[t] != 0.0) atomicAdd(&bins[t], sbins[t]);
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

// [ceil(B/16)] int32 ticket counters (zero between launches)
size_t fused_tail_workspace_bytes(int B, int /*C*/) { return align256((size_t)((B + 15) / 16) * sizeof(int)); }

int launch_fused_tail(const void* img_, int img_dtype, int normalize, const float* txt_n, float scale, const float* dac_conf, float* logits,
                      float* img_n_out, float* conf, int32_t* pred, const int64_t* labels, double* bins, int n_bins, void* workspace,
                      size_t workspace_bytes, int B, int C, int E, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(img_ && txt_n && logits, CLIPMI_ERR_ARG, "fused_tail: null pointer (img, txt_n and logits are required)");
  CLIPMI_REQUIRE(img_dtype == CLIPMI_F32 || img_dtype == CLIPMI_F16, CLIPMI_ERR_ARG, "fused_tail: bad image-feature dtype %d", img_dtype);
  const float* img = static_cast<const float*>(img_);
  CLIPMI_REQUIRE(B > 0 && C > 0 && E > 0 && E % 16 == 0, CLIPMI_ERR_SHAPE, "fused_tail: B=%d C=%d E=%d unsupported (E %% 16 == 0)", B, C, E);
  CLIPMI_REQUIRE((int64_t)B * C * 4 < 0xFFFFFFF0ll, CLIPMI_ERR_SHAPE, "fused_tail: logits matrix too large for 32-bit offsets");
  CLIPMI_REQUIRE((uintptr_t)img % 16 == 0 && (uintptr_t)txt_n % 16 == 0, CLIPMI_ERR_ARG, "fused_tail: features must be 16-byte aligned");
  CLIPMI_REQUIRE(!bins || (labels && n_bins > 0 && n_bins <= 1024), CLIPMI_ERR_ARG, "fused_tail: ECE bins need labels and 1 <= n_bins <= 1024");
  CLIPMI_REQUIRE(normalize || !img_n_out, CLIPMI_ERR_ARG, "fused_tail: img_n_out only with normalize");
  const int lds = 16 * (E * 4 + 16) + 16;
  const bool fits = E % 64 == 0 && lds <= 160 * 1024 && (!bins || 8192 + 3 * (n_bins + 1) * 8 <= 16 * (E * 4 + 16));
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
  CLIPMI_REQUIRE(workspace && workspace_bytes >= fused_tail_workspace_bytes(B, C), CLIPMI_ERR_WORKSPACE,
                 "fused_tail: workspace too small (%zu < %zu)", workspace_bytes, fused_tail_workspace_bytes(B, C));
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
