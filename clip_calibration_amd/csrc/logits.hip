// Tail of the hot path (SURVEY a-10..a-12), fp32 end to end so that logits match the fp32 oracle given the same features:
//
//   cosine_logits_kernel   logits[B,C] = scale * img_n @ txt_n^T        (zsclip.py:100-101; coop.py:215-217)
//                          fp16 matrix cores on hi/lo-split fp32 operands (fp32 accuracy), operands straight from global/L2
//   row_calibrate_kernel   pred = argmax_c ; logits[i,:] *= dac_conf[pred]  (distanse_aware_calibration.py:49-58)
//                          conf = max_c softmax(logits[i,:])               (vl_calibrator.py:91; vl_evaluator.py:68,83)
//   ece_accumulate_kernel  per-bin (count, sum conf, sum correct)          (tools/metrics.py:90-130)
//
// HBM-bound: algorithmic bytes = 4*(B*E + C*E) read + 4*B*C written (+ 4*B*C re-read/re-written from L2 when DAC is on).
#include <type_traits>

#include "common.h"

namespace clipmi {
namespace {

// ---------------------------------------------------------------------------------------------------------------
// The dot products of the tail on the fp16 matrix cores at fp32 accuracy: every fp32 operand is split into
//     x = hi + lo,   hi = fp16(x),  lo = fp16(x - hi)          (|x| <= 1: L2-normalised features)
// and a product block is accumulated (fp32 accumulators, v_mfma_f32_16x16x32_f16) as
//     acc += hi_a . hi_b ;  acc += hi_a . lo_b ;  acc += lo_a . hi_b          (in this order, per 32-wide k-step)
// The dropped lo.lo term and the residual of lo are ~2^-22 of a product: on scaled logits (|logit| <= 100, 512 terms of
// ~1/512) that is ~1e-6 absolute, against 3e-4 allowed by the oracle tests -- and 3 MFMAs of the 2.5 PFLOP/s pipe replace 8
// of the exact-f32 one (v_mfma_f32_16x16x4_f32, 1/16 of the rate): at B = 2048, C = 1000 the exact form alone needs 13.5 us of
// matrix-pipe time, more than the whole 14.4 MB of the tail take to move at 1 TB/s.  An output element depends only on its
// own row and column data and on the k order, never on where it sits in a tile: results are bit-identical between the fused
// kernel, this kernel and any batch composition (the equivariance tests rely on it).
// ---------------------------------------------------------------------------------------------------------------
struct SplitFrag { f16x8 hi, lo; };
__device__ __forceinline__ SplitFrag split8(const f32x4& p, const f32x4& q) {
  SplitFrag s;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const half_t h0 = (half_t)p[e], h1 = (half_t)q[e];
    s.hi[e] = h0;
    s.hi[4 + e] = h1;
    s.lo[e] = (half_t)(p[e] - (float)h0);
    s.lo[4 + e] = (half_t)(q[e] - (float)h1);
  }
  return s;
}
__device__ __forceinline__ f32x4 split_mfma(const f16x8& a_hi, const f16x8& a_lo, SplitFrag b, f32x4 acc) {
  CLIPMI_VALU_TO_MFMA_FENCE2(b.hi, b.lo);   // split8's conversions wrote them a moment ago
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, b.hi, acc, 0, 0, 0);
  return acc;
}

// Separate-launch form (A/B aid and the shapes the fused kernel does not take): one wave = 16 rows x 16 classes, operands
// straight from global memory.  Lane l holds row / class (l & 15), k = k0 + 8 (l >> 4) .. + 7 of a 32-wide step; E % 32 == 0
// runs on the matrix cores, the remainder of other widths (E % 16 == 0) as an exact-f32 tail step of 16.
__global__ __launch_bounds__(256) void cosine_logits_kernel(const float* __restrict__ img, const float* __restrict__ txt,
                                                            float scale, float* __restrict__ logits, int B, int C, int E) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 16;
  const int n0 = blockIdx.y * 64 + wave * 16;
  if (n0 >= C) return;  // wave-uniform
  const int r = lane & 15, g = lane >> 4;
  const int m = m0 + r < B ? m0 + r : B - 1;
  const int n = n0 + r < C ? n0 + r : C - 1;
  const float* ap = img + (int64_t)m * E + g * 8;
  const float* bp = txt + (int64_t)n * E + g * 8;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  int k0 = 0;
  for (; k0 + 32 <= E; k0 += 32) {
    SplitFrag a = split8(*reinterpret_cast<const f32x4*>(ap + k0), *reinterpret_cast<const f32x4*>(ap + k0 + 4));
    const SplitFrag b = split8(*reinterpret_cast<const f32x4*>(bp + k0), *reinterpret_cast<const f32x4*>(bp + k0 + 4));
    CLIPMI_VALU_TO_MFMA_FENCE2(a.hi, a.lo);
    acc = split_mfma(a.hi, a.lo, b, acc);
  }
  if (k0 < E) {   // E % 32 == 16: one exact-f32 step (lane: k = k0 + 4 (l >> 4) .. + 3)
    const f32x4 a = *reinterpret_cast<const f32x4*>(img + (int64_t)m * E + k0 + g * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(txt + (int64_t)n * E + k0 + g * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
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

// ---------------------------------------------------------------------------------------------------------------
// The row pass: argmax, DAC factor, softmax top-1.
//
// Without DAC the softmax denominator is defined BLOCKWISE, so that the fused kernel can form it from what each of its
// (row block, 64-column block) workgroups sees, without reading the logits back:
//     block t = columns 64 t .. 64 t + 63:   m_t = max, a_t = argmax (lowest index), s_t = wave_sum(exp(x - m_t))
//     M = max_t m_t (first maximum: lowest index),   sum = s_0 exp(m_0 - M) + s_1 exp(m_1 - M) + ...  (t ascending),  conf = 1 / sum
// One wave per row, lane l holds column 64 t + l in iteration t, computes exactly that.  With DAC (the row is re-scaled by
// f = dac[argmax] in place, so the logits are read back anyway) the sum runs lane-strided over the scaled row as before:
// sum = wave_sum_l(sum_t exp(f x - f M)).  The two forms agree to a few 1e-7 relative.
// probs == nullptr: DAC scales the row in place (predict of distanse_aware_calibration.py); probs != nullptr: the row is
// left alone and softmax(row * f) is written to probs (which may be the same buffer).  No __restrict__ on these two.
// ---------------------------------------------------------------------------------------------------------------
struct TailPartial { float m; int arg; float s; int pad; };   // one (row, column block) of the blockwise form

__device__ __forceinline__ TailPartial block_partial(float x, int col, bool live) {   // x: this lane's logit of the block (live: col < C)
  TailPartial p;
  p.m = live ? x : -INFINITY;
  p.arg = live ? col : 0x7fffffff;
  wave_argmax(p.m, p.arg);
  // a block whose live columns are all -inf (masked classes) contributes nothing: exp(-inf - -inf) would be NaN, where the lane-strided
  // form (and torch.softmax) has exp(-inf - M) = 0 for those columns.  p.m is wave-uniform here.
  p.s = wave_sum(live && p.m != -INFINITY ? __expf(x - p.m) : 0.f);
  p.pad = 0;
  return p;
}
// merge in ascending block order; call with every block of the row
__device__ __forceinline__ void merge_begin(float& M, int& arg, float& sum) { M = -INFINITY; arg = 0x7fffffff; sum = 0.f; }
__device__ __forceinline__ void merge_max(const TailPartial& p, float& M, int& arg) {
  if (p.m > M) { M = p.m; arg = p.arg; }   // ascending blocks: the first maximum keeps the lowest index
}
__device__ __forceinline__ void merge_sum(const TailPartial& p, float M, float& sum) { sum = __builtin_fmaf(p.s, __expf(p.m - M), sum); }

__device__ __forceinline__ void calibrate_row(float* lr, const float* __restrict__ dac, int C, float* pr, int lane, float& conf_out, int& pred_out) {
  if (!dac) {   // uniform: the blockwise form
    const int nt = (C + 63) >> 6;
    float M, sum;
    int arg;
    merge_begin(M, arg, sum);
    for (int t = 0; t < nt; ++t) {   // pass 1: the row maximum (first occurrence)
      const int c = t * 64 + lane;
      float v = c < C ? lr[c] : -INFINITY;
      int a = c < C ? c : 0x7fffffff;
      wave_argmax(v, a);
      if (v > M) { M = v; arg = a; }
    }
    if (arg == 0x7fffffff) arg = 0;   // all-NaN row
    for (int t = 0; t < nt; ++t) {   // pass 2: block sums against the block maxima, merged against the row maximum
      const int c = t * 64 + lane;
      const TailPartial p = block_partial(c < C ? lr[c] : 0.f, c, c < C);
      merge_sum(p, M, sum);
    }
    if (pr) {
      const float inv = 1.0f / sum;
      for (int c = lane; c < C; c += 64) pr[c] = __expf(lr[c] - M) * inv;   // same lane reads then writes element c
    }
    conf_out = 1.0f / sum;
    pred_out = arg;
    return;
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = lr[c];
    if (v > best) { best = v; bi = c; }   // first occurrence wins inside a lane (ascending c)
  }
  wave_argmax(best, bi);
  if (bi == 0x7fffffff) bi = 0;  // all-NaN row
  const float f = dac[bi];
  const float mx = best * f;
  float se = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = lr[c] * f;
    if (!pr) lr[c] = v;
    se += __expf(v - mx);
  }
  se = wave_sum(se);
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
// Grid: (B/32) x (C/64) workgroups of 4 waves (512 at B = 2048, C = 1000; two per CU).  A workgroup
//   0. starts the loads of its text fragments (each wave: its 16 classes, the first k-steps), then normalises its 32 image
//      rows (the arithmetic of l2norm_kernel, element for element: lane-strided sum, butterfly) and leaves them in LDS ALREADY
//      SPLIT into fp16 hi / lo halves (see split8 above), 16 bytes per lane and row chunk;
//   1. runs, per wave, a 32 x 16 block of the product on the fp16 matrix cores (3 MFMAs per 16 x 16 x 32 step, fp32 accuracy):
//      A fragments by ds_read_b128, text fragments from registers (split on the fly, next group of k-steps in flight);
//   2. stores its 32 x 64 logits through an LDS tile as 16-byte write-through (sc1) stores;
//   3. draws a ticket for its 32-row block; the LAST workgroup to arrive (agent-scope hand-off: cdna_hip_programming.md
//      Guideline 16, counter form) runs the row pass -- argmax, DAC factor, softmax top-1, ECE bins -- over the block's 32
//      complete rows, 8 per wave, with the arithmetic of row_calibrate_kernel, and resets the counter.
// Results are bit-identical to the three-kernel path.  The first version (16-row blocks, exact-f32 MFMAs, text fragments read
// from L2 one k-step ahead) took 24.7 us at B = 256 and 55.6 us at B = 2048 (260 GB/s): 13.5 us of that is the f32 matrix pipe
// alone, the rest the 128 x 2 MB of text re-reads and a chain of dependent L2 round trips.
// counters: ceil(B/32) int32 of the workspace, zero before the first launch, left zero by every launch.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TAIL_TILE_LD = 64 + 4;        // floats per row of the LDS logits tile
constexpr int tail_sbins_off(int rb) { return rb * TAIL_TILE_LD * 4 + 16; }

// RB = image rows per workgroup: 32 for large batches (half the text re-reads from L2), 16 for small ones (twice the
// workgroups, and a row pass of 4 instead of 8 rows per wave on the critical path of a launch that is all latency)
template <bool NORMALIZE, typename TI, int RB>
__global__ __launch_bounds__(256, 2) void fused_tail_kernel(const TI* __restrict__ img, const float* __restrict__ txt, float scale,
                                                            const float* __restrict__ dac, float* logits, float* __restrict__ img_n_out,
                                                            float* __restrict__ conf, int32_t* __restrict__ pred,
                                                            const int64_t* __restrict__ labels, double* __restrict__ bins, int n_bins,
                                                            int* counters, TailPartial* partials, int B, int C, int E, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int rsb = E * 2 + 16;                         // bytes per fp16 row of an LDS operand image (16-byte pad)
  char* hi_s = smem;
  constexpr int RPW = RB / 4;                         // rows per wave in the row-wise phases
  constexpr int NRB = RB / 16;                        // 16-row MFMA blocks per wave
  char* lo_s = smem + RB * rsb;
  int* flag = reinterpret_cast<int*>(smem + lds_bytes - 16);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * RB;
  const int n0 = blockIdx.y * 64 + wave * 16;
  const int r = lane & 15, g = lane >> 4;
  const int nsteps = E >> 5;                          // 32-wide k-steps (E % 64 == 0)
  constexpr int G = 4;                                // k-steps per register group of text fragments

  // ---- text fragments of the first group: in flight during phase 0
  const bool cols_live = n0 < C;                      // wave-uniform
  const float* bp = txt + (int64_t)(n0 + r < C ? n0 + r : C - 1) * E + g * 8;
  f32x4 b0[G][2], b1[G][2];
  auto load_group = [&](f32x4 (&dst)[G][2], int s0) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (s0 + u < nsteps) {                          // uniform
        dst[u][0] = *reinterpret_cast<const f32x4*>(bp + (s0 + u) * 32);
        dst[u][1] = *reinterpret_cast<const f32x4*>(bp + (s0 + u) * 32 + 4);
      }
    }
  };
  // (requesting ALL 16 k-steps' fragments here -- 128 registers, one trip -- was measured 2x SLOWER at B >= 1024: VMEM returns in
  // order, so phase 0's image rows then wait behind 32 text loads per lane, and every workgroup floods L2 at once)

  // ---- phase 0: the RB image rows, RPW per wave, ONE trip to memory: a lane takes 8 consecutive elements per 512-element
  //      stripe (16 / 32-byte loads), sums their squares in ascending order (l2norm_kernel's order for E % 8 == 0: fused
  //      multiply-adds, then the butterfly) and writes x * inv back out and, split, into LDS as 16 bytes per lane.
  {
    auto load8 = [&](const TI* x, f32x4& p, f32x4& q) {
      if constexpr (sizeof(TI) == 4) {
        p = *reinterpret_cast<const f32x4*>(x);
        q = *reinterpret_cast<const f32x4*>(x + 4);
      } else {
        const f16x8 h = *reinterpret_cast<const f16x8*>(x);
        p = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        q = f32x4{(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
      }
    };
    auto emit = [&](int rl, int row, int c, f32x4 p, f32x4 q, float inv) {
      if constexpr (NORMALIZE) {
        p *= inv;
        q *= inv;
      }
      if (img_n_out != nullptr && blockIdx.y == 0 && m0 + rl < B) {   // wave-uniform
        float* o = img_n_out + (int64_t)row * E + c * 8;
        *reinterpret_cast<f32x4*>(o) = p;
        *reinterpret_cast<f32x4*>(o + 4) = q;
      }
      const SplitFrag sp = split8(p, q);
      *reinterpret_cast<f16x8*>(hi_s + rl * rsb + c * 16) = sp.hi;
      *reinterpret_cast<f16x8*>(lo_s + rl * rsb + c * 16) = sp.lo;
    };
    const int nch = E >> 3;                           // 8-element chunks per row
    if (nch <= 128) {   // uniform (E <= 1024): every row of the wave in registers between the norm and the scaling
      constexpr int NU = 2;                           // 512-element stripes per row held in registers
      f32x4 xp[RPW][NU], xq[RPW][NU];
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int rl = wave * RPW + rr;
          const int row = m0 + rl < B ? m0 + rl : B - 1;
          if (lane + 64 * u < nch) load8(img + (int64_t)row * E + (lane + 64 * u) * 8, xp[rr][u], xq[rr][u]);
        }
      if (cols_live) load_group(b0, 0);   // behind the image rows (VMEM returns in order): in flight during the norms and the split
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        const int rl = wave * RPW + rr;
        const int row = m0 + rl < B ? m0 + rl : B - 1;
        float inv = 1.0f;
        if constexpr (NORMALIZE) {
          float ss = 0.f;
#pragma unroll
          for (int u = 0; u < NU; ++u)
            if (lane + 64 * u < nch) {
#pragma unroll
              for (int e = 0; e < 4; ++e) ss = __builtin_fmaf(xp[rr][u][e], xp[rr][u][e], ss);
#pragma unroll
              for (int e = 0; e < 4; ++e) ss = __builtin_fmaf(xq[rr][u][e], xq[rr][u][e], ss);
            }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
          inv = 1.0f / sqrtf(ss);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (lane + 64 * u < nch) emit(rl, row, lane + 64 * u, xp[rr][u], xq[rr][u], inv);
      }
    } else {
      if (cols_live) load_group(b0, 0);
      for (int rr = 0; rr < RPW; ++rr) {
        const int rl = wave * RPW + rr;
        const int row = m0 + rl < B ? m0 + rl : B - 1;
        const TI* x = img + (int64_t)row * E;
        float inv = 1.0f;
        if constexpr (NORMALIZE) {
          float ss = 0.f;
          for (int c = lane; c < nch; c += 64) {
            f32x4 p, q;
            load8(x + c * 8, p, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss = __builtin_fmaf(p[e], p[e], ss);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss = __builtin_fmaf(q[e], q[e], ss);
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
          inv = 1.0f / sqrtf(ss);
        }
        for (int c = lane; c < nch; c += 64) {
          f32x4 p, q;
          load8(x + c * 8, p, q);
          emit(rl, row, c, p, q, inv);
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 1: RB x 16 block per wave
  f32x4 acc[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (cols_live) {
    const char* ah = hi_s + r * rsb + g * 16;
    const char* al = lo_s + r * rsb + g * 16;
    auto run_group = [&](const f32x4 (&bq)[G][2], int s0) {
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (s0 + u < nsteps) {                        // uniform
          const SplitFrag b = split8(bq[u][0], bq[u][1]);
          const int kb = (s0 + u) * 64;               // byte offset of the k-step inside a row
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) {
            const f16x8 a_hi = *reinterpret_cast<const f16x8*>(ah + rb * 16 * rsb + kb);
            const f16x8 a_lo = *reinterpret_cast<const f16x8*>(al + rb * 16 * rsb + kb);
            acc[rb] = split_mfma(a_hi, a_lo, b, acc[rb]);
          }
        }
      }
    };
    {
      for (int s0 = 0; s0 < nsteps; s0 += 2 * G) {
        if (s0 + G < nsteps) load_group(b1, s0 + G);
        run_group(b0, s0);
        if (s0 + 2 * G < nsteps) load_group(b0, s0 + 2 * G);
        if (s0 + G < nsteps) run_group(b1, s0 + G);
      }
    }
  }
  // ---- phase 2: the workgroup's RB x 64 logits tile, through an LDS tile.  Rows 16-byte aligned (C % 4 == 0): 16-byte
  //      write-through (sc1) stores -- whole 64-byte row pieces, and no release fence is needed for the hand-off below
  //      (cdna_hip_programming.md Guideline 16, R1).  Ragged class counts: 4-byte plain stores of the same pieces (a release only
  //      where the last workgroup reads the logits back, i.e. with DAC).
  const bool wide = (C & 3) == 0;   // uniform
  {
    __syncthreads();                // every wave is done reading the image rows: the tile takes their place
    float* tile = reinterpret_cast<float*>(smem);
    // D layout: col = lane&15 (n), row = (lane>>4)*4 + reg (m)
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[(rb * 16 + g * 4 + e) * TAIL_TILE_LD + wave * 16 + r] = scale * acc[rb][e];
    __syncthreads();
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t lrs = make_rsrc(logits, (int64_t)B * C * 4);
#pragma unroll
    for (int h = 0; h < NRB; ++h) {
      const int pidx = h * 256 + threadIdx.x;
      const int trow = pidx >> 4, tq = pidx & 15;
      const int mm = m0 + trow, nn = blockIdx.y * 64 + tq * 4;
      if (mm < B && nn < C) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(tile + trow * TAIL_TILE_LD + tq * 4);
        if (wide) {                 // C % 4 == 0: rows are 16-byte aligned and a 4-column piece is entirely inside or outside
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), lrs, (int)(((int64_t)mm * C + nn) * 4), 0, 16 /* sc1 */);
        } else {                    // ragged class counts (e.g. 199): 4-byte stores, the same 64-byte row pieces per 16 threads
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (nn + e < C) logits[(int64_t)mm * C + nn + e] = v[e];
        }
      }
    }
    // ---- phase 2b (no DAC): this workgroup's share of the row pass, from the tile while it is still in LDS -- per row the block's
    //      (max, argmax, sum of exp(x - max)), 16 bytes, write-through.  The last workgroup of the row block merges n_col_blocks of
    //      them per row instead of reading RB x C logits back (the read-back chain was 14 of the kernel's 23 us at B = 256).
    if (!dac && (conf || pred || bins)) {   // uniform
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const __amdgpu_buffer_rsrc_t prs = make_rsrc(partials, (int64_t)B * gridDim.y * (int)sizeof(TailPartial));
      const int col = blockIdx.y * 64 + lane;
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
        const int trow = wave * RPW + q;
        const TailPartial p = block_partial(tile[trow * TAIL_TILE_LD + lane], col, col < C);
        if (lane == 0 && m0 + trow < B)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, p), prs, (int)(((int64_t)(m0 + trow) * gridDim.y + blockIdx.y) * (int)sizeof(TailPartial)), 0, 16 /* sc1 */);
      }
    }
  }
  if (!dac && !conf && !pred && !bins) return;   // logits only (kernel argument: uniform)

  // ---- phase 3: ticket; the last workgroup of this row block owns the row pass over the logits the others stored
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!wide && dac) {   // plain logits stores that the last workgroup reads back: write this XCD's L2 back before the ticket (write-through stores need no release)
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
  // ECE bins of this row block: summed in LDS first, then at most 3 * (n_bins + 1) global atomics per row block -- one set of
  // three per ROW on a dozen hot addresses serialised the whole grid (B = 2048: 41 us of the launch).
  double* sbins = reinterpret_cast<double*>(smem + tail_sbins_off(RB));
  if (bins) {
    for (int t = threadIdx.x; t < 3 * (n_bins + 1); t += 256) sbins[t] = 0.0;
    __syncthreads();
  }
  if (!dac) {   // uniform: merge the row block's partials (calibrate_row's blockwise form, block by block in ascending order)
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(partials, (int64_t)B * gridDim.y * (int)sizeof(TailPartial));
    const int row = m0 + threadIdx.x;   // thread t owns row t of the block
    if (threadIdx.x < RB && row < B) {
      float M, sum;
      int arg;
      merge_begin(M, arg, sum);
      const int nt = gridDim.y;
      const int base = (int)((int64_t)row * nt * (int)sizeof(TailPartial));
      if (nt <= 16) {   // uniform (C <= 1024): ONE trip, both passes from registers -- same order, same bits
        TailPartial p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)   // sc1 loads: every store of these bytes was sc1 and drained before its ticket
          if (u < nt) p[u] = __builtin_bit_cast(TailPartial, __builtin_amdgcn_raw_buffer_load_b128(prs, base + u * (int)sizeof(TailPartial), 0, 16 /* sc1 */));
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (u < nt) merge_max(p[u], M, arg);
        if (arg == 0x7fffffff) arg = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (u < nt) merge_sum(p[u], M, sum);
      } else {
      for (int t0 = 0; t0 < nt; t0 += 16) {
        TailPartial p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (t0 + u < nt) p[u] = __builtin_bit_cast(TailPartial, __builtin_amdgcn_raw_buffer_load_b128(prs, base + (t0 + u) * (int)sizeof(TailPartial), 0, 16 /* sc1 */));
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (t0 + u < nt) merge_max(p[u], M, arg);
      }
      if (arg == 0x7fffffff) arg = 0;
      for (int t0 = 0; t0 < nt; t0 += 16) {
        TailPartial p[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (t0 + u < nt) p[u] = __builtin_bit_cast(TailPartial, __builtin_amdgcn_raw_buffer_load_b128(prs, base + (t0 + u) * (int)sizeof(TailPartial), 0, 16 /* sc1 */));
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (t0 + u < nt) merge_sum(p[u], M, sum);
      }
      }
      const float cf = 1.0f / sum;
      if (conf) conf[row] = cf;
      if (pred) pred[row] = arg;
      if (bins) {
        const double x = (double)cf;
        const int b = ece_bin(x, n_bins), nb1 = n_bins + 1;
        atomicAdd(&sbins[b], 1.0);
        atomicAdd(&sbins[nb1 + b], x);
        atomicAdd(&sbins[2 * nb1 + b], (labels[row] == (int64_t)arg) ? 1.0 : 0.0);
      }
    }
  } else
  // DAC: the wave's RPW rows side by side: calibrate_row's arithmetic per row (same lane-strided order, same reductions), with
  // every load of the pass in flight at once.  C <= 1024: a lane's 16 elements of each row stay in registers between the
  // argmax and the softmax pass (one trip to L2 instead of two chains of dependent round trips).  Larger C: the same
  // arithmetic with the two passes reading memory.
  {
    const int row0 = m0 + wave * RPW;
    float* lr[RPW];
    float best[RPW], fac[RPW], mx[RPW], se[RPW];
    int bi[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      const int row = row0 + q < B ? row0 + q : B - 1;   // clamped rows recompute row B-1 and are not written
      lr[q] = logits + (int64_t)row * C;
      best[q] = -INFINITY;
      bi[q] = 0x7fffffff;
      se[q] = 0.f;
    }
    auto finish_argmax = [&](int q) {
      wave_argmax(best[q], bi[q]);
      if (bi[q] == 0x7fffffff) bi[q] = 0;
      fac[q] = dac ? dac[bi[q]] : 1.0f;
      mx[q] = best[q] * fac[q];
    };
    if (C <= 1024) {   // uniform
      float v[RPW][16];
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int q = 0; q < RPW; ++q) v[q][t] = lane + 64 * t < C ? lr[q][lane + 64 * t] : 0.f;
#pragma unroll
      for (int q = 0; q < RPW; ++q) {
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
        for (int q = 0; q < RPW; ++q) {
          const float x = lr[q][c];
          if (x > best[q]) { best[q] = x; bi[q] = c; }
        }
      }
#pragma unroll
      for (int q = 0; q < RPW; ++q) finish_argmax(q);
      for (int c = lane; c < C; c += 64) {
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
          const float x = lr[q][c] * fac[q];
          if (dac && row0 + q < B) lr[q][c] = x;
          se[q] += __expf(x - mx[q]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
      se[q] = wave_sum(se[q]);
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
  hipLaunchKernelGGL(cosine_logits_kernel, dim3((B + 15) / 16, (C + 63) / 64), dim3(256), 0, s, img_n, txt_n, scale, logits, B, C, E);
  int rc = check_launch("cosine_logits_kernel");
  if (rc != CLIPMI_OK) return rc;
  if (dac_conf || conf || pred) rc = launch_calibrate_rows(logits, dac_conf, conf, pred, B, C, s);
  return rc;
}

// Workspace: a FIXED 64 KiB of int32 ticket counters (one per 16- or 32-row block: batches up to 262 144 rows; zero between launches),
// then [B][ceil(C/64)] 16-byte row-pass partials (need no initialisation).  The counter region does not move with B: a caller may
// reuse one buffer for calls of different sizes, and a region that held another call's partials must never be read as counters.
constexpr size_t TAIL_COUNTER_BYTES = 64 * 1024;
size_t fused_tail_workspace_bytes(int B, int C) { return TAIL_COUNTER_BYTES + align256((size_t)B * ((C + 63) / 64) * sizeof(TailPartial)); }

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
  // LDS: the two split operand images (RB rows x (2 E + 16) bytes each), later overlaid by the logits tile and the ECE bins
  const int rb = B <= 512 ? 16 : 32;
  const int lds_ops = 2 * rb * (E * 2 + 16);
  const int lds_post = tail_sbins_off(rb) + (bins ? 3 * (n_bins + 1) * 8 : 0);
  const int lds = ((lds_ops > lds_post ? lds_ops : lds_post) + 15) / 16 * 16 + 16;   // + the flag
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
  CLIPMI_REQUIRE(workspace && workspace_bytes >= fused_tail_workspace_bytes(B, C), CLIPMI_ERR_WORKSPACE,
                 "fused_tail: workspace too small (%zu < %zu)", workspace_bytes, fused_tail_workspace_bytes(B, C));
  const dim3 grid((B + rb - 1) / rb, (C + 63) / 64);
  CLIPMI_REQUIRE(grid.y <= 65535, CLIPMI_ERR_SHAPE, "fused_tail: too many classes");
  int* counters = static_cast<int*>(workspace);
  CLIPMI_REQUIRE((size_t)((B + 15) / 16) * sizeof(int) <= TAIL_COUNTER_BYTES, CLIPMI_ERR_SHAPE, "fused_tail: batch too large for the ticket-counter region");
  TailPartial* partials = reinterpret_cast<TailPartial*>(static_cast<char*>(workspace) + TAIL_COUNTER_BYTES);
  CLIPMI_REQUIRE((int64_t)B * grid.y * (int64_t)sizeof(TailPartial) < 0x7FFFFFF0ll, CLIPMI_ERR_SHAPE, "fused_tail: partial table too large for 32-bit offsets");
  auto go = [&](auto kernel, DeviceOnce& once, auto* typed) {
    // the attribute is set once per (instantiation, device): to the CU's whole LDS, not to this call's size -- E and n_bins vary between calls
    ensure_dynamic_lds(kernel, 160 * 1024, once);
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, typed, txt_n, scale, dac_conf, logits, img_n_out, conf, pred, labels, bins, n_bins,
                       counters, partials, B, C, E, lds);
  };
  static DeviceOnce once[16];
  const half_t* img16 = static_cast<const half_t*>(img_);
  auto pick = [&](auto norm_tag, auto* typed, int slot) {
    constexpr bool NRM = decltype(norm_tag)::value;
    using TI = std::remove_cv_t<std::remove_pointer_t<decltype(typed)>>;
    if (rb == 32) go(fused_tail_kernel<NRM, TI, 32>, once[slot * 2 + 1], typed);
    else go(fused_tail_kernel<NRM, TI, 16>, once[slot * 2], typed);
  };
  if (img_dtype == CLIPMI_F32) {
    if (normalize) pick(std::true_type{}, img, 0);
    else pick(std::false_type{}, img, 1);
  } else {
    if (normalize) pick(std::true_type{}, img16, 2);
    else pick(std::false_type{}, img16, 3);
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
