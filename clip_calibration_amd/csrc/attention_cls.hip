// Attention for ONE query per sequence: the class token's row of the image tower's LAST block (round 6).
//
// The reference reads only x[:, 0, :] behind the last block (clip/model.py:419: ln_post on the class token), so of that block's attention
// (clip/model.py:181-183) only the class token's output row can reach the features: one query per (sequence, head) against every key.
// Same softmax(q k^T / sqrt(64)) v as attention.hip, fp32 scores / probabilities / accumulation; 2/3 of the packed q | k | v rows are read once
// (K and V of every token: 155 MB at 256 images of ViT-B/16) and one 128-byte row per (sequence, head) is written -- an HBM-bound pass.
//
// One wave per (sequence, head).  Lane (r, c) = (lane / 8, lane % 8) walks the keys r, r + 8, ... and owns the 8-dimension chunk c of a head's 64:
// a wave-instruction loads 8 whole key (value) rows of 128 bytes, 16 bytes per lane.  Scores: v_dot2 over the lane's chunk, summed over the 8 lanes
// of a row (DPP row operations inside 8-lane groups); an ONLINE softmax per lane row-group r (running maximum, running sum, 8 accumulators), the 8
// groups merged at the end as a split softmax is (ds_bpermute shuffles: 3 rounds).  No LDS, no barrier, 40 registers: 16 waves per SIMD-pair keep
// ~2 KiB per wave in flight.
#include "common.h"

namespace clipmi {
namespace {

__device__ __forceinline__ float group8_sum(float v) {   // sum over the 8 lanes that share a key row (lanes 8 r .. 8 r + 7)
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}

__global__ __launch_bounds__(256) void attention_cls_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int L, int H, int n_items) {
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= n_items) return;   // wave-uniform
  const int n = item / H, h = item - n * H;
  const int r = lane >> 3, c = lane & 7;
  const int64_t D = (int64_t)H * 64, D3 = 3 * D;
  const half_t* base = qkv + (int64_t)n * L * D3 + h * 64 + c * 8;
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const f16x8 q8 = *reinterpret_cast<const f16x8*>(base);   // the class token is row 0 of its sequence
  const h2 q[4] = {h2{q8[0], q8[1]}, h2{q8[2], q8[3]}, h2{q8[4], q8[5]}, h2{q8[6], q8[7]}};
  const float scale = 0.125f * 1.4426950408889634f;          // 1 / sqrt(64), and exp(x) = 2^(x log2 e)
  float m = -INFINITY, l = 0.f;
  float acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acc[d] = 0.f;
  const half_t* kp = base + D + (int64_t)r * D3;
  const half_t* vp = base + 2 * D + (int64_t)r * D3;
  for (int row = r; row < L; row += 8, kp += 8 * D3, vp += 8 * D3) {
    const f16x8 k8 = *reinterpret_cast<const f16x8*>(kp);
    const f16x8 v8 = *reinterpret_cast<const f16x8*>(vp);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s = __builtin_amdgcn_fdot2(h2{k8[2 * t], k8[2 * t + 1]}, q[t], s, false);
    s = group8_sum(s) * scale;
    const float mn = fmaxf(m, s);
    const float corr = __builtin_amdgcn_exp2f(m - mn);   // first row of the group: 2^(-inf) = 0 against l = acc = 0
    const float p = __builtin_amdgcn_exp2f(s - mn);
    m = mn;
    l = l * corr + p;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = acc[d] * corr + p * (float)v8[d];
  }
  // merge the 8 row groups (lanes with equal c): a group that saw no row (L < 8) carries m = -inf, l = 0 and drops out
  float mg = m;
  mg = fmaxf(mg, __shfl_xor(mg, 8));
  mg = fmaxf(mg, __shfl_xor(mg, 16));
  mg = fmaxf(mg, __shfl_xor(mg, 32));
  const float f = __builtin_amdgcn_exp2f(m - mg);
  l *= f;
  l += __shfl_xor(l, 8);
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  const float inv = 1.0f / l;
  f16x8 o;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    float a = acc[d] * f;
    a += __shfl_xor(a, 8);
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 32);
    o[d] = (half_t)(a * inv);
  }
  if (r == 0) *reinterpret_cast<f16x8*>(out + (int64_t)n * L * D + h * 64 + c * 8) = o;   // 8 lanes x 16 B: the head's 128-byte row
}

}  // namespace

// out[n * L, h * 64 ..] = attention output of token 0 of sequence n (token-major [N * L, D] buffers as launch_attention; the other rows of `out`
// are left as they are)
int launch_attention_cls(const half_t* qkv, half_t* out, int N, int L, int H, hipStream_t s) {
  if (N == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(qkv && out, CLIPMI_ERR_ARG, "attention_cls: null pointer");
  CLIPMI_REQUIRE(N > 0 && L > 0 && H > 0 && (int64_t)N * H < (1ll << 31), CLIPMI_ERR_SHAPE, "attention_cls: bad shape N=%d L=%d H=%d", N, L, H);
  CLIPMI_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 16 == 0, CLIPMI_ERR_ARG, "attention_cls: unaligned pointer");
  const int n_items = N * H;
  hipLaunchKernelGGL(attention_cls_kernel, dim3((n_items + 3) / 4), dim3(256), 0, s, qkv, out, L, H, n_items);
  return check_launch("attention_cls_kernel");
}

}  // namespace clipmi
