// k-nearest-neighbour L2 distances of query embeddings to a reference set (SURVEY f-2; reference
// trainers/calibration/proximity.py:19-70: a Python loop of torch.norm(refs - q, dim=1) + topk per query).
//
// One workgroup owns 8 queries and streams the whole reference set through LDS in 64-row x 64-column tiles (coalesced
// 256-B row segments in, padded rows out -> conflict-free column reads).  Thread t accumulates sum_e (q - r)^2 -- the
// difference form, like the reference, no |q|^2+|r|^2-2qr cancellation -- for reference row (t & 63) against queries
// 2*(t >> 6) and 2*(t >> 6) + 1, keeps a sorted K-list per query in registers, and the 64 lists of a query are merged
// through LDS at the end.  fp32 VALU bound: 3 * Nq * Nr * E flop; the reference set is re-read from L2 once per 8 queries.
#include "common.h"

namespace clipmi {
namespace {

constexpr int QB = 8, RT = 64, EC = 64, KMAX = 16;

__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ q, const float* __restrict__ refs,
                                                  float* __restrict__ out, int Nq, int Nr, int E, int K) {
  __shared__ float rs[RT][EC + 1];
  __shared__ float qs[QB][EC];
  __shared__ float cand[QB][RT * KMAX];
  const int tid = threadIdx.x;
  const int r = tid & 63, g = tid >> 6;     // reference row in the tile, query pair
  const int q0 = blockIdx.x * QB;
  float best[2][KMAX];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) best[a][k] = INFINITY;

  for (int r0 = 0; r0 < Nr; r0 += RT) {
    float d0 = 0.f, d1 = 0.f;
    for (int e0 = 0; e0 < E; e0 += EC) {
      __syncthreads();
      // stage refs[r0 .. r0+63][e0 .. e0+63] and the 8 query segments
      for (int i = tid; i < RT * (EC / 4); i += 256) {
        const int rr = i / (EC / 4), c4 = i % (EC / 4);
        const int row = r0 + rr < Nr ? r0 + rr : Nr - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(refs + (int64_t)row * E + e0 + c4 * 4);
        rs[rr][c4 * 4 + 0] = v[0]; rs[rr][c4 * 4 + 1] = v[1]; rs[rr][c4 * 4 + 2] = v[2]; rs[rr][c4 * 4 + 3] = v[3];
      }
      for (int i = tid; i < QB * EC; i += 256) {
        const int qq = i / EC, c = i % EC;
        const int row = q0 + qq < Nq ? q0 + qq : Nq - 1;
        qs[qq][c] = q[(int64_t)row * E + e0 + c];
      }
      __syncthreads();
#pragma unroll 16
      for (int c = 0; c < EC; ++c) {
        const float rv = rs[r][c];
        const float a0 = qs[2 * g][c] - rv, a1 = qs[2 * g + 1][c] - rv;
        d0 = fmaf(a0, a0, d0);
        d1 = fmaf(a1, a1, d1);
      }
    }
    if (r0 + r < Nr) {   // sorted insertion (ascending), K <= KMAX
      float v0 = d0, v1 = d1;
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const float b0 = best[0][k], b1 = best[1][k];
          best[0][k] = fminf(b0, v0); v0 = fmaxf(b0, v0);
          best[1][k] = fminf(b1, v1); v1 = fmaxf(b1, v1);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    cand[2 * g][r * KMAX + k] = k < K ? best[0][k] : INFINITY;
    cand[2 * g + 1][r * KMAX + k] = k < K ? best[1][k] : INFINITY;
  }
  __syncthreads();
  // merge: wave w handles queries 2w, 2w+1; K rounds of "take the global minimum head" over the 64 sorted lists
  const int lane = tid & 63, w = tid >> 6;
  for (int a = 0; a < 2; ++a) {
    const int qq = 2 * w + a;
    int head = 0;   // this lane's list position
    for (int k = 0; k < K; ++k) {
      float v = head < KMAX ? cand[qq][lane * KMAX + head] : INFINITY;
      float m = v;
      int ml = lane;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(m, o, 64);
        const int ol = __shfl_xor(ml, o, 64);
        if (ov < m || (ov == m && ol < ml)) { m = ov; ml = ol; }
      }
      if (lane == ml) ++head;
      if (lane == 0 && q0 + qq < Nq) out[(int64_t)(q0 + qq) * K + k] = sqrtf(m);
    }
  }
}

}  // namespace

int launch_knn(const float* q, const float* refs, float* out, int Nq, int Nr, int E, int K, hipStream_t s) {
  if (Nq == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(q && refs && out, CLIPMI_ERR_ARG, "knn: null pointer");
  CLIPMI_REQUIRE(Nq > 0 && Nr > 0 && E > 0 && E % EC == 0, CLIPMI_ERR_SHAPE, "knn: Nq=%d Nr=%d E=%d (E %% 64 == 0)", Nq, Nr, E);
  CLIPMI_REQUIRE(K >= 1 && K <= KMAX && K <= Nr, CLIPMI_ERR_SHAPE, "knn: K=%d must be in [1, min(%d, Nr)]", K, KMAX);
  CLIPMI_REQUIRE((uintptr_t)q % 16 == 0 && (uintptr_t)refs % 16 == 0, CLIPMI_ERR_ARG, "knn: unaligned pointer");
  hipLaunchKernelGGL(knn_kernel, dim3((Nq + QB - 1) / QB), dim3(256), 0, s, q, refs, out, Nq, Nr, E, K);
  return check_launch("knn_kernel");
}

}  // namespace clipmi
