// HBM-bound helper kernels of the towers: patchify (im2col + fp32->fp16 cast), class/prompt rows, token
// overwrite (MaPLe), positional add, token-embedding gather, casts, row L2 normalisation.
// All of them move 16 bytes per lane where the layout allows it.
#include "common.h"

namespace clipmi {
namespace {

// ---------------------------------------------------------------------------------------------------------
// patchify: image [B,3,R,R] -> col fp16 [B*G*G, Kpad], column k = c*P*P + ky*P + kx  (clip/model.py:598,395-397).
// One thread per 8 output columns (one 16-byte store); consecutive threads walk a col row, so stores are full
// lines and the loads are 8 consecutive pixels of one image row when P % 8 == 0.
// Algorithmic bytes per image: 3*R*R*sizeof(in) read + G*G*Kpad*2 written.
// ---------------------------------------------------------------------------------------------------------
template <typename TI, bool VEC>
__global__ __launch_bounds__(256) void patchify_kernel(const TI* __restrict__ img, half_t* __restrict__ col, int B, int R,
                                                       int P, int G, int Kpad, int64_t total_chunks) {
  const int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ch >= total_chunks) return;
  const int cpr = Kpad >> 3;  // chunks per col row
  const int64_t row = ch / cpr;
  const int k0 = (int)(ch - row * cpr) << 3;
  const int b = (int)(row / (G * G));
  const int pr = (int)(row - (int64_t)b * G * G);
  const int py = pr / G, px = pr - py * G;
  const int PP = P * P, K = 3 * PP;
  f16x8 o;
  if ((P & 7) == 0 && k0 + 8 <= K) {
    const int c = k0 / PP, rem = k0 - c * PP;
    const int ky = rem / P, kx = rem - ky * P;
    const TI* src = img + (((int64_t)b * 3 + c) * R + (py * P + ky)) * R + px * P + kx;
    if constexpr (VEC && sizeof(TI) == 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src);
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(src + 4);
      o = f16x8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)c4[0], (half_t)c4[1], (half_t)c4[2], (half_t)c4[3]};
    } else if constexpr (VEC) {
      o = *reinterpret_cast<const f16x8*>(src);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)(float)src[e];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      float v = 0.f;
      if (k < K) {
        const int c = k / PP, rem = k - c * PP;
        const int ky = rem / P, kx = rem - ky * P;
        v = (float)img[(((int64_t)b * 3 + c) * R + (py * P + ky)) * R + px * P + kx];
      }
      o[e] = (half_t)v;
    }
  }
  *reinterpret_cast<f16x8*>(col + row * Kpad + k0) = o;
}

// x0[b, 0, :] = cls + pos[0];  x0[b, tokens0 + j, :] = shallow[j]
__global__ __launch_bounds__(256) void cls_ctx_kernel(float* __restrict__ x0, const float* __restrict__ cls,
                                                      const float* __restrict__ pos, const float* __restrict__ shallow,
                                                      int tokens0, int n_ctx, int D) {
  const int b = blockIdx.x / (1 + n_ctx);
  const int j = blockIdx.x - b * (1 + n_ctx);
  const int L = tokens0 + n_ctx;
  float* dst = x0 + ((int64_t)b * L + (j == 0 ? 0 : tokens0 + j - 1)) * D;
  for (int d = threadIdx.x; d < D; d += 256) dst[d] = (j == 0) ? cls[d] + pos[d] : shallow[(int64_t)(j - 1) * D + d];
}

// x[n, first + j, :] = prompt[j, :]
__global__ __launch_bounds__(256) void overwrite_kernel(float* __restrict__ x, const float* __restrict__ prompt, int L, int D,
                                                        int first, int n_ctx) {
  const int n = blockIdx.x / n_ctx, j = blockIdx.x - n * n_ctx;
  float* dst = x + ((int64_t)n * L + first + j) * D;
  const float* src = prompt + (int64_t)j * D;
  for (int d = threadIdx.x; d < D; d += 256) dst[d] = src[d];
}

// x16[row,:] = fp16(x[row,:]); stats partial 0 = (sum, sum of squares), other partials 0: producer side of the LayerNorm fold for
// rows that no GEMM epilogue wrote (text-tower input, MaPLe token overwrite).  One wave per row.
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, half_t* __restrict__ x16,
                                                        float* __restrict__ stats, int parts, int64_t M, int L, int D, int first,
                                                        int n_ctx, int total) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= total) return;
  const int64_t row = (int64_t)(r / n_ctx) * L + first + (r % n_ctx);
  const float* xr = x + row * D;
  float s = 0.f, q = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    *reinterpret_cast<f16x4*>(x16 + row * D + c) = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane < parts) *reinterpret_cast<float2*>(stats + 2 * ((int64_t)lane * M + row)) = lane == 0 ? make_float2(s, q) : make_float2(0.f, 0.f);
}

// src rows are [C, src_L, D]; the first L <= src_L token rows of every sequence are taken (text tower: dead-row elimination)
template <typename TI>
__global__ __launch_bounds__(256) void add_pos_kernel(const TI* __restrict__ src, const float* __restrict__ pos,
                                                      float* __restrict__ xres, int L, int src_L, int D4, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int d4 = (int)(i % D4);
  const int64_t tok = i / D4;
  const int l = (int)(tok % L);
  const int64_t si = ((tok / L) * src_L + l) * D4 + d4;
  f32x4 v;
  if constexpr (sizeof(TI) == 4) {
    v = *reinterpret_cast<const f32x4*>(src + si * 4);
  } else {
    const f16x4 h = *reinterpret_cast<const f16x4*>(src + si * 4);
    v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  if (pos) v += *reinterpret_cast<const f32x4*>(pos + ((int64_t)l * D4 + d4) * 4);
  *reinterpret_cast<f32x4*>(xres + i * 4) = v;
}

__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                    const float* __restrict__ pos, float* __restrict__ xres, int L, int src_L, int D4,
                                                    int vocab, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int d4 = (int)(i % D4);
  const int64_t tok = i / D4;
  const int l = (int)(tok % L);
  int64_t id = ids[(tok / L) * src_L + l];   // ids are [C, src_L]; the first L token positions of every prompt are embedded
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
  const f32x4 e = *reinterpret_cast<const f32x4*>(table + (id * D4 + d4) * 4);
  const f32x4 p = *reinterpret_cast<const f32x4*>(pos + ((int64_t)l * D4 + d4) * 4);
  *reinterpret_cast<f32x4*>(xres + i * 4) = e + p;
}

// eot[c] = first argmax_l ids[c, l]   (text.argmax(dim=-1), clip/model.py:611) ; rows[c] = c*L + eot[c]
__global__ __launch_bounds__(64) void eot_kernel(const int64_t* __restrict__ ids, int32_t* __restrict__ eot,
                                                 int32_t* __restrict__ rows, int C, int L) {
  const int c = blockIdx.x;
  const int lane = threadIdx.x;
  int64_t best = INT64_MIN;
  int bi = 0x7fffffff;
  for (int l = lane; l < L; l += 64) {
    const int64_t v = ids[(int64_t)c * L + l];
    if (v > best) { best = v; bi = l; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) {
    if (eot) eot[c] = bi;
    if (rows) rows[c] = c * L + bi;
  }
}

__global__ __launch_bounds__(256) void rows_from_eot_kernel(const int32_t* __restrict__ eot, int32_t* __restrict__ rows, int C,
                                                            int L) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < C) {
    int e = eot[c];
    e = e < 0 ? 0 : (e >= L ? L - 1 : e);
    rows[c] = c * L + e;
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, TO* __restrict__ dst, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 4 <= n) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
    if constexpr (sizeof(TO) == 4) {
      *reinterpret_cast<f32x4*>(dst + i) = v;
    } else {
      *reinterpret_cast<f16x4*>(dst + i) = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    }
  } else {
    for (int64_t j = i; j < n; ++j) dst[j] = (TO)src[j];
  }
}

// y[c, l, :] = (TO)src[c * rows + l, :] for l < rows, 0 for rows <= l < L: the blocks' output back in the caller's [C, L, D] layout when only the first
// `rows` token rows of every sequence were computed (clipmi_text_blocks with seq_rows).  One thread per 4 elements.
template <typename TS, typename TO>
__global__ __launch_bounds__(256) void rows_out_kernel(const TS* __restrict__ src, TO* __restrict__ dst, int rows, int L, int D4, int64_t total4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int d4 = (int)(i % D4);
  const int64_t tok = i / D4;
  const int l = (int)(tok % L);
  const int64_t c = tok / L;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (l < rows) {
    const TS* p = src + ((c * rows + l) * D4 + d4) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)p[e];
  }
  TO* q = dst + i * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) q[e] = (TO)v[e];
}

template <typename TO>
__global__ __launch_bounds__(256) void cast16_kernel(const half_t* __restrict__ src, TO* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (TO)(float)src[i];
}

// out[r,:] = in[r,:] / ||in[r,:]||_2, one wave per row, fp32 math.
// Summation order (the fused tail of logits.hip forms its norms in exactly this order: its normalised rows are bit-identical):
// E % 8 == 0 and 16-byte aligned rows: a lane takes the 8-element chunks lane, lane + 64, ..., adds their squares in ascending
// order with fused multiply-adds; otherwise element lane, lane + 64, ...; then the xor butterfly 32, 16, ..., 1.
template <typename TI, typename TO = float>
__global__ __launch_bounds__(256) void l2norm_kernel(const TI* __restrict__ in, TO* __restrict__ out, int rows, int E, int vec8) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TI* x = in + (int64_t)row * E;
  float ss = 0.f;
  auto load8 = [&](const TI* p8, float (&v)[8]) {
    if constexpr (sizeof(TI) == 4) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(p8), q = *reinterpret_cast<const f32x4*>(p8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = p[e]; v[4 + e] = q[e]; }
    } else {
      const f16x8 h = *reinterpret_cast<const f16x8*>(p8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    }
  };
  if (vec8) {   // uniform
    for (int c = lane; c < (E >> 3); c += 64) {
      float v[8];
      load8(x + c * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(v[e], v[e], ss);
    }
  } else {
    for (int e = lane; e < E; e += 64) {
      const float v = (float)x[e];
      ss = __builtin_fmaf(v, v, ss);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.0f / sqrtf(ss);
  for (int e = lane; e < E; e += 64) out[(int64_t)row * E + e] = (TO)((float)x[e] * inv);   // TO = fp16: the fp32 value, rounded once
}

// out[g,:] = mean_p in[(g*P + p),:]  -- ProDA's prompt-ensemble mean of the normalised text features (proda.py:328-332)
__global__ __launch_bounds__(256) void group_mean_kernel(const float* __restrict__ in, float* __restrict__ out, int G, int P, int E) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)G * E) return;
  const int g = (int)(i / E), e = (int)(i % E);
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += in[((int64_t)g * P + p) * E + e];
  out[i] = s / (float)P;
}

}  // namespace

int launch_group_mean(const float* in, float* out, int G, int P, int E, hipStream_t s) {
  if (G == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(in && out, CLIPMI_ERR_ARG, "group_mean: null pointer");
  CLIPMI_REQUIRE(G > 0 && P > 0 && E > 0, CLIPMI_ERR_SHAPE, "group_mean: G=%d P=%d E=%d", G, P, E);
  const int64_t total = (int64_t)G * E;
  hipLaunchKernelGGL(group_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, G, P, E);
  return check_launch("group_mean_kernel");
}

int launch_patchify(const void* image, int image_dtype, half_t* col, int B, int R, int P, int Kpad, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(image && col, CLIPMI_ERR_ARG, "patchify: null pointer");
  CLIPMI_REQUIRE(B > 0 && P > 0 && R > 0 && R % P == 0, CLIPMI_ERR_SHAPE, "patchify: R=%d must be a multiple of P=%d", R, P);
  CLIPMI_REQUIRE(Kpad % 64 == 0 && Kpad >= 3 * P * P, CLIPMI_ERR_SHAPE, "patchify: Kpad=%d must be a multiple of 64 >= 3*P*P", Kpad);
  CLIPMI_REQUIRE((uintptr_t)col % 16 == 0, CLIPMI_ERR_ARG, "patchify: col must be 16-byte aligned");
  const int G = R / P;
  const int64_t total = (int64_t)B * G * G * (Kpad / 8);
  const unsigned grid = (unsigned)((total + 255) / 256);
  const bool vec = (P % 8 == 0) && (R % 8 == 0) && ((uintptr_t)image % 16 == 0);
  if (image_dtype == CLIPMI_F32) {
    if (vec)
      hipLaunchKernelGGL((patchify_kernel<float, true>), dim3(grid), dim3(256), 0, s, (const float*)image, col, B, R, P, G, Kpad, total);
    else
      hipLaunchKernelGGL((patchify_kernel<float, false>), dim3(grid), dim3(256), 0, s, (const float*)image, col, B, R, P, G, Kpad, total);
  } else if (image_dtype == CLIPMI_F16) {
    if (vec)
      hipLaunchKernelGGL((patchify_kernel<half_t, true>), dim3(grid), dim3(256), 0, s, (const half_t*)image, col, B, R, P, G, Kpad, total);
    else
      hipLaunchKernelGGL((patchify_kernel<half_t, false>), dim3(grid), dim3(256), 0, s, (const half_t*)image, col, B, R, P, G, Kpad, total);
  } else {
    set_error("patchify: bad image dtype %d", image_dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("patchify_kernel");
}

int launch_cls_and_ctx_rows(float* x0, const float* cls, const float* pos, const float* shallow, int B, int tokens0,
                            int n_ctx, int D, hipStream_t s) {
  CLIPMI_REQUIRE(x0 && cls && pos && (n_ctx == 0 || shallow), CLIPMI_ERR_ARG, "cls rows: null pointer");
  if (B == 0) return CLIPMI_OK;
  hipLaunchKernelGGL(cls_ctx_kernel, dim3(B * (1 + n_ctx)), dim3(256), 0, s, x0, cls, pos, shallow, tokens0, n_ctx, D);
  return check_launch("cls_ctx_kernel");
}

int launch_overwrite_tokens(float* x, const float* prompt, int N, int L, int D, int first, int n_ctx, hipStream_t s) {
  CLIPMI_REQUIRE(x && prompt, CLIPMI_ERR_ARG, "overwrite tokens: null pointer");
  CLIPMI_REQUIRE(first >= 0 && n_ctx > 0 && first + n_ctx <= L, CLIPMI_ERR_SHAPE, "overwrite tokens: [%d,%d) outside L=%d", first,
                 first + n_ctx, L);
  if (N == 0) return CLIPMI_OK;
  hipLaunchKernelGGL(overwrite_kernel, dim3(N * n_ctx), dim3(256), 0, s, x, prompt, L, D, first, n_ctx);
  return check_launch("overwrite_kernel");
}

int launch_row_stats(const float* x, half_t* x16, float* stats, int parts, int N, int L, int D, int first, int n_ctx, hipStream_t s) {
  CLIPMI_REQUIRE(x && x16 && stats, CLIPMI_ERR_ARG, "row_stats: null pointer");
  CLIPMI_REQUIRE(parts >= 1 && parts <= LN_MAX_PARTS, CLIPMI_ERR_ARG, "row_stats: parts=%d", parts);
  CLIPMI_REQUIRE(D % 4 == 0 && first >= 0 && n_ctx > 0 && first + n_ctx <= L, CLIPMI_ERR_SHAPE, "row_stats: bad shape");
  const int total = N * n_ctx;
  if (total == 0) return CLIPMI_OK;
  hipLaunchKernelGGL(row_stats_kernel, dim3((total + 3) / 4), dim3(256), 0, s, x, x16, stats, parts, (int64_t)N * L, L, D, first, n_ctx, total);
  return check_launch("row_stats_kernel");
}

int launch_add_pos(const void* src, int dtype, const float* pos, float* xres, int C, int L, int src_L, int D, hipStream_t s) {
  CLIPMI_REQUIRE(src && xres, CLIPMI_ERR_ARG, "add_pos: null pointer");
  CLIPMI_REQUIRE(D % 4 == 0, CLIPMI_ERR_SHAPE, "add_pos: D=%d must be a multiple of 4", D);
  CLIPMI_REQUIRE(L > 0 && L <= src_L, CLIPMI_ERR_SHAPE, "add_pos: %d rows of a %d-row sequence", L, src_L);
  const int64_t total4 = (int64_t)C * L * (D / 4);
  if (total4 == 0) return CLIPMI_OK;
  const unsigned grid = (unsigned)((total4 + 255) / 256);
  if (dtype == CLIPMI_F32)
    hipLaunchKernelGGL(add_pos_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, pos, xres, L, src_L, D / 4, total4);
  else if (dtype == CLIPMI_F16)
    hipLaunchKernelGGL(add_pos_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)src, pos, xres, L, src_L, D / 4, total4);
  else {
    set_error("add_pos: bad dtype %d", dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("add_pos_kernel");
}

int launch_embed_tokens(const int64_t* ids, const float* table, const float* pos, float* xres, int32_t* eot, int C, int L, int src_L,
                        int D, int vocab, hipStream_t s) {
  CLIPMI_REQUIRE(ids && table && pos && xres, CLIPMI_ERR_ARG, "embed: null pointer");
  CLIPMI_REQUIRE(D % 4 == 0, CLIPMI_ERR_SHAPE, "embed: D=%d must be a multiple of 4", D);
  CLIPMI_REQUIRE(L > 0 && L <= src_L, CLIPMI_ERR_SHAPE, "embed: %d rows of a %d-token prompt", L, src_L);
  if (C == 0) return CLIPMI_OK;
  const int64_t total4 = (int64_t)C * L * (D / 4);
  hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, ids, table, pos, xres, L, src_L, D / 4,
                     vocab, total4);
  int rc = check_launch("embed_kernel");
  if (rc != CLIPMI_OK) return rc;
  if (eot) {   // the argmax scans the whole prompt
    hipLaunchKernelGGL(eot_kernel, dim3(C), dim3(64), 0, s, ids, eot, (int32_t*)nullptr, C, src_L);
    rc = check_launch("eot_kernel");
  }
  return rc;
}

int launch_eot_rows(const int32_t* eot, int32_t* rows, int C, int L, hipStream_t s) {
  CLIPMI_REQUIRE(eot && rows, CLIPMI_ERR_ARG, "eot rows: null pointer");
  if (C == 0) return CLIPMI_OK;
  hipLaunchKernelGGL(rows_from_eot_kernel, dim3((C + 255) / 256), dim3(256), 0, s, eot, rows, C, L);
  return check_launch("rows_from_eot_kernel");
}

int launch_cast_f32(const float* src, void* dst, int dtype, int64_t n, hipStream_t s) {
  CLIPMI_REQUIRE(src && dst, CLIPMI_ERR_ARG, "cast: null pointer");
  if (n == 0) return CLIPMI_OK;
  const unsigned grid = (unsigned)((n / 4 + 256) / 256);
  if (dtype == CLIPMI_F32)
    hipLaunchKernelGGL(cast_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, n);
  else if (dtype == CLIPMI_F16)
    hipLaunchKernelGGL(cast_kernel<half_t>, dim3(grid), dim3(256), 0, s, src, (half_t*)dst, n);
  else {
    set_error("cast: bad dtype %d", dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("cast_kernel");
}

int launch_cast_f16(const half_t* src, void* dst, int dtype, int64_t n, hipStream_t s) {
  if (n == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(src && dst, CLIPMI_ERR_ARG, "cast16: null pointer");
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (dtype == CLIPMI_F32)
    hipLaunchKernelGGL(cast16_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, n);
  else if (dtype == CLIPMI_F16)
    hipLaunchKernelGGL(cast16_kernel<half_t>, dim3(grid), dim3(256), 0, s, src, (half_t*)dst, n);
  else {
    set_error("cast16: bad dtype %d", dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("cast16_kernel");
}

int launch_rows_out(const void* src, int src_dtype, void* dst, int dst_dtype, int C, int rows, int L, int D, hipStream_t s) {
  if (C == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(src && dst, CLIPMI_ERR_ARG, "rows_out: null pointer");
  CLIPMI_REQUIRE(D % 4 == 0 && rows > 0 && rows <= L, CLIPMI_ERR_SHAPE, "rows_out: bad shape");
  const int64_t total4 = (int64_t)C * L * (D / 4);
  const unsigned grid = (unsigned)((total4 + 255) / 256);
  const bool s32 = src_dtype == CLIPMI_F32, d32 = dst_dtype == CLIPMI_F32;
  CLIPMI_REQUIRE((s32 || src_dtype == CLIPMI_F16) && (d32 || dst_dtype == CLIPMI_F16), CLIPMI_ERR_ARG, "rows_out: bad dtype");
  if (s32 && d32) hipLaunchKernelGGL((rows_out_kernel<float, float>), dim3(grid), dim3(256), 0, s, (const float*)src, (float*)dst, rows, L, D / 4, total4);
  else if (s32) hipLaunchKernelGGL((rows_out_kernel<float, half_t>), dim3(grid), dim3(256), 0, s, (const float*)src, (half_t*)dst, rows, L, D / 4, total4);
  else if (d32) hipLaunchKernelGGL((rows_out_kernel<half_t, float>), dim3(grid), dim3(256), 0, s, (const half_t*)src, (float*)dst, rows, L, D / 4, total4);
  else hipLaunchKernelGGL((rows_out_kernel<half_t, half_t>), dim3(grid), dim3(256), 0, s, (const half_t*)src, (half_t*)dst, rows, L, D / 4, total4);
  return check_launch("rows_out_kernel");
}

int launch_l2_normalize(const void* in, int in_dtype, float* out, int rows, int E, hipStream_t s) {
  return launch_l2_normalize_to(in, in_dtype, out, CLIPMI_F32, rows, E, s);
}

int launch_l2_normalize_to(const void* in, int in_dtype, void* out_, int out_dtype, int rows, int E, hipStream_t s) {
  if (rows == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(in && out_, CLIPMI_ERR_ARG, "l2_normalize: null pointer");
  CLIPMI_REQUIRE(rows > 0 && E > 0, CLIPMI_ERR_SHAPE, "l2_normalize: bad shape");
  const dim3 grid((rows + 3) / 4);
  const int vec8 = (E % 8 == 0 && (uintptr_t)in % 16 == 0) ? 1 : 0;   // 8-element chunks (see l2norm_kernel: the summation order)
  if (out_dtype == CLIPMI_F16) {   // the exchange format of the multi-GPU path (fp16 embeddings over xGMI)
    half_t* o16 = static_cast<half_t*>(out_);
    if (in_dtype == CLIPMI_F32)
      hipLaunchKernelGGL((l2norm_kernel<float, half_t>), grid, dim3(256), 0, s, (const float*)in, o16, rows, E, vec8);
    else if (in_dtype == CLIPMI_F16)
      hipLaunchKernelGGL((l2norm_kernel<half_t, half_t>), grid, dim3(256), 0, s, (const half_t*)in, o16, rows, E, vec8);
    else {
      set_error("l2_normalize: bad dtype %d", in_dtype);
      return CLIPMI_ERR_ARG;
    }
    return check_launch("l2norm_kernel");
  }
  CLIPMI_REQUIRE(out_dtype == CLIPMI_F32, CLIPMI_ERR_ARG, "l2_normalize: bad output dtype %d", out_dtype);
  float* out = static_cast<float*>(out_);
  if (in_dtype == CLIPMI_F32)
    hipLaunchKernelGGL(l2norm_kernel<float>, grid, dim3(256), 0, s, (const float*)in, out, rows, E, vec8);
  else if (in_dtype == CLIPMI_F16)
    hipLaunchKernelGGL(l2norm_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)in, out, rows, E, vec8);
  else {
    set_error("l2_normalize: bad dtype %d", in_dtype);
    return CLIPMI_ERR_ARG;
  }
  return check_launch("l2norm_kernel");
}

}  // namespace clipmi
