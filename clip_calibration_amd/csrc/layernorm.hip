// LayerNorm with fp32 statistics (reference clip/model.py:153-159): one wave per row, the row lives in
// registers (<= 16 float4 per lane), two-pass mean/variance, 16-byte vector loads/stores.  HBM-bound:
// algorithmic bytes per row = D * (sizeof(in) + sizeof(out)).
#include "common.h"

namespace clipmi {
namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<half_t> {
  static __device__ __forceinline__ f32x4 load(const half_t* p) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  static __device__ __forceinline__ void store(half_t* p, f32x4 v) {
    *reinterpret_cast<f16x4*>(p) = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
  }
};

// (the xor butterfly the kernel has summed with since round 1: the row statistics keep their bits; common.h's DPP tree adds in another order)
__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename TI, typename TO, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, int64_t in_stride,
                                                        const int32_t* __restrict__ gather, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, TO* __restrict__ y,
                                                        int64_t out_stride, int rows, int D, float eps,
                                                        half_t* __restrict__ y16, float* __restrict__ stats_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t src_row = gather ? (int64_t)gather[row] : (int64_t)row;
  const TI* xr = x + src_row * in_stride;
  const int nvec = D >> 2;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      v[i] = Vec4<TI>::load(xr + c * 4);
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = ln_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(ln_wave_sum(q) / (float)D + eps);
  TO* yr = y + (int64_t)row * out_stride;
  float os = 0.f, oq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      if (y) Vec4<TO>::store(yr + c * 4, o);   // y == nullptr: only the fp16 copy and the row sums are wanted (fp16 residual stream)
      if (y16) {   // producer side of the LayerNorm fold (gemm.hip): fp16 copy + row sums of the OUTPUT
        Vec4<half_t>::store(y16 + (int64_t)row * out_stride + c * 4, o);
        os += (o[0] + o[1]) + (o[2] + o[3]);
        oq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
      }
    }
  }
  if (stats_out) {
    os = ln_wave_sum(os);
    oq = ln_wave_sum(oq);
    if (lane == 0) {
      *reinterpret_cast<float2*>(stats_out + 2 * (int64_t)row) = make_float2(os, oq);   // partial 0 (the only one)
    }
  }
}

template <typename TI, typename TO>
int dispatch(const void* x, int64_t in_stride, const int32_t* gather, const float* gamma, const float* beta, void* y,
             int64_t out_stride, int rows, int D, float eps, hipStream_t s, half_t* y16, float* stats_out) {
  const dim3 grid((rows + 3) / 4), block(256);
  const int nvec = D / 4;
  if (nvec <= 64 * 4) {
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, 4>), grid, block, 0, s, (const TI*)x, in_stride, gather, gamma, beta,
                       (TO*)y, out_stride, rows, D, eps, y16, stats_out);
  } else {
    hipLaunchKernelGGL((layernorm_kernel<TI, TO, 16>), grid, block, 0, s, (const TI*)x, in_stride, gather, gamma, beta,
                       (TO*)y, out_stride, rows, D, eps, y16, stats_out);
  }
  return check_launch("layernorm_kernel");
}

}  // namespace

int launch_layernorm(const void* x, int x_dtype, int64_t in_stride, const int32_t* gather_idx, const float* gamma,
                     const float* beta, void* y, int y_dtype, int64_t out_stride, int rows, int D, float eps,
                     hipStream_t s, half_t* y16, float* stats_out) {
  if (rows == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x && gamma && beta && (y || y16), CLIPMI_ERR_ARG, "layernorm: null pointer");
  CLIPMI_REQUIRE((!y16 && !stats_out) || (y16 && stats_out), CLIPMI_ERR_ARG, "layernorm: y16 and stats_out come together");
  CLIPMI_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && D <= 4096, CLIPMI_ERR_SHAPE,
                 "layernorm: rows=%d D=%d unsupported (D %% 4 == 0, D <= 4096)", rows, D);
  CLIPMI_REQUIRE(in_stride % 4 == 0 && out_stride % 4 == 0 && in_stride >= D && out_stride >= D, CLIPMI_ERR_SHAPE,
                 "layernorm: strides must be multiples of 4 and >= D");
  CLIPMI_REQUIRE((uintptr_t)x % 8 == 0 && (uintptr_t)y % 8 == 0 && (uintptr_t)gamma % 16 == 0 && (uintptr_t)beta % 16 == 0,
                 CLIPMI_ERR_ARG, "layernorm: unaligned pointer");
  const bool xi32 = x_dtype == CLIPMI_F32, yo32 = y_dtype == CLIPMI_F32;
  CLIPMI_REQUIRE((xi32 || x_dtype == CLIPMI_F16) && (yo32 || y_dtype == CLIPMI_F16), CLIPMI_ERR_ARG, "layernorm: bad dtype");
  if (xi32 && yo32) return dispatch<float, float>(x, in_stride, gather_idx, gamma, beta, y, out_stride, rows, D, eps, s, y16, stats_out);
  if (xi32 && !yo32) return dispatch<float, half_t>(x, in_stride, gather_idx, gamma, beta, y, out_stride, rows, D, eps, s, y16, stats_out);
  if (!xi32 && yo32) return dispatch<half_t, float>(x, in_stride, gather_idx, gamma, beta, y, out_stride, rows, D, eps, s, y16, stats_out);
  return dispatch<half_t, half_t>(x, in_stride, gather_idx, gamma, beta, y, out_stride, rows, D, eps, s, y16, stats_out);
}

}  // namespace clipmi
