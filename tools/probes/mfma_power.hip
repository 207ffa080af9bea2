// Power-cap probe: which MFMA shape gives more flop per joule?  Register-only loops (no LDS, no memory) on random fp16 operands,
// every CU busy with WAVES waves, long enough for the package power controller to settle.  Prints TFLOP/s and the in-kernel clock.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: 16x16x32 f16, 16 accumulator tiles (4x4 operands, the GEMM's wave tile).  MODE 1: 32x32x16 f16, 4 accumulator tiles (2x2 operands):
// the same 64x64 wave tile, the same flops per operand set, half the operand register reads per flop.
// NOPS: number of distinct operand sets rotated through (register-resident), so that operand values change between MFMAs like in a K loop.
// probe_lds<R>: the 16x16x32 loop of MODE 0 with R ds_read_b128 per 32 MFMAs refreshing the operand registers from LDS (same values:
// the LDS image is the operand set itself) -- the GEMM main loops read 12 per 32 MFMAs (8 waves of 128 x 64 per 256 x 256 tile).
template <int R>
__global__ __launch_bounds__(512) void probe_lds(const f16x8* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* clk) {
  __shared__ f16x8 img[16 * 512];
  const int lane = threadIdx.x;
  for (int i = 0; i < 16; ++i) img[i * 512 + lane] = src[(size_t)i * 512 + lane];
  __syncthreads();
  f16x8 a[2][4], b[2][4];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 4; ++i) {
      a[s][i] = img[(s * 8 + i) * 512 + lane];
      b[s][i] = img[(s * 8 + 4 + i) * 512 + lane];
    }
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc[4][4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][u], b[s][v], acc[u][v], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        // refresh operands of the OTHER set (used 16+ MFMAs from now): R / 8 reads after each group of four MFMAs
#pragma unroll
        for (int q = 0; q < R / 8; ++q) {
          const int idx = (u * (R / 8) + q) & 7;
          f16x8* dstp = idx < 4 ? &a[s ^ 1][idx] : &b[s ^ 1][idx - 4];
          const int slot = (s ^ 1) * 8 + idx;
          asm volatile("ds_read_b128 %0, %1" : "=v"(*dstp) : "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(img + slot * 512 + lane)));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  float r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  dst[blockIdx.x * 512 + lane] = r;
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && lane == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// bf16 operands (the same normal values rounded to bfloat16: 8 x 8-bit significand products instead of 11 x 11), MODE 0's issue order
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void probe_bf16(const bf16x8* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x;
  bf16x8 a[2][4], b[2][4];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 4; ++i) {
      a[s][i] = src[(size_t)(s * 8 + i) * 512 + lane];
      b[s][i] = src[(size_t)(s * 8 + 4 + i) * 512 + lane];
    }
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc[4][4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][u], b[s][v], acc[u][v], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
  }
  float r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  dst[blockIdx.x * 512 + lane] = r;
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && lane == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(const f16x8* __restrict__ src, float* __restrict__ dst, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x;
  f16x8 a[2][4], b[2][4];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 4; ++i) {
      a[s][i] = src[(size_t)(s * 8 + i) * 512 + lane];
      b[s][i] = src[(size_t)(s * 8 + 4 + i) * 512 + lane];
    }
  unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (MODE != 1) {
    // issue order of the 16 MFMAs of an operand set (operand 0 = "A" = srcA of the instruction, operand 1 = "B" = srcB):
    //   MODE 0: A held for four MFMAs, B changes every MFMA      MODE 2: B held, A changes every MFMA
    //   MODE 3: both change every MFMA (diagonal walk)            MODE 4: like 0 with two operand sets alternating every MFMA pair
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            constexpr int dummy = 0; (void)dummy;
            const int i = MODE == 0 ? u : MODE == 2 ? v : (u + v) & 3;
            const int j = MODE == 0 ? v : MODE == 2 ? u : v;
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    dst[blockIdx.x * 512 + lane] = r;
  } else {
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)        // two K = 16 halves of the same K = 32 slice: operands a[s][2*kk + i]
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][2 * kk + i], b[s][2 * kk + j], acc[i][j], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) r += acc[i][j][e];
    dst[blockIdx.x * 512 + lane] = r;
  }
  unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && lane == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? atoi(argv[1]) : 8;          // waves per workgroup (one workgroup per CU)
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const int zero = argc > 3 ? atoi(argv[3]) : 0;
  std::vector<_Float16> h(16 * 512 * 8);
  srand(1);
  for (auto& v : h) {   // Box-Muller normal, like the activations / weights of the bench
    float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
    v = zero ? (_Float16)0.f : (_Float16)(0.25f * sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
  }
  f16x8* src; float* dst; unsigned long long* clk;
  CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&dst, 256 * 512 * 4)); CK(hipMalloc(&clk, 16));
  CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  std::vector<unsigned short> hb(h.size());   // the same values as bfloat16 (round to nearest even on the fp32 bits)
  for (size_t i = 0; i < h.size(); ++i) {
    float f = (float)h[i];
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    hb[i] = (unsigned short)(u >> 16);
  }
  bf16x8* srcb; CK(hipMalloc(&srcb, hb.size() * 2));
  CK(hipMemcpy(srcb, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int mode) {
    switch (mode) {
      case 0: probe<0><<<256, waves * 64>>>(src, dst, iters, clk); break;
      case 1: probe<1><<<256, waves * 64>>>(src, dst, iters, clk); break;
      case 2: probe<2><<<256, waves * 64>>>(src, dst, iters, clk); break;
      case 3: probe<3><<<256, waves * 64>>>(src, dst, iters, clk); break;
      case 4: probe_lds<8><<<256, waves * 64>>>(src, dst, iters, clk); break;
      case 5: probe_lds<16><<<256, waves * 64>>>(src, dst, iters, clk); break;
      case 6: probe_lds<32><<<256, waves * 64>>>(src, dst, iters, clk); break;
      default: probe_bf16<<<256, waves * 64>>>(srcb, dst, iters, clk); break;
    }
  };
  static const char* names[] = {"16x16x32 A held x4", "32x32x16", "16x16x32 B held x4", "16x16x32 both change", "16x16x32 + 8 ds_read_b128 / 32 MFMA",
                                "16x16x32 + 16 ds_read_b128 / 32 MFMA", "16x16x32 + 32 ds_read_b128 / 32 MFMA", "16x16x32 bf16"};
  for (int mode = 0; mode < 8; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      // flops per wave per iteration: 2 operand sets x (64 x 64 x 32) MACs x 2
      const double flop = 2.0 * 2 * 64 * 64 * 32 * (double)iters * waves * 256;
      for (int l = 0; l < 6; ++l) {   // warm the power controller
        launch(mode);
      }
      CK(hipEventRecord(e0));
      const int L = 6;
      for (int l = 0; l < L; ++l) {
        launch(mode);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[2]; CK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
      printf("%s waves/CU %d %s: %.3f ms per launch, %.1f TFLOP/s, in-kernel clock %.0f MHz\n", names[mode], waves,
             zero ? "zeros" : "randn", ms / L, flop / (ms / L * 1e-3) / 1e12, (double)c[0] / ((double)c[1] / 100.0));
    }
  return 0;
}
