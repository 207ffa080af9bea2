// Probe (not part of the library): issue rates of the vector pipe on gfx950, per wave64 instruction, measured with s_memtime on one
// workgroup per CU.  Cases: v_exp_f32 alone, v_fma_f32 alone, v_pk_fma_f32 alone, exp + fma interleaved 1:1 (do the transcendental
// unit and the main pipe overlap inside ONE wave?), the same from 2 / 3 / 4 waves of a SIMD, exp beside MFMAs of a partner wave.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/probes/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(1024) void rate_kernel(long long* out, int iters, int mfma_waves, int indep) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
  const float c = 0.999f, d = 1e-6f;
  const int wave = threadIdx.x >> 6;
  f32x16 acc = {0};
  f16x8 fa = {1, 1, 1, 1, 1, 1, 1, 1}, fb = fa;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  f32x16 acc2 = {0};
  if (wave < mfma_waves) {   // partner waves: back-to-back MFMAs on their SIMD (waves w and w+4 share a SIMD)
    if (indep == 2) {   // accumulator in AccVGPRs
      for (int i = 0; i < iters; ++i) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n"
                     "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n v_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
                     : "+a"(acc) : "v"(fa), "v"(fb));
      }
    } else if (indep) {
      for (int i = 0; i < iters; ++i) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc2, 0, 0, 0);
      }
    } else {
      for (int i = 0; i < iters; ++i) {
        REP8(acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);)
      }
    }
  } else {
    for (int i = 0; i < iters; ++i) {
      if constexpr (MODE == 0) {   // 8 exp
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if constexpr (MODE == 1) {   // 8 fma
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if constexpr (MODE == 2) {   // 8 exp + 8 fma interleaved, independent registers
        asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %8, %8, %16, %17\n v_exp_f32 %1, %1\n v_fma_f32 %9, %9, %16, %17\n"
                     "v_exp_f32 %2, %2\n v_fma_f32 %10, %10, %16, %17\n v_exp_f32 %3, %3\n v_fma_f32 %11, %11, %16, %17\n"
                     "v_exp_f32 %4, %4\n v_fma_f32 %12, %12, %16, %17\n v_exp_f32 %5, %5\n v_fma_f32 %13, %13, %16, %17\n"
                     "v_exp_f32 %6, %6\n v_fma_f32 %14, %14, %16, %17\n v_exp_f32 %7, %7\n v_fma_f32 %15, %15, %16, %17"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                       "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                     : "v"(c), "v"(d));
      } else if constexpr (MODE == 3) {   // 8 exp + 24 fma (1:3)
        asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n"
                     "v_exp_f32 %1, %1\n v_fma_f32 %11, %11, %16, %17\n v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n"
                     "v_exp_f32 %2, %2\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n v_fma_f32 %8, %8, %16, %17\n"
                     "v_exp_f32 %3, %3\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                     "v_exp_f32 %4, %4\n v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n"
                     "v_exp_f32 %5, %5\n v_fma_f32 %15, %15, %16, %17\n v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n"
                     "v_exp_f32 %6, %6\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n v_fma_f32 %12, %12, %16, %17\n"
                     "v_exp_f32 %7, %7\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                       "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                     : "v"(c), "v"(d));
      } else if constexpr (MODE == 4) {   // 8 v_pk_fma_f32 (16 fma)
        asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                     "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                     : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0), "v"(*(double*)&b2));
      } else if constexpr (MODE == 5) {   // 8 v_max3_f32
        asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n"
                     "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if constexpr (MODE == 6) {   // 8 v_cvt_pk_f16_f32 (16 conversions)
        asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %4\n"
                     "v_cvt_pk_f16_f32 %4, %4, %5\n v_cvt_pk_f16_f32 %5, %5, %6\n v_cvt_pk_f16_f32 %6, %6, %7\n v_cvt_pk_f16_f32 %7, %7, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if constexpr (MODE == 7) {   // 8 v_rcp_f32
        asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else if constexpr (MODE == 10 || MODE == 11 || MODE == 12) {   // ONE wave: 8 x [1 MFMA + 5 v_fma] (10) / [1 MFMA + 3 v_exp] (11) / [1 MFMA + 7 v_fma] (12)
#define MF "v_mfma_f32_32x32x16_f16 %16, %17, %18, %16\n"
#define F5(a) "v_fma_f32 %" #a ", %" #a ", %19, %20\n v_fma_f32 %" #a ", %" #a ", %19, %20\n v_fma_f32 %" #a ", %" #a ", %19, %20\n v_fma_f32 %" #a ", %" #a ", %19, %20\n v_fma_f32 %" #a ", %" #a ", %19, %20\n"
#define F2(a) "v_fma_f32 %" #a ", %" #a ", %19, %20\n v_fma_f32 %" #a ", %" #a ", %19, %20\n"
#define E3(a, b, c) "v_exp_f32 %" #a ", %" #a "\n v_exp_f32 %" #b ", %" #b "\n v_exp_f32 %" #c ", %" #c "\n"
        if constexpr (MODE == 10)
          asm volatile(MF F5(0) MF F5(1) MF F5(2) MF F5(3) MF F5(4) MF F5(5) MF F5(6) MF F5(7)
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                         "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7), "+v"(acc)
                       : "v"(fa), "v"(fb), "v"(c), "v"(d));
        else if constexpr (MODE == 12)
          asm volatile(MF F5(0) F2(8) MF F5(1) F2(9) MF F5(2) F2(10) MF F5(3) F2(11) MF F5(4) F2(12) MF F5(5) F2(13) MF F5(6) F2(14) MF F5(7) F2(15)
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                         "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7), "+v"(acc)
                       : "v"(fa), "v"(fb), "v"(c), "v"(d));
        else
          asm volatile(MF E3(0, 1, 2) MF E3(3, 4, 5) MF E3(6, 7, 8) MF E3(9, 10, 11) MF E3(12, 13, 14) MF E3(15, 0, 1) MF E3(2, 3, 4) MF E3(5, 6, 7)
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                         "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7), "+v"(acc)
                       : "v"(fa), "v"(fb), "v"(c), "v"(d));
      } else if constexpr (MODE == 8) {   // 8 v_exp_f16
        asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 + acc[0] + acc[5] + acc2[3];
  if (sink == 123.456f) out[4096] = 1;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    out[wave] = t1 - t0;
    out[32 + wave] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));   // HW_ID bits 0-15: wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13
  }
}

template <int MODE>
int run(const char* what, int per_iter, long long* dev, int waves, int mfma_waves, int indep = 0) {
  const int iters = 4000;
  rate_kernel<MODE><<<256, waves * 64>>>(dev, iters, mfma_waves, indep);
  CHECK(hipDeviceSynchronize());
  rate_kernel<MODE><<<256, waves * 64>>>(dev, iters, mfma_waves, indep);
  CHECK(hipDeviceSynchronize());
  std::vector<long long> h(64);
  CHECK(hipMemcpy(h.data(), dev, 64 * 8, hipMemcpyDeviceToHost));
  static int shown = 0;
  if (shown != waves) {
    shown = waves;
    printf("   SIMD of waves 0..%d of workgroup 0:", waves - 1);
    for (int w = 0; w < waves; ++w) printf(" %lld", (h[32 + w] >> 4) & 3);
    printf("   (CU %lld)\n", (h[32] >> 8) & 15);
  }
  // readcyclecounter = s_memtime: 100 MHz-class constant clock on gfx9?  report raw ticks AND ticks relative to the fma case
  long long valu = 0;
  for (int w = mfma_waves; w < waves; ++w) valu = h[w] > valu ? h[w] : valu;
  printf("%-44s waves %2d (mfma %d): %8lld ticks  = %7.3f ticks per wave instruction", what, waves, mfma_waves, valu, (double)valu / iters / per_iter);
  if (mfma_waves) printf("   | mfma wave 0: %8lld ticks = %7.3f per MFMA", h[0], (double)h[0] / iters / 8);
  printf("\n");
  return 0;
}

int main() {
  long long* dev;
  CHECK(hipMalloc(&dev, 8 * 8192));
  CHECK(hipMemset(dev, 0, 8 * 8192));
  for (int waves : {4, 8, 12, 16}) {
    run<1>("v_fma_f32", 8, dev, waves, 0);
    run<4>("v_pk_fma_f32", 8, dev, waves, 0);
    run<5>("v_max3_f32", 8, dev, waves, 0);
    run<6>("v_cvt_pk_f16_f32", 8, dev, waves, 0);
    run<0>("v_exp_f32", 8, dev, waves, 0);
    run<8>("v_exp_f16", 8, dev, waves, 0);
    run<7>("v_rcp_f32", 8, dev, waves, 0);
    run<2>("exp + fma 1:1 (per pair)", 8, dev, waves, 0);
    run<3>("exp + 3 fma (per group of 4)", 8, dev, waves, 0);
  }
  // beside MFMAs: waves 0..3 (one per SIMD) issue MFMAs, waves 4..7 the vector work on the same SIMDs
  run<1>("v_fma_f32 beside MFMA partner", 8, dev, 8, 4);
  run<0>("v_exp_f32 beside MFMA partner", 8, dev, 8, 4);
  run<2>("exp + fma 1:1 beside MFMA partner", 8, dev, 8, 4);
  run<1>("MFMA alone (fma waves idle: iters same)", 8, dev, 4, 4);
  // one MFMA wave + two / three vector waves per SIMD: what is left of the vector pipe's 4 cycles per instruction beside a busy matrix pipe?
  for (int waves : {12, 16}) {
    run<1>("v_fma_f32 beside MFMA partner", 8, dev, waves, 4);
    run<0>("v_exp_f32 beside MFMA partner", 8, dev, waves, 4);
    run<2>("exp + fma 1:1 beside MFMA partner", 8, dev, waves, 4);
    run<5>("v_max3_f32 beside MFMA partner", 8, dev, waves, 4);
    run<6>("v_cvt_pk_f16_f32 beside MFMA partner", 8, dev, waves, 4);
  }
  // two MFMA waves + two vector waves per SIMD
  run<1>("MFMA alone, two INDEPENDENT accumulators", 8, dev, 4, 4, 1);
  run<1>("2 MFMA waves per SIMD alone (dependent chain)", 8, dev, 8, 8);
  run<1>("2 MFMA waves per SIMD alone (independent)", 8, dev, 8, 8, 1);
  run<1>("v_fma_f32 beside MFMA partner (independent acc)", 8, dev, 12, 4, 1);
  run<2>("exp + fma 1:1 beside MFMA partner (indep. acc)", 8, dev, 12, 4, 1);
  run<1>("v_fma_f32 beside MFMA partner (AccVGPR acc)", 8, dev, 12, 4, 2);
  run<0>("v_exp_f32 beside MFMA partner (AccVGPR acc)", 8, dev, 12, 4, 2);
  run<2>("exp + fma 1:1 beside MFMA partner (AccVGPR)", 8, dev, 12, 4, 2);
  run<6>("v_cvt_pk_f16_f32 beside MFMA (AccVGPR acc)", 8, dev, 12, 4, 2);
  // ONE instruction stream that interleaves MFMAs with independent vector instructions (per group of 1 MFMA + n vector instructions; 32 cycles = the MFMA alone)
  for (int waves : {4, 8, 12}) {
    run<10>("same wave: [MFMA + 5 v_fma] per group", 8, dev, waves, 0);
    run<12>("same wave: [MFMA + 7 v_fma] per group", 8, dev, waves, 0);
    run<11>("same wave: [MFMA + 3 v_exp] per group", 8, dev, waves, 0);
  }
  run<1>("v_fma_f32 beside 2 MFMA waves per SIMD", 8, dev, 16, 8);
  run<2>("exp + fma 1:1 beside 2 MFMA waves per SIMD", 8, dev, 16, 8);
  return 0;
}
