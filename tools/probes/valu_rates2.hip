// Probe (not part of the library; round 6): per-wave issue cost of the instructions the GEMM epilogues are made of, at the occupancy the
// persistent GEMM runs them (2 waves per SIMD = 8 waves per CU) and at 4 waves per SIMD.  Ticks = s_memtime around 4000 iterations of 8
// independent instructions (or of one epilogue group), per wave-instruction as ONE wave sees it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates2 tools/probes/valu_rates2.hip && /tmp/valu_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define R8(op, tail) op " %0, %0" tail "\n" op " %1, %1" tail "\n" op " %2, %2" tail "\n" op " %3, %3" tail "\n" op " %4, %4" tail "\n" op " %5, %5" tail "\n" op " %6, %6" tail "\n" op " %7, %7" tail "\n"

#define CASE_KERNEL(NAME, BODY)                                                                                                   \
  __global__ __launch_bounds__(1024) void NAME(long long* out, int iters) {                                                       \
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;     \
    float c = 0.999f, d = 1e-6f;                                                                                                  \
    int s0 = 0;                                                                                                                   \
    double p0 = a0, p1 = a1, p2 = a2, p3 = a3, q = 0.5;                                                                           \
    asm volatile("" : "+v"(c), "+v"(d));                                                                                          \
    __syncthreads();                                                                                                              \
    const long long t0 = __builtin_readcyclecounter();                                                                            \
    for (int i = 0; i < iters; ++i) {                                                                                             \
      asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), [s0] "+s"(s0),         \
                   [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3) : [c] "v"(c), [d] "v"(d), [q] "v"(q));                \
    }                                                                                                                             \
    const long long t1 = __builtin_readcyclecounter();                                                                            \
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)s0 + (float)(p0 + p1 + p2 + p3);                                                             \
    if (sink == 123.456f) out[4096] = 1;                                                                                          \
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0;                                              \
  }

// operands: %0..%7 = a0..a7 (VGPR, in/out), %[s0] an SGPR, %[c] / %[d] constants in VGPRs, %[p0]..%[p3] 64-bit register pairs, %[q] a constant pair
CASE_KERNEL(k_fma, R8("v_fma_f32", ", %[c], %[d]"))
CASE_KERNEL(k_mul, R8("v_mul_f32", ", %[c]"))
CASE_KERNEL(k_mixlo, R8("v_fma_mixlo_f16", ", %[c], %[d]"))
CASE_KERNEL(k_mixhi, R8("v_fma_mixhi_f16", ", %[c], %[d]"))
CASE_KERNEL(k_mix32, R8("v_fma_mix_f32", ", %[c], %[d] op_sel_hi:[1,0,0]"))
CASE_KERNEL(k_cvt16, R8("v_cvt_f16_f32", ""))
CASE_KERNEL(k_cvtpk, R8("v_cvt_pk_f16_f32", ", %[c]"))
CASE_KERNEL(k_pkmul16, R8("v_pk_mul_f16", ", %[c]"))
CASE_KERNEL(k_pkfma16, R8("v_pk_fma_f16", ", %[c], %[d]"))
CASE_KERNEL(k_exp, R8("v_exp_f32", ""))
CASE_KERNEL(k_rcp, R8("v_rcp_f32", ""))
CASE_KERNEL(k_readlane, "v_readlane_b32 %[s0], %0, 3\n v_readlane_b32 %[s0], %1, 3\n v_readlane_b32 %[s0], %2, 3\n v_readlane_b32 %[s0], %3, 3\n"
                        "v_readlane_b32 %[s0], %4, 3\n v_readlane_b32 %[s0], %5, 3\n v_readlane_b32 %[s0], %6, 3\n v_readlane_b32 %[s0], %7, 3\n")
CASE_KERNEL(k_writelane, "v_writelane_b32 %0, %[s0], 3\n v_writelane_b32 %1, %[s0], 3\n v_writelane_b32 %2, %[s0], 3\n v_writelane_b32 %3, %[s0], 3\n"
                         "v_writelane_b32 %4, %[s0], 3\n v_writelane_b32 %5, %[s0], 3\n v_writelane_b32 %6, %[s0], 3\n v_writelane_b32 %7, %[s0], 3\n")
CASE_KERNEL(k_add64, "v_add_f64 %[p0], %[p0], %[p0]\n v_add_f64 %[p1], %[p1], %[p1]\n v_add_f64 %[p2], %[p2], %[p2]\n v_add_f64 %[p3], %[p3], %[p3]\n"
                     "v_add_f64 %[p0], %[p0], %[p0]\n v_add_f64 %[p1], %[p1], %[p1]\n v_add_f64 %[p2], %[p2], %[p2]\n v_add_f64 %[p3], %[p3], %[p3]\n")
CASE_KERNEL(k_swap, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                    "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
// the QuickGELU epilogue per FOUR elements as the product kernel issues it today (a0..a3 = accumulators): 2 pk_fma (c) + 2 pk_fma (v) + 4 mul (t)
// + 4 exp + 4 add + 4 rcp + 2 pk_mul + 2 cvt_pk = 24 instructions
CASE_KERNEL(k_gelu_now,
            "v_pk_fma_f32 %[p2], %[p2], %[q], %[q]\n v_pk_fma_f32 %[p3], %[p3], %[q], %[q]\n"
            "v_pk_fma_f32 %[p0], %[p0], %[q], %[p2]\n v_pk_fma_f32 %[p1], %[p1], %[q], %[p3]\n"
            "v_mul_f32 %4, %0, %[c]\n v_mul_f32 %5, %1, %[c]\n v_mul_f32 %6, %2, %[c]\n v_mul_f32 %7, %3, %[c]\n"
            "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
            "v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_pk_mul_f32 %[p0], %[p0], %[p2]\n v_pk_mul_f32 %[p1], %[p1], %[p3]\n"
            "v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %2, %2, %3\n")
// proposed: z = log2e*1.702*v straight from the accumulator (scaled row / column terms), e = exp2(-z), d = k + k e, r = rcp(d), out = mixlo/hi(z r):
// 2 pk_fma (c) + 2 pk_fma (z) + 4 exp + 2 pk_fma (d) + 4 rcp + 4 mix = 18 instructions
CASE_KERNEL(k_gelu_new,
            "v_pk_fma_f32 %[p2], %[p2], %[q], %[q]\n v_pk_fma_f32 %[p3], %[p3], %[q], %[q]\n"
            "v_pk_fma_f32 %[p0], %[p0], %[q], %[p2]\n v_pk_fma_f32 %[p1], %[p1], %[q], %[p3]\n"
            "v_exp_f32 %4, -%0\n v_exp_f32 %5, -%1\n v_exp_f32 %6, -%2\n v_exp_f32 %7, -%3\n"
            "v_pk_fma_f32 %[p2], %[p2], %[q], %[q]\n v_pk_fma_f32 %[p3], %[p3], %[q], %[q]\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_fma_mixlo_f16 %0, %0, %4, 0\n v_fma_mixhi_f16 %0, %1, %5, 0\n v_fma_mixlo_f16 %2, %2, %6, 0\n v_fma_mixhi_f16 %2, %3, %7, 0\n")
// the same with scalar fma instead of the packed ones (6 + 4 + 4 + 4 + 4 = 22 instructions)
CASE_KERNEL(k_gelu_new_scalar,
            "v_fma_f32 %4, %4, %[c], %[d]\n v_fma_f32 %5, %5, %[c], %[d]\n v_fma_f32 %6, %6, %[c], %[d]\n v_fma_f32 %7, %7, %[c], %[d]\n"
            "v_fma_f32 %0, %0, %[c], %4\n v_fma_f32 %1, %1, %[c], %5\n v_fma_f32 %2, %2, %[c], %6\n v_fma_f32 %3, %3, %[c], %7\n"
            "v_exp_f32 %4, -%0\n v_exp_f32 %5, -%1\n v_exp_f32 %6, -%2\n v_exp_f32 %7, -%3\n"
            "v_fma_f32 %4, %4, %[c], %[c]\n v_fma_f32 %5, %5, %[c], %[c]\n v_fma_f32 %6, %6, %[c], %[c]\n v_fma_f32 %7, %7, %[c], %[c]\n"
            "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
            "v_fma_mixlo_f16 %0, %0, %4, 0\n v_fma_mixhi_f16 %0, %1, %5, 0\n v_fma_mixlo_f16 %2, %2, %6, 0\n v_fma_mixhi_f16 %2, %3, %7, 0\n")
// bias-only epilogue (in-proj) per four elements: today 2 pk_fma (c) + 2 pk_fma (v) + 2 cvt_pk; proposed 2 pk_fma (c) + 4 mix
CASE_KERNEL(k_bias_now,
            "v_pk_fma_f32 %[p2], %[p2], %[q], %[q]\n v_pk_fma_f32 %[p3], %[p3], %[q], %[q]\n"
            "v_pk_fma_f32 %[p0], %[p0], %[q], %[p2]\n v_pk_fma_f32 %[p1], %[p1], %[q], %[p3]\n"
            "v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %2, %2, %3\n")
CASE_KERNEL(k_bias_new,
            "v_pk_fma_f32 %[p2], %[p2], %[q], %[q]\n v_pk_fma_f32 %[p3], %[p3], %[q], %[q]\n"
            "v_fma_mixlo_f16 %0, %0, %[c], %4\n v_fma_mixhi_f16 %0, %1, %[c], %5\n v_fma_mixlo_f16 %2, %2, %[c], %6\n v_fma_mixhi_f16 %2, %3, %[c], %7\n")

template <typename K>
int run(const char* what, K kern, int per_iter, int elems, long long* dev, int waves) {
  const int iters = 4000;
  kern<<<256, waves * 64>>>(dev, iters);
  CHECK(hipDeviceSynchronize());
  kern<<<256, waves * 64>>>(dev, iters);
  CHECK(hipDeviceSynchronize());
  std::vector<long long> h(32);
  CHECK(hipMemcpy(h.data(), dev, 32 * 8, hipMemcpyDeviceToHost));
  long long t = 0;
  for (int w = 0; w < waves; ++w) t = h[w] > t ? h[w] : t;
  printf("%-34s waves/SIMD %d: %7.3f ticks per wave instruction", what, waves / 4, (double)t / iters / per_iter);
  if (elems) printf("   = %7.2f ticks per ELEMENT and wave (%d instructions per %d elements)", (double)t / iters / elems, per_iter, elems);
  printf("\n");
  return 0;
}

int main() {
  long long* dev;
  CHECK(hipMalloc(&dev, 8 * 8192));
  CHECK(hipMemset(dev, 0, 8 * 8192));
  for (int waves : {8, 16}) {
    run("v_fma_f32", k_fma, 8, 0, dev, waves);
    run("v_mul_f32", k_mul, 8, 0, dev, waves);
    run("v_fma_mixlo_f16", k_mixlo, 8, 0, dev, waves);
    run("v_fma_mixhi_f16", k_mixhi, 8, 0, dev, waves);
    run("v_fma_mix_f32 (f16 source)", k_mix32, 8, 0, dev, waves);
    run("v_cvt_f16_f32", k_cvt16, 8, 0, dev, waves);
    run("v_cvt_pk_f16_f32", k_cvtpk, 8, 0, dev, waves);
    run("v_pk_mul_f16", k_pkmul16, 8, 0, dev, waves);
    run("v_pk_fma_f16", k_pkfma16, 8, 0, dev, waves);
    run("v_exp_f32", k_exp, 8, 0, dev, waves);
    run("v_rcp_f32", k_rcp, 8, 0, dev, waves);
    run("v_readlane_b32", k_readlane, 8, 0, dev, waves);
    run("v_writelane_b32", k_writelane, 8, 0, dev, waves);
    run("v_add_f64", k_add64, 8, 0, dev, waves);
    run("v_permlane16_swap_b32", k_swap, 8, 0, dev, waves);
    run("QuickGELU epilogue today", k_gelu_now, 24, 4, dev, waves);
    run("QuickGELU epilogue proposed", k_gelu_new, 18, 4, dev, waves);
    run("QuickGELU proposed, scalar fma", k_gelu_new_scalar, 24, 4, dev, waves);
    run("bias epilogue today", k_bias_now, 6, 4, dev, waves);
    run("bias epilogue proposed", k_bias_new, 6, 4, dev, waves);
  }
  return 0;
}
