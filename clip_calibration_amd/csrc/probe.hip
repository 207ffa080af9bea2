// Ceiling probe (clipmi_probe_mfma_f16, include/clipmi.h): what the matrix pipe alone sustains on this chip, at the package power cap, on
// operands with the tower's statistics -- a register-only v_mfma_f32_16x16x32_f16 loop, no LDS and no memory inside it.  bench.py runs it
// for about a second beside its sysfs power sampler and reports TFLOP/s, W and clock in `ceiling.mfma_only`, so that the fractions of the
// 2.5 PFLOP/s datasheet peak in the same JSON line can be read against what the silicon does on toggling data.  Nothing on the product
// path calls this.  (The round-2 study with its shape / operand-order / LDS-read variants is tools/probes/mfma_power.hip.)
#include "common.h"

namespace clipmi {
namespace {

__global__ __launch_bounds__(512) void probe_mfma_f16_kernel(const f16x8* __restrict__ src, float* __restrict__ dst, int iters,
                                                              unsigned long long* __restrict__ clk) {
  const int t = threadIdx.x, nt = blockDim.x;
  f16x8 a[2][4], b[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a[s][i] = src[(size_t)(s * 8 + i) * nt + t];
      b[s][i] = src[(size_t)(s * 8 + 4 + i) * nt + t];
    }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc[4][4] = {};
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) CLIPMI_VALU_TO_MFMA_FENCE(acc[u][v]);   // zeroed by v_mov and read as SrcC by the first MFMAs (tools/mfma_hazard_scan.py)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {   // the A operand is held for four instructions, B changes every instruction (the GEMM loops' order)
          acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s][u], b[s][v], acc[u][v], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) r += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
  dst[(size_t)blockIdx.x * nt + t] = r;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (clk && blockIdx.x == 0 && t == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

extern "C" int clipmi_probe_mfma_f16(const void* operands, float* sink, unsigned long long* clocks, int waves, int iters, int* n_cus_out,
                                     clipmi_stream_t stream) {
  CLIPMI_REQUIRE(operands && sink, CLIPMI_ERR_ARG, "probe_mfma: null pointer");
  CLIPMI_REQUIRE(waves >= 1 && waves <= 8 && iters >= 1, CLIPMI_ERR_SHAPE, "probe_mfma: waves=%d (1..8) iters=%d", waves, iters);
  CLIPMI_REQUIRE((uintptr_t)operands % 16 == 0, CLIPMI_ERR_ARG, "probe_mfma: operands must be 16-byte aligned");
  const int cus = device_cus();
  if (n_cus_out) *n_cus_out = cus;
  hipLaunchKernelGGL(probe_mfma_f16_kernel, dim3(cus), dim3(waves * 64), 0, (hipStream_t)stream, static_cast<const f16x8*>(operands), sink,
                     iters, clocks);
  return check_launch("probe_mfma_f16_kernel");
}
