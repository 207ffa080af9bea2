// Stand-alone reproducer for the VALU-write -> MFMA-source hazard of profiles/r03_gpu_sharing.txt (common.h CLIPMI_VALU_TO_MFMA_FENCE).
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_hazard_repro.hip -o /tmp/mfma_hazard_repro && /tmp/mfma_hazard_repro [launches]
//
// Two kernels on two streams of one process:
//   victim   the LayerNorm shape of the library: one wave per 768-wide row, the row in 12 registers per lane, two xor-butterfly reductions,
//            ~56 VGPRs, no LDS allocation, no MFMA -- small enough to sit beside the hammer's waves on a SIMD.  Its output is compared bit for
//            bit with a run on an otherwise idle GPU.
//   hammer   the softmax -> P.V step of an attention kernel reduced to its instruction pattern: v_fma_f32 -> v_exp_f32 -> v_cvt_pk_f16_f32 into
//            the four registers of an fp16 fragment, then -- WS wait states later, exactly: the whole sequence is ONE inline-asm block, so no
//            compiler pass can pad or reorder it -- v_mfma_f32_32x32x16_f16 with that fragment as its B operand, in a loop.  Variants:
//            'r' all operands in registers; 'l' the A operand re-read from LDS every step (ds_read_b64_tr_b16 pairs, as the V tiles are).
// For each WS the table gives the victim launches (of N) whose output differs, and the hammer's own checksum against WS = 8 (the wave's
// own result is expected to be right at every WS >= the 2 that hipcc guarantees; below that the sequence is illegal even for the wave
// itself and is listed only to show where the hardware interlock ends).  Hand-written sequences carry every hazard themselves: the first
// version of this probe fed v_exp_f32 results straight into v_cvt_pk_f16_f32 -- the transcendental-result -> VALU-use hazard, which hipcc
// pads and inline asm does not -- and a quarter of its own checksums changed from run to run at every WS.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// ---- victim: layernorm.hip's kernel shape (D = 768: 3 float4 per lane)
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * 768;
  f32x4 v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i] = *reinterpret_cast<const f32x4*>(xr + (lane + i * 64) * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / 768.f;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / 768.f + 1e-5f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = (lane + i * 64) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
    *reinterpret_cast<f32x4*>(y + (size_t)row * 768 + c) = o;
  }
}

// ---- hammer.  v[124:127] is the fragment the VALU writes and the MFMA reads (named physical registers: an inline-asm operand cannot
// address the sub-registers of a 128-bit tuple).  NOPS = the s_nop text between the last write and the MFMA.
#define HAMMER_STEP(NOPS)                                                                                                    \
  asm volatile("v_fma_f32 %[t0], %[s0], %[c], %[m]\n\tv_fma_f32 %[t1], %[s1], %[c], %[m]\n\t"                                 \
               "v_exp_f32 %[t0], %[t0]\n\tv_exp_f32 %[t1], %[t1]\n\ts_nop 1\n\t" /* trans result -> VALU use: its own wait states */   \
               "v_cvt_pk_f16_f32 v124, %[t0], %[t1]\n\tv_cvt_pk_f16_f32 v125, %[t1], %[t0]\n\t"                               \
               "v_cvt_pk_f16_f32 v126, %[t0], %[t0]\n\tv_cvt_pk_f16_f32 v127, %[t1], %[t1]\n\t" NOPS                          \
               "v_mfma_f32_32x32x16_f16 %[acc], %[a], v[124:127], %[acc]\n\t"                                                 \
               : [acc] "+v"(acc), [t0] "=&v"(t0), [t1] "=&v"(t1)                                                              \
               : [a] "v"(afrag), [s0] "v"(s0), [s1] "v"(s1), [c] "v"(c), [m] "v"(m)                                           \
               : "v124", "v125", "v126", "v127")

template <int WS, bool LDSA>
__global__ __launch_bounds__(256, 2) void hammer_kernel(const f16x8* __restrict__ src, float* __restrict__ sink, int iters) {
  __shared__ f16x8 img[4 * 256];
  const int t = threadIdx.x;
  for (int i = 0; i < 4; ++i) img[i * 256 + t] = src[i * 256 + t];
  __syncthreads();
  f16x8 afrag = img[t];
  f32x16 acc = {};
  float s0 = 0.001f * (float)(t & 63), s1 = -0.002f * (float)(t & 31);
  const float c = 0.18f, m = -0.05f;
  float t0, t1;
  for (int it = 0; it < iters; ++it) {
    if constexpr (LDSA) {   // the A operand comes back from LDS every step, as the V tiles of an attention kernel do
      const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(img + ((it & 3) * 256 + t));
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(afrag) : "v"(addr) : "memory");
    }
    if constexpr (WS == 0) HAMMER_STEP("");
    else if constexpr (WS == 1) HAMMER_STEP("s_nop 0\n\t");
    else if constexpr (WS == 2) HAMMER_STEP("s_nop 1\n\t");
    else if constexpr (WS == 3) HAMMER_STEP("s_nop 2\n\t");
    else if constexpr (WS == 4) HAMMER_STEP("s_nop 3\n\t");
    else if constexpr (WS == 6) HAMMER_STEP("s_nop 5\n\t");
    else HAMMER_STEP("s_nop 7\n\t");
    s0 += 1e-4f;
    s1 -= 1e-4f;
  }
  float r = 0.f;
  for (int e = 0; e < 16; ++e) r += acc[e];
  sink[(size_t)blockIdx.x * 256 + t] = r;
}

template <bool LDSA>
static void launch_hammer(int ws, int grid, const f16x8* src, float* sink, int iters, hipStream_t s) {
  switch (ws) {
    case 0: hammer_kernel<0, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
    case 1: hammer_kernel<1, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
    case 2: hammer_kernel<2, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
    case 3: hammer_kernel<3, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
    case 4: hammer_kernel<4, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
    case 6: hammer_kernel<6, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
    default: hammer_kernel<8, LDSA><<<grid, 256, 0, s>>>(src, sink, iters); break;
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 300;
  const int rows = 197 * 256;
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<float> hx((size_t)rows * 768), hg(768, 1.f), hb(768, 0.f);
  srand(3);
  for (auto& v : hx) v = (float)(rand() % 65536) / 65536.f * 4.f - 2.f;
  std::vector<_Float16> hsrc(4 * 256 * 8);
  for (auto& v : hsrc) v = (_Float16)((float)(rand() % 65536) / 65536.f - 0.5f);
  float *x, *g, *b, *y, *yref, *sink;
  f16x8* src;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&y, hx.size() * 4)); CK(hipMalloc(&yref, hx.size() * 4));
  CK(hipMalloc(&g, 768 * 4)); CK(hipMalloc(&b, 768 * 4)); CK(hipMalloc(&src, hsrc.size() * 2));
  const int hgrid = 2 * cus;
  CK(hipMalloc(&sink, (size_t)hgrid * 256 * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(g, hg.data(), 768 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 768 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(src, hsrc.data(), hsrc.size() * 2, hipMemcpyHostToDevice));
  hipStream_t sv, sh;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
  victim_kernel<<<(rows + 3) / 4, 256, 0, sv>>>(x, g, b, yref, rows);   // reference: idle GPU
  CK(hipStreamSynchronize(sv));
  std::vector<float> href(hx.size()), hy(hx.size());
  CK(hipMemcpy(href.data(), yref, hx.size() * 4, hipMemcpyDeviceToHost));
  std::vector<float> hsink((size_t)hgrid * 256), hsink8((size_t)hgrid * 256);

  printf("victim: %d rows x 768 (LayerNorm shape, 1 wave per row); hammer: %d workgroups x 4 waves; %d victim launches per cell\n", rows, hgrid, launches);
  printf("%-28s | %-22s | %-34s | %s\n", "hammer variant", "wait states write->MFMA", "victim launches with a wrong output", "hammer's own result vs WS = 8");
  const int iters = 6000;
  for (int variant = 0; variant < 2; ++variant) {
    const int order[] = {8, 6, 4, 3, 2, 1, 0};
    for (int ws : order) {
      // the hammer's own checksum, alone on the GPU
      if (variant) launch_hammer<true>(ws, hgrid, src, sink, 64, sh); else launch_hammer<false>(ws, hgrid, src, sink, 64, sh);
      CK(hipStreamSynchronize(sh));
      CK(hipMemcpy((ws == 8 ? hsink8 : hsink).data(), sink, hsink.size() * 4, hipMemcpyDeviceToHost));
      const bool own_ok = ws == 8 || memcmp(hsink.data(), hsink8.data(), hsink.size() * 4) == 0;
      if (getenv("REPRO_DEBUG")) {
        const float* h = (ws == 8 ? hsink8 : hsink).data();
        size_t nd = 0;
        for (size_t i = 0; i < hsink.size(); ++i) nd += memcmp(&hsink8[i], &h[i], 4) != 0;
        printf("   ws %d: sink[0..3] = %.9g %.9g %.9g %.9g | sink[64] = %.9g | %zu of %zu words differ from ws 8\n", ws, h[0], h[1], h[2], h[3], h[64], nd, hsink.size());
      }
      int bad = 0;
      long long bad_words = 0;
      for (int l = 0; l < launches; ++l) {
        // keep two hammer launches queued so that the victim always runs beside one
        if (variant) { launch_hammer<true>(ws, hgrid, src, sink, iters, sh); launch_hammer<true>(ws, hgrid, src, sink, iters, sh); }
        else { launch_hammer<false>(ws, hgrid, src, sink, iters, sh); launch_hammer<false>(ws, hgrid, src, sink, iters, sh); }
        victim_kernel<<<(rows + 3) / 4, 256, 0, sv>>>(x, g, b, y, rows);
        CK(hipStreamSynchronize(sv));
        CK(hipMemcpy(hy.data(), y, hx.size() * 4, hipMemcpyDeviceToHost));
        if (memcmp(hy.data(), href.data(), hx.size() * 4) != 0) {
          ++bad;
          for (size_t i = 0; i < hy.size(); ++i) bad_words += memcmp(&hy[i], &href[i], 4) != 0;
        }
        CK(hipStreamSynchronize(sh));
      }
      printf("%-28s | %22d | %4d of %d (%lld words)%*s | %s\n", variant ? "A operand re-read from LDS" : "registers only", ws, bad, launches, bad_words, 8, "",
             own_ok ? "same bits" : "DIFFERENT");
      fflush(stdout);
    }
  }
  return 0;
}
