// Probe (not part of the library): the arithmetic core of the long-sequence attention kernel on register-resident operands -- no LDS, no DMA, no barrier --
// in two forms: (A) one 32-query tile per wave, as attention_ring_kernel's ring_attend does it (S^T = K Q^T: 4 MFMAs -> maximum -> 16 fma + 16 v_exp + 8 cvt_pk
// -> O^T += V^T P^T and the row sums: 6 MFMAs), and (B) TWO query tiles per wave, software-pipelined so that one tile's MFMAs are issued under the other tile's
// exponent work.  Cycles per (32 x 32) score tile and SIMD with 1 / 2 / 3 waves per SIMD: is form B worth a kernel?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/attn_core tools/probes/attn_core.hip && /tmp/attn_core
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 half_t;

struct Tile {
  f16x8 qf[4];
  f32x16 oacc[2], lacc;
  float m_run;
};

__device__ __forceinline__ f32x16 s_tile(const f16x8 (&kf)[4], const f16x8 (&qf)[4]) {
  const f32x16 zero = {0};
  f32x16 s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0], zero, 0, 0, 0);
#pragma unroll
  for (int ks = 1; ks < 4; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[ks], s, 0, 0, 0);
  return s;
}
__device__ __forceinline__ float tile_max(const f32x16& s, float m_run) {
  float m = m_run;
#pragma unroll
  for (int e = 0; e < 16; e += 2) m = fmaxf(fmaxf(s[e], s[e + 1]), m);
  const unsigned u = __builtin_bit_cast(unsigned, m);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(m, fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1])));
}
__device__ __forceinline__ f16x8 p_half(const f32x16& s, int ss, float mc) {
  constexpr float C = 0.125f * 1.4426950408889634f;
  f16x8 pf;
#pragma unroll
  for (int j = 0; j < 8; ++j) pf[j] = (half_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[8 * ss + j], C, -mc));
  return pf;
}
__device__ __forceinline__ void pv(Tile& t, const f16x8& pf, const f16x8 (&v8)[2], const f16x8& ones) {
  f16x8 p = pf;
  asm volatile("s_nop 3" : "+v"(p));
  t.oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v8[0], p, t.oacc[0], 0, 0, 0);
  t.oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v8[1], p, t.oacc[1], 0, 0, 0);
  t.lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, p, t.lacc, 0, 0, 0);
}

template <int FORM>
__global__ __launch_bounds__(FORM == 0 ? 768 : 512) void core_kernel(long long* out, float* sink, int n_key_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr float C = 0.125f * 1.4426950408889634f;
  f16x8 kf[4], v8[2], ones;
  Tile a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      kf[i][j] = (half_t)(0.01f * ((lane * 7 + i * 3 + j) % 13) - 0.06f);
      a.qf[i][j] = (half_t)(0.02f * ((lane * 5 + i + j * 3) % 11) - 0.1f);
      b.qf[i][j] = (half_t)(0.02f * ((lane * 3 + i * 5 + j) % 7) - 0.06f);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { v8[0][j] = (half_t)(0.01f * (lane % 9)); v8[1][j] = (half_t)(0.02f * (lane % 5)); ones[j] = (half_t)1.f; }
  a.oacc[0] = a.oacc[1] = a.lacc = b.oacc[0] = b.oacc[1] = b.lacc = f32x16{0};
  a.m_run = b.m_run = -1e30f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if constexpr (FORM == 0) {
    for (int kt = 0; kt < n_key_tiles; kt += 2) {   // groups of two key tiles, as ring_attend<2>
      asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(v8[0]), "+v"(v8[1]));   // "new" fragments every key tile
      const f32x16 s0 = s_tile(kf, a.qf);
      asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
      const f32x16 s1 = s_tile(kf, a.qf);
      const float m_new = tile_max(s1, tile_max(s0, a.m_run));
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(m_new > a.m_run) != 0ull, 0)) {
        const float alpha = __builtin_amdgcn_exp2f((a.m_run - m_new) * C);
#pragma unroll
        for (int e = 0; e < 16; ++e) { a.oacc[0][e] *= alpha; a.oacc[1][e] *= alpha; a.lacc[e] *= alpha; }
        asm volatile("s_nop 3" : "+v"(a.oacc[0]), "+v"(a.oacc[1]), "+v"(a.lacc));
      }
      a.m_run = m_new;
      const float mc = m_new * C;
      pv(a, p_half(s0, 0, mc), v8, ones);
      pv(a, p_half(s0, 1, mc), v8, ones);
      asm volatile("" : "+v"(v8[0]), "+v"(v8[1]));
      pv(a, p_half(s1, 0, mc), v8, ones);
      pv(a, p_half(s1, 1, mc), v8, ones);
    }
  } else {
    // two query tiles: tile B runs half a key tile behind tile A -- S_A(kt) is issued under the exponent work of B(kt - 1), P.V_B(kt - 1) under the maximum and
    // exponent work of A(kt), S_B(kt) under the rest of it, P.V_A(kt) under the maximum and exponent work of B(kt).  Source order = intended issue order.
    f32x16 sb = s_tile(kf, b.qf);
    float mcb;
    {
      const float m_new = tile_max(sb, b.m_run);
      b.m_run = m_new;
      mcb = m_new * C;
    }
    for (int kt = 0; kt < n_key_tiles; ++kt) {
      asm volatile("" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(v8[0]), "+v"(v8[1]));
      // --- S_A(kt)  ||  P_B(kt - 1)
      const f32x16 sa = s_tile(kf, a.qf);
      const f16x8 pb0 = p_half(sb, 0, mcb), pb1 = p_half(sb, 1, mcb);
      // --- P.V_B(kt - 1)  ||  maximum + P_A(kt)
      pv(b, pb0, v8, ones);
      pv(b, pb1, v8, ones);
      const float ma = tile_max(sa, a.m_run);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(ma > a.m_run) != 0ull, 0)) {
        const float alpha = __builtin_amdgcn_exp2f((a.m_run - ma) * C);
#pragma unroll
        for (int e = 0; e < 16; ++e) { a.oacc[0][e] *= alpha; a.oacc[1][e] *= alpha; a.lacc[e] *= alpha; }
        asm volatile("s_nop 3" : "+v"(a.oacc[0]), "+v"(a.oacc[1]), "+v"(a.lacc));
      }
      a.m_run = ma;
      const float mca = ma * C;
      // --- S_B(kt)  ||  P_A(kt)
      sb = s_tile(kf, b.qf);
      const f16x8 pa0 = p_half(sa, 0, mca), pa1 = p_half(sa, 1, mca);
      // --- P.V_A(kt)  ||  maximum of B(kt)
      pv(a, pa0, v8, ones);
      pv(a, pa1, v8, ones);
      const float mb = tile_max(sb, b.m_run);
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(mb > b.m_run) != 0ull, 0)) {
        const float alpha = __builtin_amdgcn_exp2f((b.m_run - mb) * C);
#pragma unroll
        for (int e = 0; e < 16; ++e) { b.oacc[0][e] *= alpha; b.oacc[1][e] *= alpha; b.lacc[e] *= alpha; }
        asm volatile("s_nop 3" : "+v"(b.oacc[0]), "+v"(b.oacc[1]), "+v"(b.lacc));
      }
      b.m_run = mb;
      mcb = mb * C;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc += a.oacc[0][e] + a.oacc[1][e] + a.lacc[e] + b.oacc[0][e] + b.oacc[1][e] + b.lacc[e];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int FORM>
int run(const char* what, int tiles_per_key_tile, long long* dev, float* sink, int waves) {
  const int n = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    core_kernel<FORM><<<256, waves * 64>>>(dev, sink, n);
    CHECK(hipDeviceSynchronize());
  }
  std::vector<long long> h(16);
  CHECK(hipMemcpy(h.data(), dev, 16 * 8, hipMemcpyDeviceToHost));
  long long mx = 0;
  for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
  const int per_simd = (waves + 3) / 4;
  printf("%-44s waves %2d: %9lld cycles = %7.1f per score tile and wave = %7.1f per score tile and SIMD\n", what, waves, mx,
         (double)mx / n / tiles_per_key_tile, (double)mx / n / tiles_per_key_tile / per_simd);
  return 0;
}

int main() {
  long long* dev;
  float* sink;
  CHECK(hipMalloc(&dev, 8 * 64));
  CHECK(hipMalloc(&sink, 4 * 256 * 768));
  for (int waves : {4, 8, 12}) {
    run<0>("A: one query tile per wave (as shipped)", 1, dev, sink, waves);
    if (waves <= 8) run<1>("B: two query tiles per wave, interleaved", 2, dev, sink, waves);   // 2 x the accumulators: 256 registers, two waves per SIMD at most
  }
  return 0;
}
