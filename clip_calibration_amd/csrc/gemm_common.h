// Shared by the GEMM translation units (gemm.hip, gemm_rstream.hip): kernel arguments, the LayerNorm-fold row sums, the inline-asm
// fragment reads and counted waits of the ping-pong loops.  Internal to libclipmi.so.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef CLIPMI_STORE_AUX
#define CLIPMI_STORE_AUX 0   // cache policy bits of the big output stores (2 = nt); build-time A/B: make libclipmi_gemmnt.so
#endif

// CLIPMI_ABLATE (build-time, diagnostic builds only: results are wrong with any bit set): 2 no LDS-DMA pieces inside the K loop of
// the streamed-epilogue kernel, 4 no MFMAs, 16 no fragment reads (the registers are left as they are)
#ifndef CLIPMI_ABLATE
#define CLIPMI_ABLATE 0
#endif

namespace clipmi {
namespace gemm {

constexpr int BK = 64;

struct KArgs {
  const half_t* A; int64_t lda;
  const half_t* W; int64_t ldw;
  const float* bias;
  const float* residual;
  const half_t* residual16;   // BIAS_RESIDUAL16_RELU
  void* out; int64_t ldo;
  int M, N, K;
  const float* pos; int patches; int tokens;
  int im_R, im_P, im_G;   // implicit im2col (gemm_pp_kernel<..., IM2COL>): A is the fp16 NCHW image [M / patches, 3, R, R], patch size P, grid G = R / P
  int tiles_n; int nwg;
  int band;      // n-tiles per band of the tile traversal (see tile_coords)
  int per_band; uint32_t mg_per_band, mg_band, mg_gw_last;   // gemm_stream_kernel: tiles per band and the multipliers of its divisions (launch_stream)
  const float* ln_stats; int ln_parts; const float* ln_g; float ln_inv_d; float ln_eps;
  int ln_M;      // rows per partial plane of ln_stats ([parts][ln_M] float2): the producer's M -- not this launch's, when a launch covers a row range of it
  int ln_rs;     // statistics rows per GEMM row (1; L for the class rows of a token-major buffer): tile kernels only
  half_t* x16; float* stats_out;
#ifdef CLIPMI_TUNING
  long long* stamps;   // diagnostic build only (make tuning, tools/gemm_stamps.py): per-workgroup s_memrealtime stamps
  int knob;            // diagnostic build only: ablation bits of the streamed-epilogue kernel (timing only, results wrong)
#endif
};

// Row partials of the LayerNorm fold, one lane's four consecutive values of a row: explicit fused multiply-adds, so that every
// kernel that produces them (the three fold epilogues below and gemm_rstream_kernel) rounds alike whatever the surrounding code
// lets the compiler contract -- the persistent kernel is tested bit for bit against the one-tile-per-workgroup kernels.
__device__ __forceinline__ void fold_row_sums(const f32x4& v, float& rsum, float& rsq) {
  rsum += (v[0] + v[1]) + (v[2] + v[3]);
  rsq += __builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]);
}
// The same for the fp16 stream, whose partials are those of the ROUNDED row: v_dot2_f32_f16 on the fp16 pairs (fp32 accumulation),
// four instructions per four elements instead of four conversions + eight adds / fused multiply-adds.
__device__ __forceinline__ void fold_row_sums16(const f16x4& h, float& rsum, float& rsq) {
  const f16x2 lo = f16x2{h[0], h[1]}, hi = f16x2{h[2], h[3]}, one = f16x2{(half_t)1.f, (half_t)1.f};
  rsum = __builtin_amdgcn_fdot2(hi, one, __builtin_amdgcn_fdot2(lo, one, rsum, false), false);
  rsq = __builtin_amdgcn_fdot2(hi, hi, __builtin_amdgcn_fdot2(lo, lo, rsq, false), false);
}

// Sum of a value over the four lanes that hold one row of a 16 x 16 accumulator block (lanes l, l ^ 16, l ^ 32, l ^ 48), valid in the
// lanes of the first 16-lane row (g4 == 0) -- the only ones that use it.  v_permlane16_swap / v_permlane32_swap of the value with
// itself bring the partner's copy into the lane: no LDS round trip (__shfl_xor compiles to ds_bpermute: two dependent ~100-cycle
// trips per sum).  Same association as  v += shfl_xor(v, 16); v += shfl_xor(v, 32)  in those lanes: bit-identical.
__device__ __forceinline__ float row4_sum(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // [0]: rows (0, 0, 2, 2) of v, [1]: rows (1, 1, 3, 3)
  const float s = __builtin_bit_cast(float, (unsigned int)a[0]) + __builtin_bit_cast(float, (unsigned int)a[1]);
  const unsigned int w = __builtin_bit_cast(unsigned int, s);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);   // [0]: lower half of s in both halves, [1]: upper half
  return __builtin_bit_cast(float, (unsigned int)b[0]) + __builtin_bit_cast(float, (unsigned int)b[1]);
}

// a += float(h), one v_fma_mix_f32 per element (fp16 source operand, fp32 accumulator: a + h rounded once, the bits of a conversion
// followed by an add) instead of a conversion and half a packed add -- the residual is added inside the K loop of
// gemm_rstream_kernel, where every VALU issue slot is taken from the partner wave's MFMAs
__device__ __forceinline__ void add_f16x4(f32x4& a, const f16x4& h) {
  typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ r = __builtin_bit_cast(u32x2_, h);
  float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a0) : "v"(r[0]));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a1) : "v"(r[0]));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel_hi:[1,0,0]" : "+v"(a2) : "v"(r[1]));
  asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a3) : "v"(r[1]));
  a = f32x4{a0, a1, a2, a3};
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS fragment reads as inline asm (pinned where they are written; hipcc's waitcnt pass does not see them) and the counted
// waits that name their destinations (cdna_hip_programming.md §5.7, form (ii)).
// (a function template, not a macro used inside the kernel's generic lambdas: clang does not implicitly capture a variable
// that a generic lambda names only as an asm operand)
template <int OFF>
__device__ __forceinline__ void ds_read128(f16x8& dst, uint32_t addr) {
  if constexpr (CLIPMI_ABLATE & 16) asm volatile("" : "=v"(dst) : "v"(addr));
  else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lgkm_wait1(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N>
__device__ __forceinline__ void lgkm_wait4(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait8(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e, f16x8& f, f16x8& g, f16x8& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lgkm_wait5(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}

template <int N, int H>
__device__ __forceinline__ void lgkm_wait_x(f16x8 (&x)[H]) {
  if constexpr (H == 4) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N));
  else asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]) : "n"(N));
  static_assert(H == 4 || H == 5, "half of the wave tile: 4 or 5 activation blocks");
}

// the persistent kernels address a tile with 32-bit byte offsets from its first row (257 rows of any operand must fit) and map
// virtual block ids to XCD labels through a grid that is a multiple of 8
inline bool stream_offsets_ok(const KArgs& k) {
  const int64_t lim = (1ll << 31) / (2 * 257);
  return k.lda < lim && k.ldw < lim && k.ldo < lim && (device_cus() & ~7) >= 8;
}
// gemm_stream_kernel (round 6) addresses every operand from the start of its matrix with ONE descriptor per launch: a whole matrix (plus the rows
// a lane offset can reach beyond it) has to stay below 2 GiB, and so do the tile ids times their divisors (div_magic)
inline bool stream_whole_matrix_ok(const KArgs& k) {
  const int64_t lim = (1ll << 31) - (1ll << 24);
  const int64_t tiles = (int64_t)((k.M + 255) / 256) * ((k.N + 255) / 256);
  return ((int64_t)k.M + 256) * k.lda * 2 < lim && ((int64_t)k.N + 256) * k.ldw * 2 < lim && ((int64_t)k.M + 256) * k.ldo * 2 < lim &&
         (!k.ln_stats || ((int64_t)k.ln_parts * k.ln_M + 256) * 8 < lim) && tiles * tiles < (1ll << 32);
}

// gemm_rstream.hip: the persistent row-range kernel of the fp16-stream residual GEMMs (variant 16)
bool rstream_fits(const KArgs& k);
int launch_rstream(KArgs k, hipStream_t s);

}  // namespace gemm
}  // namespace clipmi
