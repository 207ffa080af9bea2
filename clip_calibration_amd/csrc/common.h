// Internal declarations shared by the libclipmi.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include "../../include/clipmi.h"

namespace clipmi {

typedef _Float16 half_t;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CLIPMI_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define CLIPMI_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS-DMA through the buffer path (buffer_load_dwordx4 ... lds): 16 B per lane from rsrc.base + voff + soff into
// lds_base + lane*16.  Preferred over global_load_lds: hipcc treats the latter as a FLAT access that may touch LDS and
// then waits lgkmcnt(0) before every ds_read consumer (no counted waits); the MUBUF form keeps counted lgkmcnt and
// gives hardware bounds checking (bytes at or beyond num_records read as 0).  Build the descriptor from wave-uniform
// values only (kernel arguments, blockIdx).
#ifdef __HIPCC__
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
  const uint32_t n = bytes <= 0 ? 0u : (bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes);
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}
// A plain (non-template) function on purpose: when the builtin sits directly inside a kernel template with
// value-dependent operands, hipcc's host pass silently drops the kernel's host stub (undefined symbol at dlopen).
__device__ __forceinline__ void buffer_load_lds16(__amdgpu_buffer_rsrc_t rsrc, const void* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, CLIPMI_LDS_PTR(lds), 16, voff, soff, 0, 0);
}
#define CLIPMI_BUFFER_LOAD_LDS16(rsrc, lds, voff, soff) ::clipmi::buffer_load_lds16((rsrc), (lds), (voff), (soff))
// the same with cache-policy bits (1 = sc0, 2 = sc1, 4? = nt on gfx950's buffer instructions: 2 is what __builtin_nontemporal_load emits); build-time A/Bs only
template <int AUX>
__device__ __forceinline__ void buffer_load_lds16_aux(__amdgpu_buffer_rsrc_t rsrc, const void* lds, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, CLIPMI_LDS_PTR(lds), 16, voff, soff, 0, AUX);
}

// A register that VALU instructions have just written (conversions, transcendentals: the softmax's P, the tail's hi / lo split) and that
// an MFMA reads as a SOURCE operand right behind them needs wait states that hipcc does not insert on gfx950 for these sequences (it
// guarantees two).  The wave's own result is right either way -- every parity test passed without them -- but a wave of ANOTHER kernel
// resident on the same SIMD came back with a 16-lane quarter of one register overwritten (profiles/r03_gpu_sharing.txt: 24-45 of 400
// LayerNorm launches beside the attention kernel, 0 of 400 with the fence; the stand-alone reproducer and its wait-state table:
// tools/probes/mfma_hazard_repro.hip, profiles/r04_hazard_repro.txt).  `s_nop 3` = FOUR wait states by itself, which is what
// tools/mfma_hazard_scan.py (WAIT = 4) demands of every VALU-write -> MFMA-source pair: the guard holds by construction, not by what the
// scheduler happens to place in between.  The operand is named "+v" so that every instruction that writes it stays above.
// CLIPMI_FENCE_SNOP (build-time, diagnostic builds only -- `make fence_sweep`, tools/probes/hazard_fence_sweep.py): the s_nop operand of
// the fence, -1 = no wait states at all (what the library was until round 3).  The product build never defines it.
#ifndef CLIPMI_FENCE_SNOP
#define CLIPMI_FENCE_SNOP 3
#endif
#if CLIPMI_FENCE_SNOP < 0
#define CLIPMI_VALU_TO_MFMA_FENCE(x) asm volatile("" : "+v"(x))
#define CLIPMI_VALU_TO_MFMA_FENCE2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#define CLIPMI_VALU_TO_MFMA_FENCE3(x, y, z) asm volatile("" : "+v"(x), "+v"(y), "+v"(z))
#else
#define CLIPMI_VALU_TO_MFMA_FENCE(x) asm volatile("s_nop %1" : "+v"(x) : "n"(CLIPMI_FENCE_SNOP))
#define CLIPMI_VALU_TO_MFMA_FENCE2(x, y) asm volatile("s_nop %2" : "+v"(x), "+v"(y) : "n"(CLIPMI_FENCE_SNOP))
#define CLIPMI_VALU_TO_MFMA_FENCE3(x, y, z) asm volatile("s_nop %3" : "+v"(x), "+v"(y), "+v"(z) : "n"(CLIPMI_FENCE_SNOP))
#endif

// The other direction (round 5, profiles/r05_vitl_attention.txt "second hazard"): a VGPR an MFMA has just written, read by a VECTOR instruction.  hipcc
// inserts wait states for this from its own table (`s_nop 10` behind a v_mfma_f32_32x32x16_f16 in the case that showed it) and the wave's own value is
// right -- every parity test passed --, but beside a kernel that did this ten times per block (a per-group `sum += acc[0]` right behind the group's
// last MFMAs) a co-resident wave of ANOTHER kernel lost registers: 52-57 of 200 LayerNorm launches wrong beside the 257-token ring attention, 127-160 of
// 200 beside the 577-token one; 0 of 200 with 32 more wait states in front of the read, and 0 with the read moved out of the loop
// (tools/probes/ring_hazard_probe.py).  Kernels here keep MFMA results in the matrix pipe's hands inside their loops and put this fence in front of
// the one place per pass where the vector pipe takes the accumulators over.
#ifndef CLIPMI_MFMA_TO_VALU_FENCE3   // (a diagnostic build may define it away on the command line: tools/lib_ab.py timing of the fence itself)
#define CLIPMI_MFMA_TO_VALU_FENCE3(x, y, z) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(x), "+v"(y), "+v"(z))
#endif

// ---------------------------------------------------------------------------------------------------------------
// Wave-wide reductions without LDS (DPP inside a 16-lane row, v_permlane16/32_swap across rows): every lane ends up with the
// result, by the same tree in every lane -- deterministic, and ~10x shorter than six dependent ds_bpermute round trips.  A LATENCY
// optimisation, used where a reduction sits on a dependent chain (the fused tail's row pass, logits.hip).  layernorm.hip
// (ln_wave_sum) and attention.hip (the row maximum) keep the __shfl_xor butterfly, and correctly so: the wrong LayerNorm rows of
// profiles/r03_gpu_sharing.txt were first blamed on ds_bpermute and turned out to be the VALU -> MFMA hazard above, in the OTHER kernel.
//   step 1, 2: quad_perm (xor 1, xor 2)   3: row_half_mirror   4: row_mirror   5: rows (0,1) (2,3)   6: halves
// ---------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float swap16_f(float v) {   // the value of the lane 16 further / back (rows 0 <-> 1, 2 <-> 3)
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // [0]: rows (0, 0, 2, 2), [1]: rows (1, 1, 3, 3)
  const bool odd = (threadIdx.x >> 4) & 1;
  return __builtin_bit_cast(float, (unsigned int)(odd ? a[0] : a[1]));
}
__device__ __forceinline__ float swap32_f(float v) {   // the value of the lane 32 further / back
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // [0]: lower half twice, [1]: upper half twice
  const bool up = (threadIdx.x >> 5) & 1;
  return __builtin_bit_cast(float, (unsigned int)(up ? a[0] : a[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror
  v += dpp_f<0x140>(v);   // row_mirror
  v += swap16_f(v);
  v += swap32_f(v);
  return v;
}
// (value, index) -> the largest value and the LOWEST index that holds it, in every lane
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
  auto take = [&](float ov, int oi) {
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  };
  take(dpp_f<0xB1>(v), dpp_i<0xB1>(idx));
  take(dpp_f<0x4E>(v), dpp_i<0x4E>(idx));
  take(dpp_f<0x141>(v), dpp_i<0x141>(idx));
  take(dpp_f<0x140>(v), dpp_i<0x140>(idx));
  take(swap16_f(v), __builtin_bit_cast(int, swap16_f(__builtin_bit_cast(float, idx))));
  take(swap32_f(v), __builtin_bit_cast(int, swap32_f(__builtin_bit_cast(float, idx))));
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  v = fmaxf(v, swap16_f(v));
  return fmaxf(v, swap32_f(v));
}
#endif

void set_error(const char* fmt, ...);
int check_launch(const char* what);  // hipGetLastError -> CLIPMI_ERR_HIP

// Runtime switches (clipmi_set_option / clipmi_get_option).  The CLIPMI_* environment variables of the same meaning are
// read ONCE, the first time any option is looked at; nothing on a launch path calls getenv.  Plain atomics: a switch
// may be flipped between launches from any thread.
struct Options {
  std::atomic<int> gemm_variant{-1};   // CLIPMI_GEMM_VARIANT: -1 = default dispatch; 0 / 1 / 10 / 13 / 16 force one kernel family where the shape allows it (test aid)
  std::atomic<int> gemm_band{0};       // CLIPMI_GEMM_BAND: 0 = default traversal band
  std::atomic<int> gemm_stream{1};     // CLIPMI_GEMM_STREAM: 1 (default) = ping-pong persistent kernel with streamed epilogue for multi-round fp16-out GEMMs
                                       // with K >= 512 (in-proj / c_fc); 0 = one tile per workgroup (the bit-identity reference of the race screen)
  std::atomic<int> gemm_split_rows{1}; // CLIPMI_GEMM_SPLIT_ROWS: 1 (default) = a ragged last row of tiles that would open a round of its own in the persistent
                                        // kernel goes to the tile kernels as a second launch (gemm.hip launch_one); 0 = one launch
  std::atomic<int> gemm_rstream{1};    // CLIPMI_GEMM_RSTREAM: 1 (default) = persistent row-range kernel with streamed residual epilogue for the fp16-stream
                                       // residual GEMMs with K <= 1536 (out-proj); 0 = one 320 x 256 tile per workgroup (the same fp32 sum, added in another order)
  std::atomic<int> cls_only_last_block{1};   // CLIPMI_CLS_ONLY_LAST_BLOCK: 1 (default since round 6) = the image tower's last block computes K | V of every
                                             // token and everything else for the class rows only (run_block_step); 0 = every row (bench.py's headline `value`)
  std::atomic<int> ln_fold{1};         // CLIPMI_LN_FOLD
  std::atomic<int> residual_f16{2};    // CLIPMI_RESIDUAL_F16: 0 fp32 everywhere | 1 both towers | 2 image tower only (default, 'v') | 3 text tower only ('t')
  std::atomic<int> attn_loader{2};     // CLIPMI_ATTN_LOADER, 193..200-token non-causal attention: 2 (default) = attention_vision_nt_kernel (all operands by LDS-DMA
                                       // from a loader wave, fragment reads pinned by inline asm, output rows stored non-temporal); 1 = the same with plain
                                       // stores; 0 = persistent kernel (all three: same bits)
  std::atomic<int> attn_small{1};      // CLIPMI_ATTN_SMALL, attention over at most 32 tokens (the text tower after dead-row elimination): 1 (default) = one item per
                                       // wave, no workgroup barrier (attention_small_kernel); 0 = the persistent kernel
  std::atomic<int> attn_ring{1};       // CLIPMI_ATTN_RING, non-causal attention over more than 224 tokens (ViT-L/14): 1 (default) = attention_ring_kernel;
                                       // 0 = the round-1 streaming kernel
  std::atomic<int> tail_unfused{0};    // CLIPMI_TAIL_UNFUSED: 1 = the three-kernel logits tail (A/B aid)
  std::atomic<int> vision_pass{50432 * 768};   // CLIPMI_VISION_PASS: stream elements (token rows x width) of one pass of the image tower; larger batches run
                                               // as consecutive passes (each an ordinary call on its images); 0 = never split
};
// cls_only_last_block, ln_fold and residual_f16 are DEFAULTS: a model handle may override them (clipmi_model_set_option) and a
// tower call may override the stream precision (flags) -- nothing on a launch path writes to this struct.
Options& options();

#ifdef CLIPMI_TUNING
// Diagnostic build only (make TUNING=1): device buffer of 8 int64 per workgroup that the GEMM kernels stamp with
// s_memrealtime at their phase boundaries (clipmi_tuning_set_stamps, tools/gemm_stamps.py).  Absent from the product build.
extern std::atomic<long long*> g_tuning_stamps;
extern std::atomic<int> g_tuning_knob;   // ablation bits (clipmi_tuning_set_knob)
#endif

// Per-device state.  One process may drive several GPUs: kernel attributes (dynamic LDS size) are set once per
// (kernel, device) and the CU count is cached per device.  Thread-safe (idempotent HIP calls guarded by atomics).
int current_device();                  // hipGetDevice, -1 on failure
int device_cus();                      // multiProcessorCount of the current device (256 when it cannot be read)
struct DeviceOnce {
  std::atomic<uint64_t> mask{0};
  // true exactly until `mark()` has been called for the current device (devices >= 64 are never cached)
  bool needed(int dev) const { return dev < 0 || dev >= 64 || !((mask.load(std::memory_order_acquire) >> dev) & 1); }
  void mark(int dev) { if (dev >= 0 && dev < 64) mask.fetch_or(1ull << dev, std::memory_order_release); }
};
template <typename Fn>
inline void ensure_dynamic_lds(Fn fn, int bytes, DeviceOnce& once) {
  const int dev = current_device();
  if (!once.needed(dev)) return;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) (void)hipGetLastError();
  once.mark(dev);
}

#define CLIPMI_REQUIRE(cond, code, ...)        \
  do {                                         \
    if (!(cond)) {                             \
      ::clipmi::set_error(__VA_ARGS__);        \
      return (code);                           \
    }                                          \
  } while (0)

// Extra epilogue used only by the vision tower: scatter patch rows into the token matrix and add the
// positional embedding (clip/model.py:396-402).
enum { EPI_PATCH_POS = 100, EPI_RESIDUAL_FOLD = 101 /* BIAS_RESIDUAL + fp16 copy + row partials: own kernel instantiation */,
       EPI_RESIDUAL_FOLD16 = 102 /* same, but the residual stream itself is the fp16 copy: read x16, write x16, no fp32 pass */ };

struct GemmArgs {
  const half_t* A; int64_t lda;
  const half_t* W; int64_t ldw;
  const float* bias;
  const void* residual;   // fp32 [M,N] (BIAS_RESIDUAL) or fp16 [M,N] (BIAS_RESIDUAL16_RELU), ld = ldo
  void* out; int64_t ldo; int out_dtype;
  int M, N, K, epilogue;
  // EPI_PATCH_POS: out row = (m / patches) * tokens + (m % patches) + 1, plus pos[(m % patches) + 1][n]
  const float* pos; int patches; int tokens;
  // EPI_PATCH_POS with im_P != 0: implicit im2col -- A is the fp16 NCHW image batch [M / patches, 3, im_R, im_R] itself (conv1 has stride =
  // kernel = im_P in {8, 16, 32}: column k = c P^2 + ky P + kx of patch row m is a pixel address), K = 3 P^2, lda unused
  int im_R = 0, im_P = 0;
  // LayerNorm folding (see gemm.hip "LayerNorm folded into the GEMMs").  Row statistics live in a partial buffer
  // stats[p * M + m] = (sum, sum of squares) of row m over the p-th column tile of the producer; consumers add the
  // ln_parts partials in a fixed order (deterministic, no atomics, nothing to zero).
  //   consumer (BIAS / BIAS_QUICKGELU): ln_stats != NULL -> out = epi(rstd[m]*acc - rstd[m]*mean[m]*ln_g[n] + bias[n])
  //   producer (BIAS_RESIDUAL): x16 != NULL -> also store fp16(out) to x16 (ld = ldo), write this tile's row partials to
  //   stats_out and report the number of partials per row (= its n-tile count) through *parts_out (host pointer)
  const float* ln_stats; int ln_parts; const float* ln_g; int ln_dim; float ln_eps;
  float* ln_rows = nullptr;   // consumer, optional scratch [M][2] fp32: lets the streamed-epilogue kernel reduce the partials once per GEMM
  // consumer over a strided subset of the producer's rows (the class rows of the last image block): statistics row of GEMM row m =
  // ln_stats[p * ln_plane + m * ln_row_stride]; ln_plane 0 = M rows per partial plane
  int64_t ln_plane = 0; int ln_row_stride = 1;
  half_t* x16; float* stats_out; int* parts_out;
  bool residual_f16 = false;   // producer: the residual operand is x16 itself (fp16 stream, updated in place); `residual`/`out` unused
};
constexpr int LN_MAX_PARTS = 8;
int launch_gemm(const GemmArgs& a, hipStream_t s);

int launch_layernorm(const void* x, int x_dtype, int64_t in_stride, const int32_t* gather_idx, const float* gamma,
                     const float* beta, void* y, int y_dtype, int64_t out_stride, int rows, int D, float eps,
                     hipStream_t s, half_t* y16 = nullptr, float* stats_out = nullptr);
// x16[row,:] = fp16(x[row,:]); stats partial 0 of the row = (sum, sum of squares), partials 1..parts-1 = 0; for rows
// n*L + first + j, j < n_ctx.  M = N*L is the partial stride.
int launch_row_stats(const float* x, half_t* x16, float* stats, int parts, int N, int L, int D, int first, int n_ctx, hipStream_t s);
int launch_attention(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s);
// attention_cls.hip: the output row of token 0 of every sequence only (the image tower's last block: clip/model.py:419 reads nothing else)
int launch_attention_cls(const half_t* qkv, half_t* out, int N, int L, int H, hipStream_t s);
int launch_patchify(const void* image, int image_dtype, half_t* col, int B, int R, int P, int Kpad, hipStream_t s);
// patch_embed.hip: conv1 as a GEMM whose loader reads the (fp16) NCHW image directly (+ pos, token-row scatter) and ln_pre over every token row with the
// class / shallow-prompt rows formed on the fly   (clip/model.py:394-402,413)
bool patch_embed_fits(int B, int R, int P, int D);
size_t patch_embed_scratch_bytes(int B, int R, int image_dtype);   // fp16 copy of an fp32 image batch (0 for an fp16 image)
int launch_patch_embed(const void* image, int image_dtype, void* scratch, const half_t* conv_w, int64_t ldw, const float* pos, void* x0, int x0_dtype,
                       int B, int R, int P, int D, int tokens, hipStream_t s);
int launch_embed_ln(const void* x0, int x0_dtype, int add_pos, const float* cls, const float* pos, const float* shallow, const float* gamma,
                    const float* beta, float* y, half_t* y16, float* stats_out, int B, int L, int tokens0, int D, float eps, hipStream_t s);
// x0[b, 0, :] = cls + pos[0]; x0[b, tokens0 + j, :] = shallow[j] (MaPLe)     (clip/model.py:398-402,459-460)
int launch_cls_and_ctx_rows(float* x0, const float* cls, const float* pos, const float* shallow, int B, int tokens0,
                            int n_ctx, int D, hipStream_t s);
// x[n, first + j, :] = prompt[j, :]  for j < n_ctx       (clip/model.py:301-328)
int launch_overwrite_tokens(float* x, const float* prompt, int N, int L, int D, int first, int n_ctx, hipStream_t s);
// xres[c,l,:] = float(src[c,l,:]) + (pos ? pos[l,:] : 0)
int launch_add_pos(const void* src, int dtype, const float* pos, float* xres, int C, int L, int src_L, int D, hipStream_t s);  // first L of src_L rows
// xres[c,l,:] = table[ids[c,l],:] + pos[l,:] ; eot[c] = argmax_l ids[c,l]     (clip/model.py:601-603,611)
int launch_embed_tokens(const int64_t* ids, const float* table, const float* pos, float* xres, int32_t* eot, int C,
                        int L, int src_L, int D, int vocab, hipStream_t s);   // ids [C, src_L]; the first L positions are embedded
int launch_eot_rows(const int32_t* eot, int32_t* rows, int C, int L, hipStream_t s);  // rows[c] = c*L + eot[c]
int launch_cast_f32(const float* src, void* dst, int dtype, int64_t n, hipStream_t s);
int launch_cast_f16(const half_t* src, void* dst, int dtype, int64_t n, hipStream_t s);
int launch_rows_out(const void* src, int src_dtype, void* dst, int dst_dtype, int C, int rows, int L, int D, hipStream_t s);   // [C*rows, D] -> [C, L, D], zeros behind `rows`
int launch_group_mean(const float* in, float* out, int G, int P, int E, hipStream_t s);
int launch_l2_normalize(const void* in, int in_dtype, float* out, int rows, int E, hipStream_t s);
int launch_l2_normalize_to(const void* in, int in_dtype, void* out, int out_dtype, int rows, int E, hipStream_t s);
int launch_logits(const float* img_n, const float* txt_n, float scale, const float* dac_conf, float* logits,
                  float* conf, int32_t* pred, int B, int C, int E, hipStream_t s);
size_t fused_tail_workspace_bytes(int B, int C);
int launch_fused_tail(const void* img, int img_dtype, int normalize, const float* txt_n, float scale, const float* dac_conf, float* logits,
                      float* img_n_out, float* conf, int32_t* pred, const int64_t* labels, double* bins, int n_bins, void* workspace,
                      size_t workspace_bytes, int B, int C, int E, hipStream_t s);
int launch_calibrate_rows(float* logits, const float* dac_conf, float* conf, int32_t* pred, int B, int C, hipStream_t s);
int launch_softmax_rows(const float* logits, const float* dac_conf, float* probs, float* conf, int32_t* pred, int B, int C,
                        hipStream_t s);
int launch_conv3x3(const half_t* x, const half_t* w, const float* bias, half_t* out, int B, int H, int W, int C, int Cout, int relu,
                   hipStream_t s);
int launch_im2col3x3_nchw(const void* image, int dtype, half_t* col, int B, int Cin, int H, int W, int stride, int Kpad, hipStream_t s);
int launch_im2col3x3_nhwc(const half_t* x, half_t* col, int B, int H, int W, int C, int Kpad, hipStream_t s);
int launch_avgpool_nhwc(const half_t* x, half_t* y, int B, int H, int W, int C, int k, hipStream_t s);
int launch_attnpool_tokens(const half_t* x, const float* pos, half_t* tokens, int B, int HW, int C, hipStream_t s);
int launch_attnpool(const half_t* q, const half_t* kv, half_t* out, int B, int T, int heads, hipStream_t s);
int launch_adapter_blend(const float* f, const float* w1, const float* w2, float ratio, float* out, int B, int E, int H, hipStream_t s);
int launch_scale_add(const float* a, const float* b, float alpha, float* out, int64_t n, hipStream_t s);
int launch_cocoop_ctx(const float* img_n, const float* w1, const float* b1, const float* w2, const float* b2, const float* ctx,
                      float* ctx_shifted, int B, int E, int H, int D, int n_ctx, hipStream_t s);
int launch_cocoop_prompts(const void* base, int base_dtype, const float* ctx_shifted, half_t* prompts, int nb, int C, int L, int D,
                          int n_ctx, hipStream_t s);
int launch_logits_per_image(const float* img_n, const float* txt, float scale, const float* dac_conf, float* logits, float* conf,
                            int32_t* pred, float* txt_n_last, int B, int C, int E, hipStream_t s);
int launch_ece_accumulate(const float* conf, const int32_t* pred, const int64_t* labels, int n, double* bins,
                          int n_bins, hipStream_t s);

int launch_knn(const float* q, const float* refs, float* out, int Nq, int Nr, int E, int K, hipStream_t s);

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

}  // namespace clipmi
