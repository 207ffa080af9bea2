// Patch embedding of the image tower as a GEMM that reads the image itself, and ln_pre as one row pass  (reference clip/model.py:394-402:
// conv1 -> reshape -> permute -> cat(class_embedding) -> + positional_embedding; :413 ln_pre; :597-598 the dtype cast of encode_image).
//
//   cast_image_kernel    fp32 NCHW pixels -> fp16 NCHW pixels (image.type(self.dtype), clip/model.py:598); skipped for an fp16 image
//   gemm_pp_kernel<IM2COL>  x0[b, 1 + p, :] = patch(b, p) @ conv_w^T + pos[1 + p]       (gemm.hip; M = B * G * G patches, N = width, K = 3 P^2)
//   embed_ln_kernel      x[b, l, :] = ln_pre(l == 0 ? cls + pos[0] : l < 1 + G^2 ? x0[b, l] : shallow[l - 1 - G^2])   -> the residual stream
//
// conv1 has stride = kernel = P, so its im2col matrix is a pure ADDRESS MAP of the NCHW image: column k = c P^2 + ky P + kx of patch
// (b, py, px) is pixel (b, c, py P + ky, px P + kx).  With P in {8, 16, 32} every 16-byte LDS slot of the activation operand (8 consecutive
// k) is 8 consecutive fp16 pixels of one image row, and a 64-deep K-step is 64 / P whole row segments of one channel: the operand is staged by
// LDS-DMA (buffer_load ... lds) exactly like a dense GEMM's, with the per-lane source offset = patch origin + (ky, kx) of the slot and the
// scalar offset = (channel, first ky) of the K-step.  No im2col matrix is written or read (77 MB each way at batch 256) and the GEMM's
// loader moves the same bytes the im2col GEMM's did.  The epilogue stores the token rows -- in fp16 through a wave-private LDS transpose (16 B
// per lane, 128 contiguous bytes per row) when the residual stream is fp16 (the reference's own GPU path holds the conv output in fp16:
// clip/model.py:395-397 on a convert_weights model), in fp32 when the stream is fp32 -- and can add the positional rows; the image tower lets
// embed_ln_kernel add them instead (fp32, before the statistics).  The class row and MaPLe's shallow prompt rows are the same for every image
// and never pass through memory: embed_ln_kernel forms them on the fly; it is layernorm_kernel's arithmetic (two-pass statistics in fp32, the same butterfly) with those
// three row sources and the outputs the blocks want (fp32 stream and / or fp16 operand copy + the LayerNorm-fold row sums of the output).
//
// Why the cast is its own pass (profiles/r04_patch_embed.txt): the first form of the GEMM staged fp32 pixels THROUGH REGISTERS (two
// 16-byte loads, four v_cvt_pk_f16_f32, one ds_write_b128 per slot) -- correct, and 140 us at batch 256: with 128 accumulator registers
// per lane a wave can hold one K-step of pixels in flight, 1 us of MFMAs to cover a load that takes ~3 us when every CU pulls 64 KB of
// fp32 per K-step through its L1, three times over (one per n-tile).  A 231 MB streaming cast (38 us) in front of an all-DMA GEMM is faster.
//
// Algorithmic bytes per image (ViT-B/16, fp32 input): cast 602 KB in + 301 KB out; GEMM 301 KB of pixels (x the n-tiles that miss L2) in,
// 302 KB of fp16 rows out; ln_pre 302 KB in, 302 KB out.  Before: patchify 602 + 301 KB, GEMM 301 KB + 605 KB of fp32 rows, class rows,
// ln_pre 605 + 302 KB.
#include "gemm_common.h"

namespace clipmi {
namespace {

using namespace gemm;

// fp32 -> fp16 pixels, 8 per thread (two 16-byte loads, one 16-byte store)
__global__ __launch_bounds__(256) void cast_image_kernel(const float* __restrict__ src, half_t* __restrict__ dst, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x4 lo = *reinterpret_cast<const f32x4*>(src + i * 8), hi = *reinterpret_cast<const f32x4*>(src + i * 8 + 4);
  *reinterpret_cast<f16x8*>(dst + i * 8) =
      f16x8{(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3], (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
}

// ---------------------------------------------------------------------------------------------------------------
// ln_pre over every token row, one wave per row (layernorm.hip's arithmetic and reduction order).  Row sources:
//   l == 0            cls + pos[0]                     (clip/model.py:398-401; identical for every image)
//   1 <= l < tokens0  x0[b, l, :] as the GEMM left it (fp16 or fp32) (+ pos[l] when add_pos: clip/model.py:401)
//   l >= tokens0      shallow[l - tokens0, :]          (MaPLe's shallow prompt tokens, clip/model.py:459-460: appended after pos is added)
// Outputs: y (fp32 stream, optional) and / or y16 + stats (fp16 operand copy + LayerNorm-fold row sums of the output, partial 0).
// One row per wave, gamma / beta / pos re-read from L2 per row: measured against several rows per wave with gamma / beta held in registers
// (6 KB less L2 traffic per row): 43.7 us against 58.8 (one row, 32 more registers) and 62.7 us (four rows) -- latency hiding by many small
// waves beats the saved bytes (profiles/r04_patch_embed.txt).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float eln_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename TI, int NV>
__global__ __launch_bounds__(256) void embed_ln_kernel(const TI* __restrict__ x0, const float* __restrict__ cls, const float* __restrict__ pos,
                                                       const float* __restrict__ shallow, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y, half_t* __restrict__ y16,
                                                       float* __restrict__ stats_out, int rows, int L, int tokens0, int D, float eps, int add_pos) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int l = row % L;
  const int nvec = D >> 2;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      if (l == 0) {
        v[i] = *reinterpret_cast<const f32x4*>(cls + c * 4) + *reinterpret_cast<const f32x4*>(pos + c * 4);
      } else if (l >= tokens0) {
        v[i] = *reinterpret_cast<const f32x4*>(shallow + (int64_t)(l - tokens0) * D + c * 4);
      } else {
        if constexpr (sizeof(TI) == 4) {
          v[i] = *reinterpret_cast<const f32x4*>(x0 + (int64_t)row * D + c * 4);
        } else {
          const f16x4 h = *reinterpret_cast<const f16x4*>(x0 + (int64_t)row * D + c * 4);
          v[i] = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
        if (add_pos) v[i] += *reinterpret_cast<const f32x4*>(pos + (int64_t)l * D + c * 4);   // (the GEMM left the bare conv output)
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = eln_wave_sum(s) / (float)D;
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
  const float rstd = rsqrtf(eln_wave_sum(q) / (float)D + eps);
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
      if (y) *reinterpret_cast<f32x4*>(y + (int64_t)row * D + c * 4) = o;
      if (y16) {
        *reinterpret_cast<f16x4*>(y16 + (int64_t)row * D + c * 4) = f16x4{(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
        os += (o[0] + o[1]) + (o[2] + o[3]);
        oq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
      }
    }
  }
  if (stats_out) {
    os = eln_wave_sum(os);
    oq = eln_wave_sum(oq);
    if (lane == 0) *reinterpret_cast<float2*>(stats_out + 2 * (int64_t)row) = make_float2(os, oq);
  }
}

}  // namespace

bool patch_embed_fits(int B, int R, int P, int D) {
  const int64_t image_bytes = (int64_t)B * 3 * R * R * 2;   // as fp16
  return (P == 8 || P == 16 || P == 32) && R % P == 0 && D % 8 == 0 && image_bytes < 0x7FFFFF00ll;
}

size_t patch_embed_scratch_bytes(int B, int R, int image_dtype) { return image_dtype == CLIPMI_F32 ? align256((size_t)B * 3 * R * R * 2) : 0; }

int launch_patch_embed(const void* image, int image_dtype, void* scratch, const half_t* conv_w, int64_t ldw, const float* pos, void* x0, int x0_dtype,
                       int B, int R, int P, int D, int tokens, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(image && conv_w && x0, CLIPMI_ERR_ARG, "patch_embed: null pointer");
  CLIPMI_REQUIRE(image_dtype != CLIPMI_F32 || (scratch && (uintptr_t)scratch % 16 == 0), CLIPMI_ERR_ARG,
                 "patch_embed: an fp32 image needs a 16-byte aligned scratch buffer of patch_embed_scratch_bytes for its fp16 copy");
  CLIPMI_REQUIRE(image_dtype == CLIPMI_F16 || image_dtype == CLIPMI_F32, CLIPMI_ERR_ARG, "patch_embed: image dtype %d", image_dtype);
  CLIPMI_REQUIRE(x0_dtype == CLIPMI_F16 || x0_dtype == CLIPMI_F32, CLIPMI_ERR_ARG, "patch_embed: output dtype %d", x0_dtype);
  CLIPMI_REQUIRE(patch_embed_fits(B, R, P, D), CLIPMI_ERR_SHAPE,
                 "patch_embed: needs P in {8, 16, 32}, R %% P == 0, width %% 8 == 0 and an fp16 image batch below 2 GB (B=%d R=%d P=%d D=%d)", B, R, P, D);
  CLIPMI_REQUIRE((uintptr_t)image % 16 == 0 && (uintptr_t)conv_w % 16 == 0 && (uintptr_t)pos % 16 == 0 && (uintptr_t)x0 % 16 == 0,
                 CLIPMI_ERR_ARG, "patch_embed: pointers must be 16-byte aligned");
  CLIPMI_REQUIRE(ldw % 8 == 0 && ldw >= 3 * P * P, CLIPMI_ERR_SHAPE, "patch_embed: weight rows must keep 16-byte alignment");
  const int G = R / P;
  CLIPMI_REQUIRE(tokens >= G * G + 1, CLIPMI_ERR_SHAPE, "patch_embed: tokens=%d < 1 + %d patches", tokens, G * G);
  const half_t* image16 = static_cast<const half_t*>(image);
  if (image_dtype == CLIPMI_F32) {
    const int64_t n8 = (int64_t)B * 3 * R * R / 8;   // R % 8 == 0
    hipLaunchKernelGGL(cast_image_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, static_cast<const float*>(image), static_cast<half_t*>(scratch), n8);
    const int rc = check_launch("cast_image_kernel");
    if (rc) return rc;
    image16 = static_cast<const half_t*>(scratch);
  }
  GemmArgs a{};
  a.A = image16; a.lda = 3 * P * P; a.W = conv_w; a.ldw = ldw; a.out = x0; a.ldo = D; a.out_dtype = x0_dtype;
  a.M = B * G * G; a.N = D; a.K = 3 * P * P; a.epilogue = EPI_PATCH_POS;
  a.pos = pos; a.patches = G * G; a.tokens = tokens; a.im_R = R; a.im_P = P;
  return launch_gemm(a, s);
}

int launch_embed_ln(const void* x0, int x0_dtype, int add_pos, const float* cls, const float* pos, const float* shallow, const float* gamma,
                    const float* beta, float* y, half_t* y16, float* stats_out, int B, int L, int tokens0, int D, float eps, hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x0 && cls && pos && gamma && beta && (y || y16), CLIPMI_ERR_ARG, "embed_ln: null pointer");
  CLIPMI_REQUIRE((!y16 && !stats_out) || (y16 && stats_out), CLIPMI_ERR_ARG, "embed_ln: y16 and stats_out come together");
  CLIPMI_REQUIRE(L >= tokens0 && tokens0 >= 1 && (L == tokens0 || shallow), CLIPMI_ERR_ARG, "embed_ln: L=%d tokens0=%d / shallow prompt missing", L, tokens0);
  CLIPMI_REQUIRE(D > 0 && D % 4 == 0 && D <= 4096, CLIPMI_ERR_SHAPE, "embed_ln: D=%d unsupported (D %% 4 == 0, D <= 4096)", D);
  CLIPMI_REQUIRE((int64_t)B * L < (1ll << 31), CLIPMI_ERR_SHAPE, "embed_ln: too many rows");
  const int rows = B * L;
  const dim3 grid((rows + 3) / 4), block(256);
  const bool wide = D / 4 > 64 * 4;
#define ELN_LAUNCH(TI, NV) \
  hipLaunchKernelGGL((embed_ln_kernel<TI, NV>), grid, block, 0, s, (const TI*)x0, cls, pos, shallow, gamma, beta, y, y16, stats_out, rows, L, tokens0, D, eps, add_pos)
  if (x0_dtype == CLIPMI_F32) { if (wide) ELN_LAUNCH(float, 16); else ELN_LAUNCH(float, 4); }
  else { if (wide) ELN_LAUNCH(half_t, 16); else ELN_LAUNCH(half_t, 4); }
#undef ELN_LAUNCH
  return check_launch("embed_ln_kernel");
}

}  // namespace clipmi
