/*
 * clipmi.h -- C ABI of libclipmi.so: the MI355X (gfx950) CLIP inference-and-calibration hot path.
 *
 * The reference (ml-stat-Sustech/CLIP_Calibration) is pure Python and has no FFI of its own; its "operator
 * API" for this path is the attribute surface of the object returned by clip.build_model (reference
 * clip/model.py:656-699) plus DistanseAwareCalibration.predict and the ECE metric.  Each entry point below
 * names the reference interface it replaces.  Conventions:
 *
 *   - every pointer is a DEVICE pointer owned by the caller (torch-ROCm tensors in the Python host code)
 *     unless the parameter is documented as "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *     never synchronises the device; nothing is allocated or freed inside a launch call (graph-capture safe);
 *   - weights are BORROWED: the handle stores the pointers given to clipmi_set_*_weights, the caller keeps the
 *     tensors alive and re-binds after moving them;
 *   - the return value is CLIPMI_OK (0) or a negative error code; nothing throws across the boundary;
 *     clipmi_last_error() returns a thread-local detail string for the most recent failure;
 *   - activations are token-major ("NLD"): row (n*L + l) of a [N*L, D] matrix.
 *
 * dtype codes: CLIPMI_F16 = IEEE half, CLIPMI_F32 = float.  All GEMMs run on v_mfma_f32_16x16x32_f16 with fp32
 * accumulation; LayerNorm, softmax, residual stream, L2 norms and the final logits are fp32.
 */
#ifndef CLIPMI_H
#define CLIPMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIPMI_ABI_VERSION 13

typedef void* clipmi_stream_t; /* hipStream_t */

enum {
  CLIPMI_OK = 0,
  CLIPMI_ERR_ARG = -1,         /* null pointer / bad enum */
  CLIPMI_ERR_SHAPE = -2,       /* shape the kernels do not support (see each call) */
  CLIPMI_ERR_HIP = -3,         /* a HIP runtime call failed (launch error, no device) */
  CLIPMI_ERR_WORKSPACE = -4,   /* workspace too small */
  CLIPMI_ERR_STATE = -5        /* weights not bound */
};

enum { CLIPMI_F16 = 0, CLIPMI_F32 = 1 };

enum {
  CLIPMI_EPI_NONE = 0,          /* out = acc */
  CLIPMI_EPI_BIAS = 1,          /* out = acc + bias[n] */
  CLIPMI_EPI_BIAS_QUICKGELU = 2,/* t = acc + bias[n]; out = t * sigmoid(1.702 t)   (clip/model.py:162-164) */
  CLIPMI_EPI_BIAS_RESIDUAL = 3, /* out = residual[m,n] + acc + bias[n]             (clip/model.py:186-187) */
  CLIPMI_EPI_BIAS_RELU = 4,     /* out = max(acc + bias[n], 0): conv + folded BatchNorm + ReLU (clip/model.py:45-46,139) */
  CLIPMI_EPI_BIAS_RESIDUAL16_RELU = 5 /* out = max(acc + bias[n] + residual16[m,n], 0): the tail of a Bottleneck
                                         (clip/model.py:48-55); `residual` points at fp16 [M,N] (ld = ldo), fp16 out only */
};

int clipmi_abi_version(void);
const char* clipmi_strerror(int code);
const char* clipmi_last_error(void);

/* Runtime switches, process-wide.  The reference has none (its knobs are yacs config keys read by Dassl); these select
 * between parity-tested implementations of the same operator and exist for tests, A/B measurements and the DEFAULT precision
 * policy.  Each option also has an environment spelling that is read ONCE, at the first launch; after that only
 * clipmi_set_option changes it (no launch path calls getenv, and no launch path writes an option).  Unknown names return
 * CLIPMI_ERR_ARG.
 *   gemm_variant     (CLIPMI_GEMM_VARIANT)    -1 = default dispatch; 0 (128 x 128 tiles), 1 (256 x 256, 16 waves), 10 / 'a' (320 x 256
 *                                             ping-pong), 13 / 's' (persistent, streamed fp16 epilogue), 16 / 'r' (persistent row ranges,
 *                                             streamed residual epilogue) force one kernel family where the shape allows it (test aid)
 *   gemm_band        (CLIPMI_GEMM_BAND)       0 = default; n-tiles per traversal band
 *   gemm_stream      (CLIPMI_GEMM_STREAM)     1 (default) = ping-pong persistent kernel with streamed epilogue for multi-round fp16-out
 *                                             GEMMs, K >= 512 (in-proj / c_fc); 0 = one tile per workgroup (same bits with the bias epilogue)
 *   gemm_rstream     (CLIPMI_GEMM_RSTREAM)    1 (default) = persistent row-range kernel with streamed residual epilogue for the fp16-stream
 *                                             residual GEMMs with K <= 1536 (out-proj); 0 = one 320 x 256 tile per workgroup.  The row-range
 *                                             kernel adds the residual inside its K loop: both round the same fp32 sum once, in another order
 *   gemm_split_rows  (CLIPMI_GEMM_SPLIT_ROWS) 1 (default) = where the ragged last row of 256-row tiles (M % 256 <= 128 rows) would open a round of
 *                                             its own in the persistent kernel (ViT-L/14@336 at 64 images: c_fc 9.06 rounds), those rows go to
 *                                             the tile kernels as a second launch; 0 = one launch
 *   attn_loader      (CLIPMI_ATTN_LOADER)     2 (default) = 193..200-token non-causal attention runs the kernel whose operands all arrive
 *                                             by LDS-DMA from a dedicated loader wave and whose output rows are stored non-temporal;
 *                                             1 = the same with plain stores; 0 = the persistent kernel (all three: same bits)
 *   attn_ring        (CLIPMI_ATTN_RING)       1 (default) = non-causal attention over more than 224 tokens (ViT-L/14: 257, 577) runs the ring kernel
 *                                             (persistent workgroups, loader wave, 128-key blocks through a three-slot LDS ring); 0 = the round-1
 *                                             streaming kernel (also taken for a causal mask of that length)
 *   attn_small       (CLIPMI_ATTN_SMALL)      1 (default) = attention over at most 32 tokens (the text tower after dead-row elimination) gives every
 *                                             (sequence, head) item to ONE wave, no workgroup barrier; 0 = the persistent kernel (same bits)
 *   tail_unfused     (CLIPMI_TAIL_UNFUSED)    1 = clipmi_logits / clipmi_fused_tail as separate launches instead of the fused tail kernel
 *   vision_pass      (CLIPMI_VISION_PASS)     stream elements (token rows x width) of one pass of clipmi_encode_image, default 50432 * 768
 *                                             (256 images of ViT-B/16, 128 of ViT-L/14, 64 of ViT-L/14@336): a batch of one and a half
 *                                             passes or more runs as consecutive passes on the same stream and workspace -- each pass is an ordinary
 *                                             call on its images: features equal to the one-pass result up to the library's usual batch
 *                                             dependence (tile and kernel choice follow the row count: <= 2e-4 on normalised features),
 *                                             +8..10 % images/s at 512-1024 images of ViT-B/16; 0 = never split
 * and the process-wide DEFAULTS of the three per-model settings (clipmi_model_set_option overrides them per handle):
 *   cls_only_last_block (CLIPMI_CLS_ONLY_LAST_BLOCK)  1 (default since round 6) = the image tower's LAST block computes what the class row
 *                    needs and nothing else -- K | V of every token, Q / attention / out-proj / ln_2 / c_fc / c_proj of the class rows, the only
 *                    rows ln_post reads (clip/model.py:419) -- as GEMMs with M = batch and row stride L * D and a one-query attention kernel;
 *                    same per-element arithmetic, features equal to the every-row computation to ~1e-6 in cosine, 7 % fewer flops per image of
 *                    ViT-B/16; 0 = every block computes every token row (what bench.py's headline `value` times)
 *   ln_fold          (CLIPMI_LN_FOLD)         1 = ln_1 / ln_2 inside the GEMM epilogues (default), 0 = LayerNorm kernels
 *   residual_f16     (CLIPMI_RESIDUAL_F16)    0 = fp32 residual stream, 1 = fp16 on both towers, 2 = image tower only
 *                                             (default; env 'v'), 3 = text tower only (env 't') */
int clipmi_set_option(const char* name, int value);
int clipmi_get_option(const char* name, int* value);

/* ------------------------------------------------------------------------------------------------------
 * Operator level (stateless).  These are the kernels; the tower drivers below are sequences of them.
 * ---------------------------------------------------------------------------------------------------- */

/* nn.Linear / in_proj / out_proj / c_fc / c_proj / `x @ proj` (clip/model.py:174-176,183,422,611):
 * out[M,N] = epilogue(A[M,K] @ W[N,K]^T).  A, W fp16 row-major with leading dimensions lda, ldw (elements);
 * bias fp32[N] or NULL; residual fp32 [M,N] (fp16 for BIAS_RESIDUAL16_RELU; ld = ldo) or NULL; out fp16 or fp32 (out_dtype), ld = ldo.
 * Requires K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0; M, N otherwise arbitrary. */
int clipmi_gemm_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                    const void* residual, void* out, int64_t ldo, int out_dtype,
                    int M, int N, int K, int epilogue, clipmi_stream_t stream);

/* The residual GEMM of a block on the fp16 residual stream (out-proj / c_proj, clip/model.py:186-187: `x = x + f(x)` on fp16 tensors):
 * x16[m,n] = fp16(x16[m,n] + A[m,:] . W[n,:] + bias[n]) IN PLACE (one rounding of the fp32 sum), plus the LayerNorm-fold row
 * partials the next GEMM consumes: stats[(t * M + m) * 2 + {0,1}] = (sum, sum of squares) over the 256 columns of column tile t of
 * the ROUNDED row m; *parts (host int) receives the number of column tiles of the kernel that ran: (N + 255) / 256 with every default
 * dispatch, (N + 127) / 128 when gemm_variant = 0 is forced (partials are then per 128 columns); at most 8 either way.  A fp16 [M,K]
 * (lda), W fp16 [N,K] (ldw), bias fp32 [N], x16 fp16 [M,N] (ldx), stats fp32 [8 * M * 2] (room for the largest count).  K % 64 == 0, N % 8 == 0, ldx % 8 == 0, 16-byte aligned. */
int clipmi_gemm_residual_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* x16, int64_t ldx,
                             float* stats, int* parts, int M, int N, int K, clipmi_stream_t stream);

/* LayerNorm subclass with fp32 statistics (clip/model.py:153-159): y[r,:] = (x[row(r),:] - mean) * rsqrt(var+eps)
 * * gamma + beta.  row(r) = gather_idx ? gather_idx[r] : r, addressed with in_stride (elements).  D % 4 == 0,
 * D <= 4096. */
int clipmi_layernorm(const void* x, int x_dtype, int64_t in_stride, const int32_t* gather_idx,
                     const float* gamma, const float* beta, void* y, int y_dtype, int64_t out_stride,
                     int rows, int D, float eps, clipmi_stream_t stream);

/* nn.MultiheadAttention core after the packed in-projection (clip/model.py:181-183; SURVEY a-5a):
 * qkv fp16 [N*L, 3*D] (q | k | v, head h at columns h*64..h*64+63 of each third), out fp16 [N*L, D] =
 * merge_heads(softmax(q k^T / 8 + mask) v).  head_dim is 64 (D == 64*H).  causal != 0 applies the text
 * tower's mask (clip/model.py:585-591). */
int clipmi_attention(const void* qkv, void* out, int N, int L, int H, int causal, clipmi_stream_t stream);

/* image.type(dtype) + the im2col half of conv1 (clip/model.py:598,395-397): image [B,3,R,R] (fp32 or fp16,
 * NCHW) -> col fp16 [B*(R/P)^2, Kpad], column c*P*P + ky*P + kx, zero padded up to Kpad (a multiple of 64). */
int clipmi_patchify(const void* image, int image_dtype, void* col, int B, int R, int P, int Kpad,
                    clipmi_stream_t stream);

/* image.type(dtype) + conv1 + reshape / permute + positional embedding of the PATCH rows as a GEMM whose loader reads the image itself
 * (clip/model.py:598,395-397,401; conv1 has stride = kernel = P, so its im2col matrix is an address map of the NCHW image):
 *   x0[b * tokens + 1 + p, :] = sum_k pixel(b, p, k) * conv_w[:, k] + pos[1 + p, :]        p = py * (R/P) + px,  k = c*P*P + ky*P + kx
 * image [B,3,R,R] fp32 or fp16; an fp32 image is first cast to fp16 into `scratch` (clipmi_patch_embed_scratch_bytes bytes, 16-byte aligned;
 * NULL for an fp16 image); conv_w fp16 [D, ldw] = conv1.weight.reshape(D, 3*P*P) (ldw >= 3*P*P); pos fp32 [1 + (R/P)^2, D] or NULL (the bare conv
 * output: clipmi_embed_ln then adds pos -- what clipmi_encode_image does, the positional rows cost this GEMM's epilogue 13 us); x0 fp16 or fp32
 * (x0_dtype) [B * tokens, D]: only the patch rows are written -- row 0 (class token) and rows beyond 1 + (R/P)^2 (prompt tokens) of every
 * sequence are left to clipmi_embed_ln.  fp16 operands, fp32 accumulation, pos added in fp32, one rounding.  Requires P in {8, 16, 32},
 * R % P == 0, D % 8 == 0, an fp16 image batch below 2 GB; 16-byte aligned pointers.  CLIPMI_ERR_SHAPE otherwise (clipmi_encode_image then
 * takes clipmi_patchify + clipmi_gemm_f16: ViT-L/14). */
size_t clipmi_patch_embed_scratch_bytes(int B, int R, int image_dtype);
int clipmi_patch_embed(const void* image, int image_dtype, void* scratch, const void* conv_w, int64_t ldw, const float* pos, void* x0,
                       int x0_dtype, int B, int R, int P, int D, int tokens, clipmi_stream_t stream);

/* cat(class_embedding) + positional embedding of the class row + ln_pre over every token row (clip/model.py:398-402,413; MaPLe's shallow
 * prompt rows, :459-460), one wave per row:  row (b, l) = l == 0 ? cls + pos[0] : l < tokens0 ? x0[b * L + l] (+ pos[l] if add_pos) : shallow[l - tokens0];
 * out = LayerNorm(row) with fp32 statistics.  x0 fp16 / fp32 [B * L, D] as clipmi_patch_embed left it; cls fp32 [D]; pos fp32 [tokens0, D];
 * shallow fp32 [L - tokens0, D] or NULL when L == tokens0; y fp32 [B * L, D] or NULL; y16 fp16 [B * L, D] + stats fp32 [B * L * 2]
 * ((sum, sum of squares) of each output row: the LayerNorm-fold partial the first in-projection consumes) or both NULL. */
int clipmi_embed_ln(const void* x0, int x0_dtype, int add_pos, const float* cls, const float* pos, const float* shallow,
                    const float* gamma, const float* beta, float* y, void* y16, float* stats, int B, int L, int tokens0, int D, float eps,
                    clipmi_stream_t stream);

/* Row L2 normalisation  f / ||f||  (zsclip.py:99; coop.py:212-213): in (fp16|fp32) [rows,E] -> out fp32. */
int clipmi_l2_normalize(const void* in, int in_dtype, float* out, int rows, int E, clipmi_stream_t stream);

/* The same with a choice of output type: out fp32 or fp16 [rows,E] (out_dtype).  fp16 is the exchange format of the
 * multi-GPU path: the fp32 quotient rounded once (what the reference's fp16 GPU path holds after zsclip.py:99). */
int clipmi_l2_normalize_to(const void* in, int in_dtype, void* out, int out_dtype, int rows, int E, clipmi_stream_t stream);

/* Fused  logits = (scale * img_n) @ txt_n^T  (zsclip.py:100-101, coop.py:215-217, tempscaling.py:53-56)
 * + DistanseAwareCalibration.predict (distanse_aware_calibration.py:49-58: logits[i,:] *= conf[argmax_i])
 * + softmax top-1 (vl_calibrator.py:91, vl_evaluator.py:68,83): conf[i] = max_c softmax(logits[i,:]),
 * pred[i] = argmax.  img_n [B,E], txt_n [C,E] fp32, ALREADY L2-normalised.  dac_conf fp32[C] or NULL.
 * logits fp32 [B,C] (required), conf fp32[B], pred int32[B] (either may be NULL).  E % 16 == 0. */
int clipmi_logits(const float* img_n, const float* txt_n, float scale, const float* dac_conf,
                  float* logits, float* conf, int32_t* pred, int B, int C, int E, clipmi_stream_t stream);

/* The tail of the path as ONE launch (the fused normalise + matmul + DAC-temperature kernel):
 *   img_n = img / ||img||                      zsclip.py:99, coop.py:212-213            (normalize != 0; else img is used as given)
 *   logits = scale * img_n @ txt_n^T           zsclip.py:100-101, coop.py:215-217, tempscaling.py:53-56
 *   pred = argmax_c; logits[i,:] *= dac[pred]  distanse_aware_calibration.py:49-58      (dac_conf != NULL)
 *   conf = max_c softmax(logits[i,:])          vl_calibrator.py:91, vl_evaluator.py:68,83
 *   ECE bins += (1, conf, pred == label)       tools/metrics.py:90-130                  (bins != NULL; layout of clipmi_ece_accumulate)
 * img fp32 or fp16 [B,E] (img_dtype; un-normalised tower output when normalize != 0, e.g. the fp16 embeddings gathered from
 * all ranks when normalize == 0), txt_n fp32 [C,E] L2-normalised; logits fp32 [B,C] required;
 * img_n_out fp32 [B,E] (the normalised image features of the reference's 3-tuple; NULL to skip; only with normalize), conf fp32 [B],
 * pred int32 [B], labels int64 [B] + bins float64 [3*(n_bins+1)]: each may be NULL.  Results are bit-identical to
 * clipmi_l2_normalize + clipmi_logits + clipmi_ece_accumulate (the ECE sums of confidences up to the order of their atomics).  E % 64 == 0 and E <= 2048 run fused; other shapes, and option
 * tail_unfused = 1, run the separate launches (which need img_n_out or fp32 normalised input, and conf + pred when bins are given).
 * workspace: clipmi_fused_tail_workspace_bytes(B, C) bytes of device memory whose first 64 KiB (ticket counters, one int32 per 16- or
 * 32-row block) are ZERO before the first launch; every launch leaves them zero again.  The rest holds one 16-byte partial (max, argmax,
 * sum of exponentials) per (row, 64-column block), written and consumed inside a launch: no initialisation.  One workspace serves ONE
 * launch at a time (launches on one stream are fine; concurrent launches on different streams need a workspace each).  Re-zero the counters
 * after a launch that failed.  Features must be finite with |x| <= 65504 (L2-normalised ones are <= 1): the products run on the
 * fp16 matrix cores with every fp32 operand split into fp16 hi + lo halves, ~1e-6 absolute on logits of scale 100. */
size_t clipmi_fused_tail_workspace_bytes(int B, int C);
int clipmi_fused_tail(const void* img, int img_dtype, int normalize, const float* txt_n, float scale, const float* dac_conf, float* logits,
                      float* img_n_out, float* conf, int32_t* pred, const int64_t* labels, double* bins, int n_bins,
                      void* workspace, size_t workspace_bytes, int B, int C, int E, clipmi_stream_t stream);

/* The row pass of clipmi_logits alone, on logits that already exist -- DistanseAwareCalibration.predict
 * (distanse_aware_calibration.py:49-58) + softmax top-1: pred = argmax_c logits[i,:]; if dac_conf != NULL the row is
 * multiplied in place by dac_conf[pred]; conf[i] = max_c softmax(row).  conf / pred may be NULL. */
int clipmi_calibrate_rows(float* logits, const float* dac_conf, float* conf, int32_t* pred, int B, int C,
                          clipmi_stream_t stream);

/* VLCalibration.predict on its DAC / plain branches (trainers/calibration/vl_calibrator.py:83-109): probs[i,:] =
 * softmax(logits[i,:] * (dac_conf ? dac_conf[argmax_i] : 1)), the full probability matrix the reference hands to
 * VLClassification.evaluate (vl_evaluator.py:59).  logits fp32 [B,C] are NOT modified; probs fp32 [B,C] (required, may
 * alias logits); conf / pred as above, may be NULL. */
int clipmi_softmax_rows(const float* logits, const float* dac_conf, float* probs, float* conf, int32_t* pred,
                        int B, int C, clipmi_stream_t stream);

/* ModifiedResNet image tower (clip/model.py:10-150) -- SURVEY f-4.  Activations NHWC fp16; a 1x1 convolution is
 * clipmi_gemm_f16 on the [B*H*W, C] rows with the folded BatchNorm as bias (CLIPMI_EPI_BIAS_RELU / _RESIDUAL16_RELU).
 *  clipmi_im2col3x3_nchw   stem conv1 (clip/model.py:106, stride 2, pad 1) from the NCHW image (fp32|fp16):
 *                          col fp16 [B*Ho*Wo, Kpad], column c*9 + ky*3 + kx, zero padded (Kpad % 64 == 0)
 *  clipmi_im2col3x3_nhwc   every other 3x3 convolution (stride 1, pad 1; clip/model.py:20,108,110):
 *                          col[(b,y,x), (ky*3+kx)*C + c] = x[b, y+ky-1, x+kx-1, c]; C % 8 == 0
 *  clipmi_avgpool_nhwc     nn.AvgPool2d(k) (clip/model.py:23,33,112): x [B,H,W,C] -> y [B,H/k,W/k,C]
 *  clipmi_attnpool_tokens  AttentionPool2d token build (clip/model.py:69-71): tokens fp16 [B, HW+1, C] = [mean | x] + pos (fp32 [HW+1, C])
 *  clipmi_attnpool         its attention with the mean token as the only query (clip/model.py:72-90): q fp16 [B,C] (projected,
 *                          biased), kv fp16 [B*T, 2C] (k | v projected, biased), head_dim 64 -> out fp16 [B, C] (before c_proj)
 *  clipmi_conv3x3_nhwc     Bottleneck.conv2 + bn2 (+ ReLU) as an IMPLICIT GEMM (no im2col matrix): x fp16 [B,H,W,C], w fp16
 *                          [Cout, 9*C] tap-major ((ky*3+kx)*C + c, BatchNorm folded), bias fp32 [Cout] -> out fp16 [B,H,W,Cout];
 *                          stride 1, pad 1, C % 64 == 0, Cout % 8 == 0 */
int clipmi_conv3x3_nhwc(const void* x, const void* w, const float* bias, void* out, int B, int H, int W, int C, int Cout,
                        int relu, clipmi_stream_t stream);
int clipmi_im2col3x3_nchw(const void* image, int image_dtype, void* col, int B, int Cin, int H, int W, int stride, int Kpad,
                          clipmi_stream_t stream);
int clipmi_im2col3x3_nhwc(const void* x, void* col, int B, int H, int W, int C, int Kpad, clipmi_stream_t stream);
int clipmi_avgpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int k, clipmi_stream_t stream);
int clipmi_attnpool_tokens(const void* x, const float* pos, void* tokens, int B, int HW, int C, clipmi_stream_t stream);
int clipmi_attnpool(const void* q, const void* kv, void* out, int B, int T, int heads, clipmi_stream_t stream);

/* CLIP-Adapter's feature blend (trainers/classification/clip_adapter.py:138-172): out[b,:] = ratio * relu(W2 relu(W1 f[b,:]))
 * + (1 - ratio) * f[b,:]; feats [B,E] (un-normalised image features), w1 [H,E], w2 [E,H], no biases; all fp32. */
int clipmi_adapter_blend(const float* feats, const float* w1, const float* w2, float ratio, float* out, int B, int E,
                         int H, clipmi_stream_t stream);

/* TaskRes' classifier (trainers/classification/taskres.py:105-106): out = a + alpha * b over n fp32 elements
 * (a = base text features, b = learned residual).  out may alias a. */
int clipmi_scale_add(const float* a, const float* b, float alpha, float* out, long long n, clipmi_stream_t stream);

/* ProDA's classifier (trainers/classification/proda.py:316-333): out[g,:] = mean over the P prompts of class g of the
 * (already L2-normalised) text features in [(g*P + p), :].  fp32; the mean is NOT re-normalised, as in the reference. */
int clipmi_group_mean(const float* in, float* out, int G, int P, int E, clipmi_stream_t stream);

/* CoCoOp (trainers/classification/cocoop.py:154-199) -- SURVEY f-4: instance-conditioned prompts.  All fp32 unless noted.
 *  clipmi_cocoop_ctx       PromptLearner.forward's meta-net and shift (:154-161): ctx_shifted[b,t,:] = ctx[t,:] +
 *                          W2 relu(W1 img_n[b] + b1) + b2.  img_n [B,E] (L2-normalised image features), w1 [H,E], b1 [H],
 *                          w2 [D,H], b2 [D], ctx [n_ctx,D] -> ctx_shifted [B,n_ctx,D].
 *  clipmi_cocoop_prompts   construct_prompts for `n_images` images x C classes (:163-171): prompts[(b,c),l,:] =
 *                          ctx_shifted[b,l-1,:] for 1 <= l <= n_ctx, else base[c,l,:] (base = token embedding of the
 *                          "X X .. X name." prompts, [C,L,D] fp16|fp32).  prompts fp16 [n_images*C, L, D], the input of
 *                          clipmi_text_encoder.  D % 8 == 0.
 *  clipmi_logits_per_image the loop body of CustomCLIP.forward (:193-199): logits[b,c] = scale * <img_n[b],
 *                          txt[b,c,:] / ||txt[b,c,:]||>, txt [B,C,E] UN-normalised text-encoder outputs; then the same
 *                          DAC / softmax top-1 row pass as clipmi_logits.  txt_n_last [C,E] (may be NULL) receives the
 *                          normalised text features of the last image -- what the reference's 3-tuple carries. */
int clipmi_cocoop_ctx(const float* img_n, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* ctx, float* ctx_shifted, int B, int E, int H, int D, int n_ctx, clipmi_stream_t stream);
int clipmi_cocoop_prompts(const void* base, int base_dtype, const float* ctx_shifted, void* prompts, int n_images,
                          int C, int L, int D, int n_ctx, clipmi_stream_t stream);
int clipmi_logits_per_image(const float* img_n, const float* txt, float scale, const float* dac_conf, float* logits,
                            float* conf, int32_t* pred, float* txt_n_last, int B, int C, int E, clipmi_stream_t stream);

/* Device-side accumulation of the ECE statistics (tools/metrics.py:90-130) -- SURVEY f-1.  bins: float64
 * [3*(n_bins+1)] = per-bin (count, sum_conf, sum_correct), bin n_bins collects conf == 1.0 (the digitize
 * quirk).  Accumulates over calls; zero it with hipMemsetAsync.  The final reduction to a scalar is host
 * code (clip_calibration_amd.metrics.ece_from_bins). */
int clipmi_ece_accumulate(const float* conf, const int32_t* pred, const int64_t* labels, int n,
                          double* bins, int n_bins, clipmi_stream_t stream);

/* get_knn_dists / get_val_image_knn_dists (trainers/calibration/proximity.py:19-70) -- SURVEY f-2: for every query row
 * the K smallest L2 distances ||refs[j] - queries[i]||_2, ascending, out fp32 [Nq, K].  queries [Nq,E], refs [Nr,E]
 * fp32; E % 64 == 0; 1 <= K <= min(16, Nr).  (The "val image" variant asks for K+1 against itself and drops column 0.) */
int clipmi_knn_dists(const float* queries, const float* refs, float* out, int Nq, int Nr, int E, int K,
                     clipmi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Multi-GPU exchange: one process per GPU, the image batch sharded over the ranks, weights and text
 * features replicated and resident; per step ONE all-gather of the per-GPU L2-normalised image embeddings (fp16 [B/G,E])
 * before the shared logits kernel.  Replaces the reference's nn.DataParallel wrapping, which re-broadcasts the weights on
 * every call (trainers/classification/coop.py:266-272, trainers/calibration/tempscaling.py:117-120).  RCCL (librccl,
 * opened on first use) over xGMI; gather, not reduce, so results do not depend on the rank count.
 *   clipmi_comm_unique_id  rank 0: fills id_out (host, CLIPMI_COMM_ID_BYTES) -- hand it to every rank out of band
 *   clipmi_comm_create     every rank, collectively, AFTER selecting its GPU (hipSetDevice / torch.cuda.set_device)
 *   clipmi_comm_ranks      what the communicator itself reports (world size, this rank)
 *   clipmi_allgather       out[r*bytes .. (r+1)*bytes) = rank r's `in`, on every rank; asynchronous on `stream`
 * ---------------------------------------------------------------------------------------------------- */
#define CLIPMI_COMM_ID_BYTES 128
typedef struct clipmi_comm clipmi_comm;
int clipmi_comm_unique_id(void* id_out);
int clipmi_comm_create(const void* id, int world, int rank, clipmi_comm** out);
int clipmi_comm_destroy(clipmi_comm* comm);
int clipmi_comm_ranks(const clipmi_comm* comm, int* world, int* rank);
int clipmi_allgather(clipmi_comm* comm, const void* in, void* out, size_t bytes_per_rank, clipmi_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Model level.  One handle per CLIP model per GPU.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct clipmi_model clipmi_model;

typedef struct clipmi_geometry {
  int32_t embed_dim;        /* E */
  int32_t image_resolution; /* R */
  int32_t patch_size;       /* P */
  int32_t vision_width;     /* Dv (multiple of 64) */
  int32_t vision_layers;
  int32_t context_length;   /* 77 */
  int32_t vocab_size;
  int32_t text_width;       /* Dt */
  int32_t text_layers;
  int32_t text_heads;       /* Dt / 64 */
} clipmi_geometry;

/* ResidualAttentionBlock parameters (clip/model.py:167-188).  GEMM weights fp16 [out,in] row-major exactly as in
 * the checkpoint; biases and LayerNorm parameters fp32. */
typedef struct clipmi_block_weights {
  const float* ln1_g; const float* ln1_b;
  const void* w_qkv;  const float* b_qkv;   /* attn.in_proj_weight [3D,D], in_proj_bias [3D] */
  const void* w_out;  const float* b_out;   /* attn.out_proj [D,D] */
  const float* ln2_g; const float* ln2_b;
  const void* w_fc;   const float* b_fc;    /* mlp.c_fc [4D,D] */
  const void* w_proj; const float* b_proj;  /* mlp.c_proj [D,4D] */
  /* Optional LayerNorm-folded operands (all six or none; NULL = run ln_1 / ln_2 as separate kernels):
   *   w_*_f = fp16(gamma * W) row-wise over the input dim, g_*[n] = sum_k float(w_*_f[n,k]),
   *   c_*[n] = sum_k beta[k] * W[n,k] + b[n];  LN(x) W^T + b = rstd*(x w_f^T) - rstd*mean*g + c. */
  const void* w_qkv_f; const float* g_qkv; const float* c_qkv;   /* ln_1 folded into attn.in_proj */
  const void* w_fc_f;  const float* g_fc;  const float* c_fc;    /* ln_2 folded into mlp.c_fc */
} clipmi_block_weights;

typedef struct clipmi_vision_weights {
  const void* conv_w;                /* visual.conv1.weight as fp16 [Dv, Kpad], Kpad = roundup(3*P*P, 64), zero padded */
  const float* class_embedding;      /* [Dv] */
  const float* positional_embedding; /* [(R/P)^2 + 1, Dv] */
  const float* ln_pre_g; const float* ln_pre_b;
  const float* ln_post_g; const float* ln_post_b;
  const void* proj_t;                /* visual.proj^T as fp16 [E, Dv] */
  const clipmi_block_weights* blocks;/* host array [vision_layers] (copied by the call) */
} clipmi_vision_weights;

typedef struct clipmi_text_weights {
  const float* token_embedding;      /* [vocab, Dt] fp32 */
  const float* positional_embedding; /* [context_length, Dt] */
  const float* ln_final_g; const float* ln_final_b;
  const void* proj_t;                /* text_projection^T as fp16 [E, Dt] */
  const clipmi_block_weights* blocks;/* host array [text_layers] (copied by the call) */
} clipmi_text_weights;

/* MaPLe prompt injection (clip/model.py:287-331,447-478; maple.py:170-187).  `shallow` (vision only) is
 * appended after the positional embedding; deep[i] overwrites, before block i+1, the LAST n_ctx tokens (vision)
 * or tokens 1..n_ctx (text).  fp32, already rounded through fp16 by the caller (the reference's .half()). */
typedef struct clipmi_prompt_hook {
  int32_t n_ctx;
  int32_t n_deep;
  const float* shallow;  /* [n_ctx, D] or NULL */
  const float* deep;     /* [n_deep, n_ctx, D] or NULL */
} clipmi_prompt_hook;

int clipmi_create(const clipmi_geometry* geom, clipmi_model** out);
int clipmi_destroy(clipmi_model* m);

/* Per-model settings.  The reference chooses precision per model (cfg.TRAINER.<X>.PREC, trainers/classification/coop.py:243-245) and
 * builds a second CLIP inside the same process (trainers/classification/base_learner.py:262-272), so these live on the handle:
 * names "residual_f16" (0..3, see clipmi_set_option), "ln_fold" (0 / 1), "cls_only_last_block" (0 / 1); value -1 (the initial
 * state) follows the process-wide option of the same name.  clipmi_model_get_option returns the EFFECTIVE value.
 * Threading contract: calls on DIFFERENT handles may run concurrently from different threads and streams (each call touches only
 * its handle, its workspace and its stream); calls on one handle, and clipmi_model_set_option on it, are serialised by the caller. */
int clipmi_model_set_option(clipmi_model* m, const char* name, int value);
int clipmi_model_get_option(const clipmi_model* m, const char* name, int* value);

/* Per-call flags of the tower calls (last argument before the stream): 0, or ONE of the two stream-precision overrides for THIS
 * call only (no state is changed): CoCoOp's per-image text passes run the fp16 stream while the same handle's zero-shot features
 * keep the fp32 stream.  CLIPMI_CALL_STREAM_F16 needs the LayerNorm-folded operands (else CLIPMI_ERR_STATE). */
#define CLIPMI_CALL_DEFAULT 0u
#define CLIPMI_CALL_STREAM_F32 1u /* residual stream in fp32 (+ fp16 operand shadow) */
#define CLIPMI_CALL_STREAM_F16 2u /* residual stream in fp16: the reference's own GPU precision (clip/model.py:186-187) */
int clipmi_set_vision_weights(clipmi_model* m, const clipmi_vision_weights* w);
int clipmi_set_text_weights(clipmi_model* m, const clipmi_text_weights* w);

/* Bytes of scratch the caller must pass to the tower calls for `batch` images / `n_prompts` prompts.  clipmi_encode_image works a large
 * batch in passes (option vision_pass): the vision figure is that of the largest pass, under the option's value at the time of the call to
 * clipmi_encode_image -- size the workspace after any clipmi_set_option("vision_pass", ...).  The text figure is for a call with the
 * same `seq_rows` (below; 0 = the whole context). */
size_t clipmi_vision_workspace_bytes(const clipmi_model* m, int batch, int n_ctx);
size_t clipmi_text_workspace_bytes(const clipmi_model* m, int n_prompts, int seq_rows);

/* CLIP.encode_image / VisionTransformer.forward (clip/model.py:597-598,394-424; MaPLe :447-478 when hook != NULL):
 * image [B,3,R,R] (fp32|fp16) -> out fp32 [B,E] (un-normalised, as the reference returns).  Batches of one and a half passes or more
 * (option vision_pass) run as consecutive passes on `stream`, each an ordinary call on its images (an image's features depend on its batch only through the
 * tile and kernel choice, which follows the row count: <= 2e-4 on normalised features, as between any two batch sizes). */
int clipmi_encode_image(clipmi_model* m, const void* image, int image_dtype, int batch,
                        const clipmi_prompt_hook* hook, float* out, void* workspace, size_t workspace_bytes,
                        unsigned flags, clipmi_stream_t stream);

/* `clip_model.transformer(x)` as the trainers' TextEncoder uses it (coop.py:58-60, maple.py:64-66): the causal
 * blocks only.  x fp16|fp32 [C, L, Dt] token-major in, y same dtype/shape out (x == y allowed when seq_rows covers the context).
 * seq_rows (see clipmi_text_encoder below; 0 = every row): an OPT-IN of the caller, who alone knows where its prompts end -- the blocks return
 * every row, so by default they run every row.  With 0 < seq_rows < L only the first seq_rows token rows of every sequence are read and
 * computed; y receives them in place and ZEROS in the rows behind (Python: `clip_model.transformer.live_rows = ...`, INTEGRATION.md Level 1). */
int clipmi_text_blocks(clipmi_model* m, const void* x, void* y, int dtype, int n_prompts, int seq_rows,
                       const clipmi_prompt_hook* hook, void* workspace, size_t workspace_bytes,
                       unsigned flags, clipmi_stream_t stream);

/* TextEncoder.forward (coop.py:56-67, maple.py:60-74): prompts (fp16|fp32) [C,L,Dt] WITHOUT positional embedding,
 * eot int32[C] = tokenized_prompts.argmax(-1)  ->  out fp32 [C,E] = ln_final(blocks(prompts + pos))[eot] @ text_projection.
 *
 * seq_rows -- dead-row elimination.  The text blocks mask causally (clip/model.py:585-591: token l attends to tokens <= l only) and the
 * one row that leaves the tower is the EOT row (clip/model.py:611, coop.py:65), so no token behind the LAST prompt's EOT can influence any
 * output.  With 0 < seq_rows < L the tower embeds, runs and reads only the first `seq_rows` token positions of every prompt (the inputs keep
 * their [C,L,..] layout; rows >= seq_rows are never read): identical features, L / seq_rows times fewer rows through every GEMM
 * ("X X X X a photo of a <name>." ends at token ~22 of 77).  The CALLER guarantees max(eot) < seq_rows -- it holds the tokenised prompts on the
 * host side and computes the bound once, without a per-call device sync -- and 1 + hook->n_ctx <= seq_rows; an EOT index outside is clamped
 * to seq_rows - 1 as it is to L - 1 today.  A VIOLATION IS NOT DETECTED: with a bound that is too small the call returns CLIPMI_OK and the
 * features of token row seq_rows - 1 for every prompt whose EOT lies behind it (the EOT indices live on the device; checking them would cost the
 * sync this argument exists to avoid).  Callers that cannot vouch for the bound pass 0.  seq_rows <= 0 or >= L: the whole context. */
int clipmi_text_encoder(clipmi_model* m, const void* prompts, int dtype, const int32_t* eot, int n_prompts, int seq_rows,
                        const clipmi_prompt_hook* hook, float* out, void* workspace, size_t workspace_bytes,
                        unsigned flags, clipmi_stream_t stream);

/* CLIP.encode_text (clip/model.py:600-613): ids int64 [C,L] -> out fp32 [C,E]; EOT row = argmax(ids), computed on device.
 * seq_rows as above (the argmax always scans all L ids). */
int clipmi_encode_text(clipmi_model* m, const int64_t* ids, int n_prompts, int seq_rows, float* out, void* workspace,
                       size_t workspace_bytes, unsigned flags, clipmi_stream_t stream);

/* Timing aid for bench.py (the per-kernel roofline of its JSON line): the five launches of the vision tower's residual
 * block 0 -- 0 in-proj, 1 attention, 2 out-proj + residual, 3 c_fc + QuickGELU, 4 c_proj + residual (clip/model.py:181-188)
 * -- issued exactly as clipmi_encode_image issues them (LayerNorm fold, fp16 stream, tile selection) on the operands the
 * workspace holds (call clipmi_encode_image on `batch` images first), each `iters` times back to back after one untimed launch,
 * the `iters` launches bracketed by ONE pair of hipEvents on `stream` (the GPU stays under load for the whole measurement).
 * ms_out: host float[5], mean milliseconds per launch.  only = -1 times all five, 0..4 just that one (the others report 0).
 * Synchronises the stream; overwrites the activations in the workspace (the residual steps accumulate `iters` + 1 times). */
int clipmi_profile_block(clipmi_model* m, int batch, int iters, int only, void* workspace, size_t workspace_bytes,
                         float* ms_out, clipmi_stream_t stream);

/* Timing aid for bench.py (`roofline.kernels[].us_in_tower`): ONE real pass of clipmi_encode_image on `batch` images (no prompt hook;
 * the batch must be a single pass of the tower, option vision_pass) with a hipEvent recorded on `stream` behind every launch, so that
 * each kernel is timed IN PLACE -- behind its real predecessor, on operands that predecessor has just written -- instead of back to
 * back with itself as clipmi_profile_block does.  us_out (host float[n_us]) receives the microseconds between consecutive events:
 *   [0 .. *n_pre_out)                       the embedding launches in front of the first block (today: patch embedding, ln_pre)
 *   [*n_pre_out + 5 * layer + step]         step 0 in-proj, 1 attention, 2 out-proj, 3 c_fc, 4 c_proj of each residual block
 *   the last two                            ln_post on the class rows, the final projection
 * (an interval holds the kernel and the launch gap in front of it, so the entries add up to the pass).  Returns the number of
 * entries written (> 0) or a negative error code; synchronises the stream; `out` receives the features as usual. */
int clipmi_encode_image_timed(clipmi_model* m, const void* image, int image_dtype, int batch, float* out, void* workspace,
                              size_t workspace_bytes, unsigned flags, float* us_out, int n_us, int* n_pre_out, clipmi_stream_t stream);

/* Ceiling probe for bench.py (`ceiling.mfma_only`; not on any product path): a register-only loop of v_mfma_f32_16x16x32_f16 -- no LDS, no
 * memory inside the loop -- on one workgroup of `waves` waves (1..8: two per SIMD, 256 registers each) per CU, every wave holding two register-resident sets of 4 + 4
 * operand fragments loaded once from `operands` (fp16 [16][waves * 64][8]: whatever distribution the caller wants the matrix pipe to
 * multiply, e.g. N(0, 0.25^2) like the tower's operands) and issuing iters x 32 MFMAs (one operand held for four instructions, as the GEMM
 * loops do).  2 * 2 * 64 * 64 * 32 flop per wave and iteration.  `sink` (fp32 [CUs * waves * 64]) keeps the results live; `clocks` (NULL or
 * 2 x uint64) receives workgroup 0's elapsed shader cycles and 100 MHz ticks.  *n_cus_out (host, may be NULL) = workgroups launched. */
int clipmi_probe_mfma_f16(const void* operands, float* sink, unsigned long long* clocks, int waves, int iters, int* n_cus_out,
                          clipmi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIPMI_H */
