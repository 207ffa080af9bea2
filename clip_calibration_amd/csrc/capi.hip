// C ABI of libclipmi.so (include/clipmi.h): error plumbing, the model handle and the tower drivers, i.e. the
// sequence of kernel launches that stands in for VisionTransformer.forward / Transformer.forward / encode_text
// (reference clip/model.py:394-424, 334-359, 600-613).  Host-side C++ only; kernels live in the other .hip files.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "common.h"

namespace clipmi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CLIPMI_ERR_HIP;
  }
  return CLIPMI_OK;
}

namespace {

struct OptionDesc { const char* name; const char* env; std::atomic<int> Options::*field; };
const OptionDesc kOptions[] = {
    {"gemm_variant", "CLIPMI_GEMM_VARIANT", &Options::gemm_variant},
    {"gemm_band", "CLIPMI_GEMM_BAND", &Options::gemm_band},
    {"gemm_stream", "CLIPMI_GEMM_STREAM", &Options::gemm_stream},
    {"gemm_rstream", "CLIPMI_GEMM_RSTREAM", &Options::gemm_rstream},
    {"gemm_split_rows", "CLIPMI_GEMM_SPLIT_ROWS", &Options::gemm_split_rows},
    {"cls_only_last_block", "CLIPMI_CLS_ONLY_LAST_BLOCK", &Options::cls_only_last_block},
    {"ln_fold", "CLIPMI_LN_FOLD", &Options::ln_fold},
    {"residual_f16", "CLIPMI_RESIDUAL_F16", &Options::residual_f16},
    {"attn_loader", "CLIPMI_ATTN_LOADER", &Options::attn_loader},
    {"attn_ring", "CLIPMI_ATTN_RING", &Options::attn_ring},
    {"attn_small", "CLIPMI_ATTN_SMALL", &Options::attn_small},
    {"tail_unfused", "CLIPMI_TAIL_UNFUSED", &Options::tail_unfused},
    {"vision_pass", "CLIPMI_VISION_PASS", &Options::vision_pass},
};

// environment spelling -> option value: decimal integers, plus the historical letters of two switches
// (CLIPMI_GEMM_VARIANT a/s/r = 10/13/16, CLIPMI_RESIDUAL_F16 v/t = 2/3)
int parse_option(const char* name, const char* text) {
  if (!strcmp(name, "gemm_variant")) {
    switch (text[0]) { case 'a': return 10; case 's': return 13; case 'r': return 16; default: break; }
  }
  if (!strcmp(name, "residual_f16")) {
    switch (text[0]) { case 'v': return 2; case 't': return 3; default: break; }
  }
  return atoi(text);
}

Options g_options;
std::once_flag g_options_once;

std::atomic<int> g_cus[64];

}  // namespace

#ifdef CLIPMI_TUNING
std::atomic<long long*> g_tuning_stamps{nullptr};
std::atomic<int> g_tuning_knob{0};
#endif

Options& options() {
  std::call_once(g_options_once, [] {
    for (const OptionDesc& d : kOptions) {
      const char* e = getenv(d.env);
      if (e && e[0]) (g_options.*(d.field)).store(parse_option(d.name, e), std::memory_order_relaxed);
    }
  });
  return g_options;
}

int current_device() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return dev;
}

int device_cus() {
  const int dev = current_device();
  if (dev >= 0 && dev < 64) {
    const int cached = g_cus[dev].load(std::memory_order_relaxed);
    if (cached > 0) return cached;
  }
  int n = 0;
  if (dev >= 0 && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  if (n <= 0) n = 256;
  if (dev >= 0 && dev < 64) g_cus[dev].store(n, std::memory_order_relaxed);
  return n;
}

}  // namespace clipmi

using namespace clipmi;

struct clipmi_model {
  clipmi_geometry g;
  bool has_vision = false, has_text = false;
  // per-handle settings (clipmi_model_set_option): -1 = follow the process-wide default of the same name
  std::atomic<int> opt_residual_f16{-1}, opt_ln_fold{-1}, opt_cls_only{-1};
  int residual_mode() const { const int v = opt_residual_f16.load(std::memory_order_relaxed); return v >= 0 ? v : options().residual_f16.load(std::memory_order_relaxed); }
  int ln_fold() const { const int v = opt_ln_fold.load(std::memory_order_relaxed); return v >= 0 ? v : options().ln_fold.load(std::memory_order_relaxed); }
  int cls_only() const { const int v = opt_cls_only.load(std::memory_order_relaxed); return v >= 0 ? v : options().cls_only_last_block.load(std::memory_order_relaxed); }
  clipmi_vision_weights vw;
  clipmi_text_weights tw;
  std::vector<clipmi_block_weights> vblocks, tblocks;
  int grid() const { return g.image_resolution / g.patch_size; }
  int tokens0() const { return grid() * grid() + 1; }
  int kpad() const { return round_up(3 * g.patch_size * g.patch_size, 64); }
  size_t col_bytes(int batch) const { return (size_t)batch * grid() * grid() * kpad() * 2; }
};

namespace {

// Workspace of one transformer tower over M = n_seq * L token rows of width D.
struct TowerWs {
  float* xres;    // [M, D]  fp32 residual stream
  half_t* xn;     // [M, D]  LayerNorm output / generic fp16 row buffer
  half_t* qkv;    // [M, 3D] (vision: aliased by the fp32 pre-ln_pre embeddings [M, D])
  half_t* att;    // [M, D]
  half_t* hid;    // [M, 4D] (vision: aliased by the im2col matrix)
  int32_t* idx;   // [2 * n_seq] eot / gather rows (text)
  float* stats;   // [LN_MAX_PARTS][M][2] row-sum partials of the residual stream (LayerNorm fold)
  size_t bytes;
};

TowerWs carve(void* base, int64_t M, int D, int n_seq, size_t hid_min_bytes = 0) {
  TowerWs w;
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t n) {
    char* r = p ? p + off : nullptr;
    off += align256(n);
    return r;
  };
  w.xres = reinterpret_cast<float*>(take((size_t)M * D * 4));
  w.xn = reinterpret_cast<half_t*>(take((size_t)M * D * 2));
  w.qkv = reinterpret_cast<half_t*>(take((size_t)M * D * 6));
  w.att = reinterpret_cast<half_t*>(take((size_t)M * D * 2));
  const size_t hid_bytes = (size_t)M * D * 8;
  w.hid = reinterpret_cast<half_t*>(take(hid_bytes > hid_min_bytes ? hid_bytes : hid_min_bytes));
  w.idx = reinterpret_cast<int32_t*>(take((size_t)n_seq * 8));
  w.stats = reinterpret_cast<float*>(take((size_t)M * 8 * LN_MAX_PARTS));
  w.bytes = off;
  return w;
}

int check_block(const clipmi_block_weights& b) {
  const void* ptrs[] = {b.ln1_g, b.ln1_b, b.w_qkv, b.b_qkv, b.w_out, b.b_out, b.ln2_g, b.ln2_b, b.w_fc, b.b_fc, b.w_proj, b.b_proj};
  for (const void* p : ptrs) {
    CLIPMI_REQUIRE(p != nullptr, CLIPMI_ERR_ARG, "block weights: null pointer");
    CLIPMI_REQUIRE((uintptr_t)p % 16 == 0, CLIPMI_ERR_ARG, "block weights: pointers must be 16-byte aligned");
  }
  const void* f[] = {b.w_qkv_f, b.g_qkv, b.c_qkv, b.w_fc_f, b.g_fc, b.c_fc};
  int n_set = 0;
  for (const void* p : f) {
    n_set += p != nullptr;
    CLIPMI_REQUIRE((uintptr_t)p % 16 == 0, CLIPMI_ERR_ARG, "block weights: folded operands must be 16-byte aligned");
  }
  CLIPMI_REQUIRE(n_set == 0 || n_set == 6, CLIPMI_ERR_ARG, "block weights: give all six LayerNorm-folded operands or none");
  return CLIPMI_OK;
}

// One ResidualAttentionBlock (clip/model.py:185-188) over M = n_seq*L rows, as five launches:
//   0 in-proj (+ ln_1)   1 attention   2 out-proj + residual   3 c_fc + QuickGELU (+ ln_2)   4 c_proj + residual
// folded == true: on entry AND on exit w.xn holds fp16(xres) and w.stats the *parts row-sum partials of xres (written
// by the residual epilogues); ln_1 / ln_2 are applied inside the in-proj / c_fc GEMM epilogues.
// folded == false: separate LayerNorm kernels (launched with steps 0 and 3).
// cls_only (image tower, LAST block, option cls_only_last_block, the product default since round 6): only the class token's row of every
// sequence leaves the tower (clip/model.py:419: ln_post(x[:, 0, :])), so the last block computes what that row needs and nothing else:
//   0  the in-projection's K | V thirds for every token (every token is a key) -- the same GEMM on weight rows D .. 3D -- and its Q third for
//      the n_seq class rows (M = n_seq, row stride L * D, LayerNorm statistics read with the same stride);
//   1  attention of ONE query per (sequence, head) against every key (attention_cls.hip);
//   2..4  out-proj, c_fc, c_proj and their LayerNorms on the class rows: the same GEMMs with M = n_seq and row stride L * D.
// Same arithmetic per element as the every-row block; of a 12-layer tower's 35.1 GFLOP per image 2.6 are never issued.  bench.py's headline
// `value` runs every row of every block (option 0) and reports this path as `value_cls_only`.
int run_block_step(int step, const clipmi_block_weights& b, const TowerWs& w, int n_seq, int L, int D, int causal, bool folded,
                   int* parts, hipStream_t s, bool f16res, bool cls_only = false) {
  const int H = D / 64;
  const bool cls = cls_only && step >= 2;
  const int M = cls ? n_seq : n_seq * L;              // rows of this step's GEMM
  const int64_t rowD = cls ? (int64_t)L * D : D;      // distance between consecutive rows in the [n_seq * L, D] buffers
  const int64_t Mfull = (int64_t)n_seq * L;
  int rc;
  GemmArgs a{};
  switch (step) {
    case 0: {
      if (!folded) {
        if ((rc = launch_layernorm(w.xres, CLIPMI_F32, D, nullptr, b.ln1_g, b.ln1_b, w.xn, CLIPMI_F16, D, M, D, 1e-5f, s))) return rc;
        a.W = (const half_t*)b.w_qkv; a.bias = b.b_qkv;
      } else {
        a.W = (const half_t*)b.w_qkv_f; a.bias = b.c_qkv; a.ln_stats = w.stats; a.ln_parts = *parts; a.ln_g = b.g_qkv; a.ln_dim = D;
        a.ln_eps = 1e-5f;
        if (*parts < LN_MAX_PARTS) a.ln_rows = w.stats + (size_t)2 * (LN_MAX_PARTS - 1) * Mfull;   // the last partial slot is free
      }
      a.A = w.xn; a.lda = D; a.ldw = D; a.out = w.qkv; a.ldo = 3 * D;
      a.out_dtype = CLIPMI_F16; a.M = M; a.N = 3 * D; a.K = D; a.epilogue = CLIPMI_EPI_BIAS;
      if (!cls_only) return launch_gemm(a, s);
      GemmArgs q = a;                                  // Q third, class rows: q | k | v row n * L of the packed buffer, columns 0 .. D
      q.lda = (int64_t)L * D; q.ldo = (int64_t)L * 3 * D; q.M = n_seq; q.N = D; q.ln_rows = nullptr;
      q.ln_plane = Mfull; q.ln_row_stride = L;         // the producer wrote one statistics row per token row
      a.W += (int64_t)D * D; a.bias += D; a.out = w.qkv + D; a.N = 2 * D;   // K | V thirds, every row
      if (a.ln_g) a.ln_g += D;
      if ((rc = launch_gemm(a, s))) return rc;
      return launch_gemm(q, s);
    }
    case 1:
      if (cls_only && !causal) return launch_attention_cls(w.qkv, w.att, n_seq, L, H, s);
      return launch_attention(w.qkv, w.att, n_seq, L, H, causal, s);
    case 2:
      a.A = w.att; a.lda = rowD; a.W = (const half_t*)b.w_out; a.ldw = D; a.bias = b.b_out; a.residual = w.xres; a.out = w.xres;
      a.ldo = rowD; a.out_dtype = CLIPMI_F32; a.M = M; a.N = D; a.K = D; a.epilogue = CLIPMI_EPI_BIAS_RESIDUAL;
      if (folded) { a.x16 = w.xn; a.stats_out = w.stats; a.parts_out = parts; a.residual_f16 = f16res; }
      return launch_gemm(a, s);
    case 3:
      if (!folded) {
        if ((rc = launch_layernorm(w.xres, CLIPMI_F32, rowD, nullptr, b.ln2_g, b.ln2_b, w.xn, CLIPMI_F16, rowD, M, D, 1e-5f, s))) return rc;
        a.W = (const half_t*)b.w_fc; a.bias = b.b_fc;
      } else {
        a.W = (const half_t*)b.w_fc_f; a.bias = b.c_fc; a.ln_stats = w.stats; a.ln_parts = *parts; a.ln_g = b.g_fc; a.ln_dim = D;
        a.ln_eps = 1e-5f;
        if (*parts < LN_MAX_PARTS) a.ln_rows = w.stats + (size_t)2 * (LN_MAX_PARTS - 1) * Mfull;
      }
      a.A = w.xn; a.lda = rowD; a.ldw = D; a.out = w.hid; a.ldo = 4 * D;
      a.out_dtype = CLIPMI_F16; a.M = M; a.N = 4 * D; a.K = D; a.epilogue = CLIPMI_EPI_BIAS_QUICKGELU;
      return launch_gemm(a, s);
    default:
      a.A = w.hid; a.lda = 4 * D; a.W = (const half_t*)b.w_proj; a.ldw = 4 * D; a.bias = b.b_proj; a.residual = w.xres; a.out = w.xres;
      a.ldo = rowD; a.out_dtype = CLIPMI_F32; a.M = M; a.N = D; a.K = 4 * D; a.epilogue = CLIPMI_EPI_BIAS_RESIDUAL;
      if (folded) { a.x16 = w.xn; a.stats_out = w.stats; a.parts_out = parts; a.residual_f16 = f16res; }
      return launch_gemm(a, s);
  }
}

// clipmi_encode_image_timed: one hipEvent behind every launch of a real tower pass (recorded asynchronously, read after the pass).  All events are
// created BEFORE the first launch (reserve): between launches tick() only records, so no event creation lands in a measured launch gap.
struct LaunchTimer {
  hipStream_t s;
  std::vector<hipEvent_t> pool, ev;
  bool ok = true;
  bool reserve(int n) {
    for (int i = 0; i < n; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { ok = false; return false; }
      pool.push_back(e);
    }
    return true;
  }
  void tick() {
    if (ev.size() >= pool.size()) { ok = false; return; }
    hipEvent_t e = pool[ev.size()];
    ev.push_back(e);
    if (hipEventRecord(e, s) != hipSuccess) ok = false;
  }
  ~LaunchTimer() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};

int run_block(const clipmi_block_weights& b, const TowerWs& w, int n_seq, int L, int D, int causal, bool folded, int* parts,
              hipStream_t s, bool f16res = false, bool cls_only = false, LaunchTimer* timer = nullptr) {
  for (int step = 0; step < 5; ++step) {
    const int rc = run_block_step(step, b, w, n_seq, L, D, causal, folded, parts, s, f16res, cls_only);
    if (rc) return rc;
    if (timer) timer->tick();
  }
  return CLIPMI_OK;
}

// ln_1 / ln_2 are applied inside the GEMM epilogues whenever the folded operands are bound (option ln_fold = 0, process-wide or
// per handle, switches back to the separate LayerNorm kernels).  Per residual block at B = 256 (profiles/r01_ln_fold.txt): the two
// LayerNorm launches (2 x 39 us) disappear; the consumers pay +5-6 us each and the residual producers +13 us each (the fp16
// shadow of the stream is another 77 MB in their store burst): net -41 us per block, +3 % end to end.
bool fold_enabled(const clipmi_model* m, const std::vector<clipmi_block_weights>& blocks) {
  if (m->ln_fold() == 0) return false;
  for (const auto& b : blocks)
    if (!b.w_qkv_f) return false;
  return !blocks.empty();
}

// Residual-stream precision (needs the fold).  The reference's GPU path keeps the stream in fp16 (clip/model.py:186-187
// adds fp16 tensors); here it is
//   image tower: fp16 by default -- the stream IS the fp16 operand copy the fold already writes, so the two residual
//                GEMMs of a block move 154 MB each instead of 387 MB (+10 % end to end at B = 256); against the fp32
//                stream the image-side cosine error goes 3.1e-5 -> 8.7e-5 (tests/precision_modes.py), tolerance 1e-3;
//   text tower:  fp32 with an fp16 shadow -- its features are computed once per class list and reused for every image.
// Resolution order: the call's flags (CLIPMI_CALL_STREAM_F32 / _F16), then the handle's residual_f16 setting, then the
// process-wide option (env CLIPMI_RESIDUAL_F16) = 0 (fp32 everywhere) | 2 / v (default) | 3 / t | 1 (both towers fp16).
// *rc_out: CLIPMI_ERR_STATE when the call demands the fp16 stream but the fold is off (no fp16 operand copy to carry it).
bool residual_f16_enabled(const clipmi_model* m, bool folded, bool vision, unsigned flags, int* rc_out) {
  *rc_out = CLIPMI_OK;
  const unsigned prec = flags & (CLIPMI_CALL_STREAM_F32 | CLIPMI_CALL_STREAM_F16);
  if (prec == (CLIPMI_CALL_STREAM_F32 | CLIPMI_CALL_STREAM_F16) || (flags & ~(CLIPMI_CALL_STREAM_F32 | CLIPMI_CALL_STREAM_F16))) {
    set_error("tower call: bad flags 0x%x", flags);
    *rc_out = CLIPMI_ERR_ARG;
    return false;
  }
  if (prec == CLIPMI_CALL_STREAM_F32) return false;
  if (prec == CLIPMI_CALL_STREAM_F16) {
    if (!folded) {
      set_error("tower call: CLIPMI_CALL_STREAM_F16 needs the LayerNorm-folded operands and ln_fold != 0");
      *rc_out = CLIPMI_ERR_STATE;
    }
    return folded;
  }
  if (!folded) return false;
  const int mode = m->residual_mode();
  return mode == 1 || (vision ? mode == 2 : mode == 3);
}

int check_hook(const clipmi_prompt_hook* hook, int layers, bool vision) {
  if (!hook) return CLIPMI_OK;
  CLIPMI_REQUIRE(hook->n_ctx > 0 && hook->n_deep >= 0 && hook->n_deep <= layers - 1, CLIPMI_ERR_SHAPE,
                 "prompt hook: n_ctx=%d n_deep=%d invalid for %d layers", hook->n_ctx, hook->n_deep, layers);
  CLIPMI_REQUIRE(hook->n_deep == 0 || hook->deep, CLIPMI_ERR_ARG, "prompt hook: deep prompts missing");
  CLIPMI_REQUIRE(!vision || hook->shallow, CLIPMI_ERR_ARG, "prompt hook: the vision tower needs the shallow prompt");
  return CLIPMI_OK;
}

// text blocks on w.xres (already holds embeddings + pos)
// L = token rows per prompt that are computed (the whole context, or the caller's seq_rows: see include/clipmi.h clipmi_text_encoder)
int run_text_blocks(clipmi_model* m, const TowerWs& w, int C, int L, const clipmi_prompt_hook* hook, bool folded, bool f16res, hipStream_t s) {
  const int D = m->g.text_width;
  int rc, parts = 1;
  if (folded && (rc = launch_row_stats(w.xres, w.xn, w.stats, 1, C, L, D, 0, L, s))) return rc;   // input rows of block 0
  for (int i = 0; i < m->g.text_layers; ++i) {
    if (hook && i > 0 && i - 1 < hook->n_deep) {
      if ((rc = launch_overwrite_tokens(w.xres, hook->deep + (int64_t)(i - 1) * hook->n_ctx * D, C, L, D, 1, hook->n_ctx, s))) return rc;
      if (folded && (rc = launch_row_stats(w.xres, w.xn, w.stats, parts, C, L, D, 1, hook->n_ctx, s))) return rc;
    }
    if ((rc = run_block(m->tblocks[i], w, C, L, D, 1, folded, &parts, s, f16res))) return rc;
  }
  return CLIPMI_OK;
}

// ln_final on the EOT rows + text_projection (clip/model.py:607-611)
int run_text_tail(clipmi_model* m, const TowerWs& w, int C, float* out, bool f16res, hipStream_t s) {
  const int D = m->g.text_width, E = m->g.embed_dim;
  int rc;
  half_t* rows = f16res ? w.att : w.xn;   // fp16 stream mode: w.xn IS the stream, the gathered rows go to the idle attention buffer
  if (f16res) rc = launch_layernorm(w.xn, CLIPMI_F16, D, w.idx + C, m->tw.ln_final_g, m->tw.ln_final_b, rows, CLIPMI_F16, D, C, D, 1e-5f, s);
  else rc = launch_layernorm(w.xres, CLIPMI_F32, D, w.idx + C, m->tw.ln_final_g, m->tw.ln_final_b, rows, CLIPMI_F16, D, C, D, 1e-5f, s);
  if (rc) return rc;
  GemmArgs a{};
  a.A = rows; a.lda = D; a.W = (const half_t*)m->tw.proj_t; a.ldw = D; a.out = out; a.ldo = E; a.out_dtype = CLIPMI_F32;
  a.M = C; a.N = E; a.K = D; a.epilogue = CLIPMI_EPI_NONE;
  return launch_gemm(a, s);
}

// rows of a prompt the tower works on: the caller's bound on the last live token (dead-row elimination), else the whole context
int live_rows(const clipmi_model* m, int seq_rows) { return seq_rows > 0 && seq_rows < m->g.context_length ? seq_rows : m->g.context_length; }

int text_prologue(clipmi_model* m, int n_prompts, int seq_rows, const clipmi_prompt_hook* hook, void* ws, size_t ws_bytes, unsigned flags, TowerWs* w,
                  bool* folded, bool* f16res) {
  CLIPMI_REQUIRE(m, CLIPMI_ERR_ARG, "null model");
  CLIPMI_REQUIRE(m->has_text, CLIPMI_ERR_STATE, "text weights not bound (clipmi_set_text_weights)");
  {
    int prc;
    *folded = fold_enabled(m, m->tblocks);
    *f16res = residual_f16_enabled(m, *folded, false, flags, &prc);
    if (prc) return prc;
  }
  CLIPMI_REQUIRE(n_prompts >= 0, CLIPMI_ERR_SHAPE, "n_prompts=%d", n_prompts);
  CLIPMI_REQUIRE((int64_t)n_prompts * m->g.context_length < (1ll << 31), CLIPMI_ERR_SHAPE, "too many prompt tokens");
  int rc = check_hook(hook, m->g.text_layers, false);
  if (rc) return rc;
  const int L = live_rows(m, seq_rows);
  if (hook) CLIPMI_REQUIRE(1 + hook->n_ctx <= L, CLIPMI_ERR_SHAPE, "prompt hook: n_ctx=%d does not fit the %d token rows that are computed", hook->n_ctx, L);
  *w = carve(ws, (int64_t)n_prompts * L, m->g.text_width, n_prompts);
  CLIPMI_REQUIRE(ws || n_prompts == 0, CLIPMI_ERR_ARG, "null workspace");
  CLIPMI_REQUIRE(ws_bytes >= w->bytes, CLIPMI_ERR_WORKSPACE, "text workspace too small: %zu < %zu", ws_bytes, w->bytes);
  CLIPMI_REQUIRE((uintptr_t)ws % 256 == 0, CLIPMI_ERR_ARG, "workspace must be 256-byte aligned");
  return CLIPMI_OK;
}

}  // namespace

extern "C" {

int clipmi_abi_version(void) { return CLIPMI_ABI_VERSION; }

const char* clipmi_strerror(int code) {
  switch (code) {
    case CLIPMI_OK: return "ok";
    case CLIPMI_ERR_ARG: return "invalid argument";
    case CLIPMI_ERR_SHAPE: return "unsupported shape";
    case CLIPMI_ERR_HIP: return "HIP runtime error";
    case CLIPMI_ERR_WORKSPACE: return "workspace too small";
    case CLIPMI_ERR_STATE: return "weights not bound";
    default: return "unknown error";
  }
}

const char* clipmi_last_error(void) { return g_err; }

int clipmi_set_option(const char* name, int value) {
  CLIPMI_REQUIRE(name, CLIPMI_ERR_ARG, "set_option: null name");
  for (const OptionDesc& d : kOptions)
    if (!strcmp(d.name, name)) {
      (options().*(d.field)).store(value, std::memory_order_relaxed);
      return CLIPMI_OK;
    }
  set_error("set_option: unknown option '%s'", name);
  return CLIPMI_ERR_ARG;
}

#ifdef CLIPMI_TUNING
int clipmi_tuning_set_stamps(void* device_buffer) {   // tuning build only; not part of include/clipmi.h
  g_tuning_stamps.store(static_cast<long long*>(device_buffer), std::memory_order_relaxed);
  return CLIPMI_OK;
}
#endif

#ifdef CLIPMI_TUNING
int clipmi_tuning_set_knob(int bits) {   // tuning build only
  g_tuning_knob.store(bits, std::memory_order_relaxed);
  return CLIPMI_OK;
}
#endif

int clipmi_get_option(const char* name, int* value) {
  CLIPMI_REQUIRE(name && value, CLIPMI_ERR_ARG, "get_option: null pointer");
  for (const OptionDesc& d : kOptions)
    if (!strcmp(d.name, name)) {
      *value = (options().*(d.field)).load(std::memory_order_relaxed);
      return CLIPMI_OK;
    }
  set_error("get_option: unknown option '%s'", name);
  return CLIPMI_ERR_ARG;
}

// ---------------------------------------------------------------- operator level
int clipmi_gemm_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual,
                    void* out, int64_t ldo, int out_dtype, int M, int N, int K, int epilogue, clipmi_stream_t stream) {
  CLIPMI_REQUIRE(epilogue >= CLIPMI_EPI_NONE && epilogue <= CLIPMI_EPI_BIAS_RESIDUAL16_RELU, CLIPMI_ERR_ARG, "gemm: bad epilogue %d", epilogue);
  if (M == 0) return CLIPMI_OK;
  GemmArgs a{};
  a.A = (const half_t*)A; a.lda = lda; a.W = (const half_t*)W; a.ldw = ldw; a.bias = bias; a.residual = residual;
  a.out = out; a.ldo = ldo; a.out_dtype = out_dtype; a.M = M; a.N = N; a.K = K; a.epilogue = epilogue;
  return launch_gemm(a, (hipStream_t)stream);
}

int clipmi_layernorm(const void* x, int x_dtype, int64_t in_stride, const int32_t* gather_idx, const float* gamma,
                     const float* beta, void* y, int y_dtype, int64_t out_stride, int rows, int D, float eps,
                     clipmi_stream_t stream) {
  return launch_layernorm(x, x_dtype, in_stride, gather_idx, gamma, beta, y, y_dtype, out_stride, rows, D, eps, (hipStream_t)stream);
}

int clipmi_attention(const void* qkv, void* out, int N, int L, int H, int causal, clipmi_stream_t stream) {
  return launch_attention((const half_t*)qkv, (half_t*)out, N, L, H, causal, (hipStream_t)stream);
}

int clipmi_patchify(const void* image, int image_dtype, void* col, int B, int R, int P, int Kpad, clipmi_stream_t stream) {
  return launch_patchify(image, image_dtype, (half_t*)col, B, R, P, Kpad, (hipStream_t)stream);
}

size_t clipmi_patch_embed_scratch_bytes(int B, int R, int image_dtype) { return (B < 0 || R < 0) ? 0 : patch_embed_scratch_bytes(B, R, image_dtype); }

int clipmi_patch_embed(const void* image, int image_dtype, void* scratch, const void* conv_w, int64_t ldw, const float* pos, void* x0, int x0_dtype,
                       int B, int R, int P, int D, int tokens, clipmi_stream_t stream) {
  return launch_patch_embed(image, image_dtype, scratch, (const half_t*)conv_w, ldw, pos, x0, x0_dtype, B, R, P, D, tokens, (hipStream_t)stream);
}

int clipmi_embed_ln(const void* x0, int x0_dtype, int add_pos, const float* cls, const float* pos, const float* shallow, const float* gamma,
                    const float* beta, float* y, void* y16, float* stats, int B, int L, int tokens0, int D, float eps, clipmi_stream_t stream) {
  return launch_embed_ln(x0, x0_dtype, add_pos, cls, pos, shallow, gamma, beta, y, (half_t*)y16, stats, B, L, tokens0, D, eps, (hipStream_t)stream);
}

int clipmi_l2_normalize(const void* in, int in_dtype, float* out, int rows, int E, clipmi_stream_t stream) {
  return launch_l2_normalize(in, in_dtype, out, rows, E, (hipStream_t)stream);
}

int clipmi_l2_normalize_to(const void* in, int in_dtype, void* out, int out_dtype, int rows, int E, clipmi_stream_t stream) {
  return launch_l2_normalize_to(in, in_dtype, out, out_dtype, rows, E, (hipStream_t)stream);
}

int clipmi_logits(const float* img_n, const float* txt_n, float scale, const float* dac_conf, float* logits, float* conf,
                  int32_t* pred, int B, int C, int E, clipmi_stream_t stream) {
  return launch_logits(img_n, txt_n, scale, dac_conf, logits, conf, pred, B, C, E, (hipStream_t)stream);
}

size_t clipmi_fused_tail_workspace_bytes(int B, int C) { return (B < 0 || C < 0) ? 0 : fused_tail_workspace_bytes(B, C); }

int clipmi_fused_tail(const void* img, int img_dtype, int normalize, const float* txt_n, float scale, const float* dac_conf, float* logits,
                      float* img_n_out, float* conf, int32_t* pred, const int64_t* labels, double* bins, int n_bins,
                      void* workspace, size_t workspace_bytes, int B, int C, int E, clipmi_stream_t stream) {
  return launch_fused_tail(img, img_dtype, normalize, txt_n, scale, dac_conf, logits, img_n_out, conf, pred, labels, bins, n_bins, workspace,
                           workspace_bytes, B, C, E, (hipStream_t)stream);
}

int clipmi_calibrate_rows(float* logits, const float* dac_conf, float* conf, int32_t* pred, int B, int C, clipmi_stream_t stream) {
  return launch_calibrate_rows(logits, dac_conf, conf, pred, B, C, (hipStream_t)stream);
}
int clipmi_conv3x3_nhwc(const void* x, const void* w, const float* bias, void* out, int B, int H, int W, int C, int Cout, int relu,
                        clipmi_stream_t stream) {
  return launch_conv3x3((const half_t*)x, (const half_t*)w, bias, (half_t*)out, B, H, W, C, Cout, relu, (hipStream_t)stream);
}
int clipmi_im2col3x3_nchw(const void* image, int image_dtype, void* col, int B, int Cin, int H, int W, int stride, int Kpad,
                          clipmi_stream_t stream) {
  return launch_im2col3x3_nchw(image, image_dtype, (half_t*)col, B, Cin, H, W, stride, Kpad, (hipStream_t)stream);
}
int clipmi_im2col3x3_nhwc(const void* x, void* col, int B, int H, int W, int C, int Kpad, clipmi_stream_t stream) {
  return launch_im2col3x3_nhwc((const half_t*)x, (half_t*)col, B, H, W, C, Kpad, (hipStream_t)stream);
}
int clipmi_avgpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int k, clipmi_stream_t stream) {
  return launch_avgpool_nhwc((const half_t*)x, (half_t*)y, B, H, W, C, k, (hipStream_t)stream);
}
int clipmi_attnpool_tokens(const void* x, const float* pos, void* tokens, int B, int HW, int C, clipmi_stream_t stream) {
  return launch_attnpool_tokens((const half_t*)x, pos, (half_t*)tokens, B, HW, C, (hipStream_t)stream);
}
int clipmi_attnpool(const void* q, const void* kv, void* out, int B, int T, int heads, clipmi_stream_t stream) {
  return launch_attnpool((const half_t*)q, (const half_t*)kv, (half_t*)out, B, T, heads, (hipStream_t)stream);
}
int clipmi_adapter_blend(const float* feats, const float* w1, const float* w2, float ratio, float* out, int B, int E, int H,
                         clipmi_stream_t stream) {
  return launch_adapter_blend(feats, w1, w2, ratio, out, B, E, H, (hipStream_t)stream);
}
int clipmi_scale_add(const float* a, const float* b, float alpha, float* out, long long n, clipmi_stream_t stream) {
  return launch_scale_add(a, b, alpha, out, (int64_t)n, (hipStream_t)stream);
}
int clipmi_group_mean(const float* in, float* out, int G, int P, int E, clipmi_stream_t stream) {
  return launch_group_mean(in, out, G, P, E, (hipStream_t)stream);
}
int clipmi_cocoop_ctx(const float* img_n, const float* w1, const float* b1, const float* w2, const float* b2, const float* ctx,
                      float* ctx_shifted, int B, int E, int H, int D, int n_ctx, clipmi_stream_t stream) {
  return launch_cocoop_ctx(img_n, w1, b1, w2, b2, ctx, ctx_shifted, B, E, H, D, n_ctx, (hipStream_t)stream);
}
int clipmi_cocoop_prompts(const void* base, int base_dtype, const float* ctx_shifted, void* prompts, int n_images, int C, int L,
                          int D, int n_ctx, clipmi_stream_t stream) {
  return launch_cocoop_prompts(base, base_dtype, ctx_shifted, (half_t*)prompts, n_images, C, L, D, n_ctx, (hipStream_t)stream);
}
int clipmi_logits_per_image(const float* img_n, const float* txt, float scale, const float* dac_conf, float* logits, float* conf,
                            int32_t* pred, float* txt_n_last, int B, int C, int E, clipmi_stream_t stream) {
  return launch_logits_per_image(img_n, txt, scale, dac_conf, logits, conf, pred, txt_n_last, B, C, E, (hipStream_t)stream);
}
int clipmi_softmax_rows(const float* logits, const float* dac_conf, float* probs, float* conf, int32_t* pred, int B, int C,
                        clipmi_stream_t stream) {
  return launch_softmax_rows(logits, dac_conf, probs, conf, pred, B, C, (hipStream_t)stream);
}

int clipmi_knn_dists(const float* queries, const float* refs, float* out, int Nq, int Nr, int E, int K, clipmi_stream_t stream) {
  return launch_knn(queries, refs, out, Nq, Nr, E, K, (hipStream_t)stream);
}

int clipmi_ece_accumulate(const float* conf, const int32_t* pred, const int64_t* labels, int n, double* bins, int n_bins,
                          clipmi_stream_t stream) {
  return launch_ece_accumulate(conf, pred, labels, n, bins, n_bins, (hipStream_t)stream);
}

// The fp16-stream residual GEMM of a block as an operator (out-proj / c_proj of the image tower, clip/model.py:186-187):
// x16[m,n] = fp16(x16[m,n] + A[m,:] . W[n,:] + bias[n]) in place, plus the per-row (sum, sum of squares) partials of the ROUNDED
// values, one pair per 256-column tile: stats[(t * M + m) * 2 ..].  *parts (host) receives the number of column tiles.
int clipmi_gemm_residual_f16(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* x16, int64_t ldx,
                             float* stats, int* parts, int M, int N, int K, clipmi_stream_t stream) {
  CLIPMI_REQUIRE(parts, CLIPMI_ERR_ARG, "gemm_residual_f16: null parts pointer");
  *parts = 0;
  if (M == 0) return CLIPMI_OK;
  GemmArgs a{};
  a.A = (const half_t*)A; a.lda = lda; a.W = (const half_t*)W; a.ldw = ldw; a.bias = bias; a.ldo = ldx; a.out_dtype = CLIPMI_F32;
  a.M = M; a.N = N; a.K = K; a.epilogue = CLIPMI_EPI_BIAS_RESIDUAL;
  a.x16 = (half_t*)x16; a.stats_out = stats; a.parts_out = parts; a.residual_f16 = true;
  return launch_gemm(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------- model level
int clipmi_create(const clipmi_geometry* geom, clipmi_model** out) {
  CLIPMI_REQUIRE(geom && out, CLIPMI_ERR_ARG, "create: null pointer");
  const clipmi_geometry& g = *geom;
  CLIPMI_REQUIRE(g.vision_width > 0 && g.vision_width % 64 == 0 && g.text_width > 0 && g.text_width % 64 == 0, CLIPMI_ERR_SHAPE,
                 "create: widths must be multiples of 64 (head_dim 64): vision %d text %d", g.vision_width, g.text_width);
  CLIPMI_REQUIRE(g.text_heads * 64 == g.text_width, CLIPMI_ERR_SHAPE, "create: text_heads*64 != text_width");
  CLIPMI_REQUIRE(g.patch_size > 0 && g.image_resolution % g.patch_size == 0, CLIPMI_ERR_SHAPE, "create: resolution %% patch != 0");
  CLIPMI_REQUIRE(g.embed_dim > 0 && g.embed_dim % 16 == 0, CLIPMI_ERR_SHAPE, "create: embed_dim must be a multiple of 16");
  CLIPMI_REQUIRE(g.vision_layers > 0 && g.text_layers > 0 && g.context_length > 0 && g.vocab_size > 0, CLIPMI_ERR_SHAPE,
                 "create: non-positive geometry field");
  clipmi_model* m = new (std::nothrow) clipmi_model();
  CLIPMI_REQUIRE(m, CLIPMI_ERR_ARG, "create: out of host memory");
  m->g = g;
  *out = m;
  return CLIPMI_OK;
}

int clipmi_destroy(clipmi_model* m) {
  delete m;
  return CLIPMI_OK;
}

int clipmi_model_set_option(clipmi_model* m, const char* name, int value) {
  CLIPMI_REQUIRE(m && name, CLIPMI_ERR_ARG, "model_set_option: null pointer");
  std::atomic<int>* f = !strcmp(name, "residual_f16") ? &m->opt_residual_f16 : !strcmp(name, "ln_fold") ? &m->opt_ln_fold
                        : !strcmp(name, "cls_only_last_block") ? &m->opt_cls_only : nullptr;
  CLIPMI_REQUIRE(f, CLIPMI_ERR_ARG, "model_set_option: unknown option '%s' (residual_f16, ln_fold, cls_only_last_block)", name);
  CLIPMI_REQUIRE(value >= -1 && value <= (f == &m->opt_residual_f16 ? 3 : 1), CLIPMI_ERR_ARG, "model_set_option: %s = %d out of range", name, value);
  f->store(value, std::memory_order_relaxed);
  return CLIPMI_OK;
}

int clipmi_model_get_option(const clipmi_model* m, const char* name, int* value) {
  CLIPMI_REQUIRE(m && name && value, CLIPMI_ERR_ARG, "model_get_option: null pointer");
  if (!strcmp(name, "residual_f16")) *value = m->residual_mode();
  else if (!strcmp(name, "ln_fold")) *value = m->ln_fold();
  else if (!strcmp(name, "cls_only_last_block")) *value = m->cls_only();
  else {
    set_error("model_get_option: unknown option '%s'", name);
    return CLIPMI_ERR_ARG;
  }
  return CLIPMI_OK;
}

int clipmi_set_vision_weights(clipmi_model* m, const clipmi_vision_weights* w) {
  CLIPMI_REQUIRE(m && w && w->blocks, CLIPMI_ERR_ARG, "set_vision_weights: null pointer");
  const void* ptrs[] = {w->conv_w, w->class_embedding, w->positional_embedding, w->ln_pre_g, w->ln_pre_b, w->ln_post_g, w->ln_post_b, w->proj_t};
  for (const void* p : ptrs) CLIPMI_REQUIRE(p && (uintptr_t)p % 16 == 0, CLIPMI_ERR_ARG, "set_vision_weights: null/unaligned pointer");
  for (int i = 0; i < m->g.vision_layers; ++i) {
    int rc = check_block(w->blocks[i]);
    if (rc) return rc;
  }
  m->vblocks.assign(w->blocks, w->blocks + m->g.vision_layers);
  m->vw = *w;
  m->vw.blocks = m->vblocks.data();
  m->has_vision = true;
  return CLIPMI_OK;
}

int clipmi_set_text_weights(clipmi_model* m, const clipmi_text_weights* w) {
  CLIPMI_REQUIRE(m && w && w->blocks, CLIPMI_ERR_ARG, "set_text_weights: null pointer");
  const void* ptrs[] = {w->token_embedding, w->positional_embedding, w->ln_final_g, w->ln_final_b, w->proj_t};
  for (const void* p : ptrs) CLIPMI_REQUIRE(p && (uintptr_t)p % 16 == 0, CLIPMI_ERR_ARG, "set_text_weights: null/unaligned pointer");
  for (int i = 0; i < m->g.text_layers; ++i) {
    int rc = check_block(w->blocks[i]);
    if (rc) return rc;
  }
  m->tblocks.assign(w->blocks, w->blocks + m->g.text_layers);
  m->tw = *w;
  m->tw.blocks = m->tblocks.data();
  m->has_text = true;
  return CLIPMI_OK;
}

// Images per pass of the image tower.  Throughput per image peaks where one pass works on about 50 432 x 768 stream elements (ViT-B/16:
// 256 images; ViT-L/14: 128; ViT-L/14@336: 64; ViT-B/32: 992) and falls 8-10 % for passes two to four times that size, whose
// activations (0.3 GB of MLP hidden state per pass at the peak) no longer find their consumers' reads in the 256 MB Infinity Cache
// (profiles/r03_batch_passes.txt).  A larger batch is therefore run as consecutive passes on the same stream and workspace.  Each pass is an
// ordinary call on its images: against one pass over the whole batch the features differ as they do between any two batch sizes (tile and
// kernel choice follow the row count: 1.5e-4 on normalised ViT-B/16 features; bit for bit when the same kernels are chosen).  Option vision_pass = the
// element count (0 = never split).
static int pass_images(int batch, int L, int D) {
  const int64_t elems = options().vision_pass.load(std::memory_order_relaxed);
  if (elems <= 0) return batch;
  int64_t p = elems / ((int64_t)L * D);
  p = p >= 32 ? p / 32 * 32 : (p < 1 ? 1 : p);
  if ((int64_t)batch * 2 < p * 3) return batch;   // less than one and a half passes: not worth a ragged second one
  return (int)p;
}

size_t clipmi_vision_workspace_bytes(const clipmi_model* m, int batch, int n_ctx) {
  if (!m || batch < 0 || n_ctx < 0) return 0;
  const int L = m->tokens0() + n_ctx;
  const int pass = pass_images(batch, L, m->g.vision_width);
  const int rem = batch % pass;
  const int b = batch <= pass ? batch : pass + (rem < pass / 4 ? rem : 0);   // a remainder below a quarter pass joins the last pass
  return carve(nullptr, (int64_t)b * L, m->g.vision_width, b, m->col_bytes(b)).bytes;
}

size_t clipmi_text_workspace_bytes(const clipmi_model* m, int n_prompts, int seq_rows) {
  if (!m || n_prompts < 0) return 0;
  return carve(nullptr, (int64_t)n_prompts * live_rows(m, seq_rows), m->g.text_width, n_prompts).bytes;
}

static int encode_image_pass(clipmi_model* m, const void* image, int image_dtype, int batch, const clipmi_prompt_hook* hook, float* out,
                             void* workspace, size_t workspace_bytes, unsigned flags, clipmi_stream_t stream, LaunchTimer* timer = nullptr,
                             int* n_pre = nullptr);

int clipmi_encode_image(clipmi_model* m, const void* image, int image_dtype, int batch, const clipmi_prompt_hook* hook, float* out,
                        void* workspace, size_t workspace_bytes, unsigned flags, clipmi_stream_t stream) {
  CLIPMI_REQUIRE(m, CLIPMI_ERR_ARG, "encode_image: null model");
  CLIPMI_REQUIRE(m->has_vision, CLIPMI_ERR_STATE, "vision weights not bound (clipmi_set_vision_weights)");
  CLIPMI_REQUIRE(batch >= 0, CLIPMI_ERR_SHAPE, "encode_image: batch=%d", batch);
  if (batch == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(image && out && workspace, CLIPMI_ERR_ARG, "encode_image: null pointer");
  CLIPMI_REQUIRE(image_dtype == CLIPMI_F16 || image_dtype == CLIPMI_F32, CLIPMI_ERR_ARG, "encode_image: image dtype %d", image_dtype);
  int rc = check_hook(hook, m->g.vision_layers, true);
  if (rc) return rc;
  const int L = m->tokens0() + (hook ? hook->n_ctx : 0);
  const int pass = pass_images(batch, L, m->g.vision_width);
  const size_t image_bytes = (size_t)3 * m->g.image_resolution * m->g.image_resolution * (image_dtype == CLIPMI_F16 ? 2 : 4);
  for (int lo = 0; lo < batch;) {
    int n = batch - lo < pass ? batch - lo : pass;
    if (batch - lo - n < pass / 4) n = batch - lo;   // a short remainder joins this pass
    rc = encode_image_pass(m, static_cast<const char*>(image) + (size_t)lo * image_bytes, image_dtype, n, hook, out + (size_t)lo * m->g.embed_dim,
                           workspace, workspace_bytes, flags, stream);
    if (rc) return rc;
    lo += n;
  }
  return CLIPMI_OK;
}

static int encode_image_pass(clipmi_model* m, const void* image, int image_dtype, int batch, const clipmi_prompt_hook* hook, float* out,
                             void* workspace, size_t workspace_bytes, unsigned flags, clipmi_stream_t stream, LaunchTimer* timer, int* n_pre) {
  int rc = CLIPMI_OK;
  auto tick = [&] { if (timer) timer->tick(); };
  hipStream_t s = (hipStream_t)stream;
  const clipmi_geometry& g = m->g;
  const int G = m->grid(), L0 = m->tokens0(), n_ctx = hook ? hook->n_ctx : 0, L = L0 + n_ctx;
  const int D = g.vision_width, E = g.embed_dim, Kpad = m->kpad();
  CLIPMI_REQUIRE((int64_t)batch * L < (1ll << 31) / 4, CLIPMI_ERR_SHAPE, "encode_image: batch too large for one call");
  const TowerWs w = carve(workspace, (int64_t)batch * L, D, batch, m->col_bytes(batch));
  CLIPMI_REQUIRE(workspace_bytes >= w.bytes, CLIPMI_ERR_WORKSPACE, "vision workspace too small: %zu < %zu", workspace_bytes, w.bytes);
  CLIPMI_REQUIRE((uintptr_t)workspace % 256 == 0, CLIPMI_ERR_ARG, "workspace must be 256-byte aligned");

  const bool folded = fold_enabled(m, m->vblocks);
  const bool f16res = residual_f16_enabled(m, folded, true, flags, &rc);
  if (rc) return rc;
  int parts = 1;
  tick();   // start of the pass
  if (patch_embed_fits(batch, g.image_resolution, g.patch_size, D) && Kpad == 3 * g.patch_size * g.patch_size) {
    // patch_embed.hip: (an fp32 image is cast to fp16 in one streaming pass, into the MLP hidden buffer;) the patch GEMM's loader reads the
    // NCHW image itself -- the im2col matrix is an address map --, adds pos and scatters the token rows: fp16 rows when the stream is fp16 (the
    // precision the reference's GPU path holds them in), fp32 rows otherwise; ln_pre then runs over every token row, forming the class row and
    // MaPLe's shallow prompt rows on the fly, and leaves the stream (and / or its fp16 operand copy + row sums for the first in-projection's fold).
    const int x0_dtype = f16res ? CLIPMI_F16 : CLIPMI_F32;
    void* x0 = w.qkv;                              // [B*L, D] embeddings before ln_pre (fp16 or fp32: at most 4 of the region's 6 bytes per element)
    if ((rc = launch_patch_embed(image, image_dtype, w.hid /* >= col_bytes = the fp16 image's size */, (const half_t*)m->vw.conv_w, Kpad,
                                 nullptr /* pos: added by ln_pre's row pass */, x0, x0_dtype, batch, g.image_resolution, g.patch_size, D, L, s)))
      return rc;
    tick();   // (one interval: the cast of an fp32 image + the GEMM)
    if ((rc = launch_embed_ln(x0, x0_dtype, 1, m->vw.class_embedding, m->vw.positional_embedding, hook ? hook->shallow : nullptr, m->vw.ln_pre_g,
                              m->vw.ln_pre_b, f16res ? nullptr : w.xres, folded ? w.xn : nullptr, folded ? w.stats : nullptr, batch, L, L0, D, 1e-5f, s)))
      return rc;
    tick();
    if (n_pre) *n_pre = 2;
  } else {
    // patch sizes that are not a multiple of 8 (ViT-L/14: a row segment of 14 pixels is not a whole number of 16-byte LDS slots): im2col
    // matrix + GEMM + class rows + ln_pre as four launches
    half_t* col = w.hid;                             // [B*G*G, Kpad]
    float* x0 = reinterpret_cast<float*>(w.qkv);     // [B*L, D] embeddings before ln_pre
    if ((rc = launch_patchify(image, image_dtype, col, batch, g.image_resolution, g.patch_size, Kpad, s))) return rc;
    tick();
    GemmArgs a{};
    a.A = col; a.lda = Kpad; a.W = (const half_t*)m->vw.conv_w; a.ldw = Kpad; a.out = x0; a.ldo = D; a.out_dtype = CLIPMI_F32;
    a.M = batch * G * G; a.N = D; a.K = Kpad; a.epilogue = EPI_PATCH_POS;
    a.pos = m->vw.positional_embedding; a.patches = G * G; a.tokens = L;
    if ((rc = launch_gemm(a, s))) return rc;
    tick();
    if ((rc = launch_cls_and_ctx_rows(x0, m->vw.class_embedding, m->vw.positional_embedding, hook ? hook->shallow : nullptr, batch, L0,
                                      n_ctx, D, s)))
      return rc;
    tick();
    // fp16 residual stream: nothing reads the fp32 copy of ln_pre's output (the blocks work on w.xn) -- 155 MB less to write at batch 256
    if ((rc = launch_layernorm(x0, CLIPMI_F32, D, nullptr, m->vw.ln_pre_g, m->vw.ln_pre_b, f16res ? nullptr : w.xres, CLIPMI_F32, D, batch * L, D,
                               1e-5f, s, folded ? w.xn : nullptr, folded ? w.stats : nullptr)))
      return rc;
    tick();
    if (n_pre) *n_pre = 4;   // patchify, patch-embedding GEMM, class / context rows, ln_pre
  }
  for (int i = 0; i < g.vision_layers; ++i) {
    if (hook && i > 0 && i - 1 < hook->n_deep) {
      if ((rc = launch_overwrite_tokens(w.xres, hook->deep + (int64_t)(i - 1) * n_ctx * D, batch, L, D, L - n_ctx, n_ctx, s))) return rc;
      if (folded && (rc = launch_row_stats(w.xres, w.xn, w.stats, parts, batch, L, D, L - n_ctx, n_ctx, s))) return rc;
    }
    const bool cls_only = i == g.vision_layers - 1 && m->cls_only() == 1;
    if ((rc = run_block(m->vblocks[i], w, batch, L, D, 0, folded, &parts, s, f16res, cls_only, timer))) return rc;
  }
  // ln_post on the class token only, then @ proj (clip/model.py:419-422)
  half_t* cls_rows = f16res ? w.att : w.xn;
  if (f16res) rc = launch_layernorm(w.xn, CLIPMI_F16, (int64_t)L * D, nullptr, m->vw.ln_post_g, m->vw.ln_post_b, cls_rows, CLIPMI_F16, D, batch, D, 1e-5f, s);
  else rc = launch_layernorm(w.xres, CLIPMI_F32, (int64_t)L * D, nullptr, m->vw.ln_post_g, m->vw.ln_post_b, cls_rows, CLIPMI_F16, D, batch, D, 1e-5f, s);
  if (rc) return rc;
  tick();
  GemmArgs a{};
  a.A = cls_rows; a.lda = D; a.W = (const half_t*)m->vw.proj_t; a.ldw = D; a.out = out; a.ldo = E; a.out_dtype = CLIPMI_F32;
  a.M = batch; a.N = E; a.K = D; a.epilogue = CLIPMI_EPI_NONE;
  rc = launch_gemm(a, s);
  tick();
  return rc;
}

int clipmi_encode_image_timed(clipmi_model* m, const void* image, int image_dtype, int batch, float* out, void* workspace, size_t workspace_bytes,
                              unsigned flags, float* us_out, int n_us, int* n_pre_out, clipmi_stream_t stream) {
  CLIPMI_REQUIRE(m && image && out && workspace && us_out && n_pre_out, CLIPMI_ERR_ARG, "encode_image_timed: null pointer");
  CLIPMI_REQUIRE(m->has_vision, CLIPMI_ERR_STATE, "vision weights not bound (clipmi_set_vision_weights)");
  CLIPMI_REQUIRE(image_dtype == CLIPMI_F16 || image_dtype == CLIPMI_F32, CLIPMI_ERR_ARG, "encode_image_timed: image dtype %d", image_dtype);
  const int L = m->tokens0();
  CLIPMI_REQUIRE(batch > 0 && pass_images(batch, L, m->g.vision_width) >= batch, CLIPMI_ERR_SHAPE,
                 "encode_image_timed: batch=%d must be one pass of the tower (option vision_pass)", batch);
  LaunchTimer timer;
  timer.s = (hipStream_t)stream;
  const int max_events = 1 + 4 + 5 * m->g.vision_layers + 2;   // pass start, at most 4 embedding launches, 5 per block, ln_post + projection
  CLIPMI_REQUIRE(n_us >= max_events - 1, CLIPMI_ERR_ARG, "encode_image_timed: us_out holds %d entries, up to %d needed", n_us, max_events - 1);
  CLIPMI_REQUIRE(timer.reserve(max_events), CLIPMI_ERR_HIP, "encode_image_timed: hipEventCreate failed");
  int n_pre = 0;
  int rc = encode_image_pass(m, image, image_dtype, batch, nullptr, out, workspace, workspace_bytes, flags, stream, &timer, &n_pre);
  if (rc) return rc;
  const int n = (int)timer.ev.size() - 1;
  CLIPMI_REQUIRE(timer.ok && n == n_pre + 5 * m->g.vision_layers + 2, CLIPMI_ERR_HIP, "encode_image_timed: event bookkeeping failed");
  CLIPMI_REQUIRE(n_us >= n, CLIPMI_ERR_ARG, "encode_image_timed: us_out holds %d entries, %d needed", n_us, n);
  if (hipEventSynchronize(timer.ev.back()) != hipSuccess) {
    set_error("encode_image_timed: hipEventSynchronize failed");
    return CLIPMI_ERR_HIP;
  }
  for (int i = 0; i < n; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, timer.ev[i], timer.ev[i + 1]) != hipSuccess) {
      set_error("encode_image_timed: hipEventElapsedTime failed");
      return CLIPMI_ERR_HIP;
    }
    us_out[i] = 1e3f * ms;
  }
  *n_pre_out = n_pre;
  return n;
}

int clipmi_text_blocks(clipmi_model* m, const void* x, void* y, int dtype, int n_prompts, int seq_rows, const clipmi_prompt_hook* hook,
                       void* workspace, size_t workspace_bytes, unsigned flags, clipmi_stream_t stream) {
  TowerWs w;
  bool folded, f16res;
  int rc = text_prologue(m, n_prompts, seq_rows, hook, workspace, workspace_bytes, flags, &w, &folded, &f16res);
  if (rc) return rc;
  if (n_prompts == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x && y, CLIPMI_ERR_ARG, "text_blocks: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int Lc = m->g.context_length, L = live_rows(m, seq_rows), D = m->g.text_width;
  CLIPMI_REQUIRE(L == Lc || x != y, CLIPMI_ERR_ARG, "text_blocks: in place (x == y) only without seq_rows");
  if ((rc = launch_add_pos(x, dtype, nullptr, w.xres, n_prompts, L, Lc, D, s))) return rc;
  if ((rc = run_text_blocks(m, w, n_prompts, L, hook, folded, f16res, s))) return rc;
  if (L < Lc)   // the caller said that only the first `seq_rows` token rows matter: they go back to their places, the rows behind them are zero
    return f16res ? launch_rows_out(w.xn, CLIPMI_F16, y, dtype, n_prompts, L, Lc, D, s) : launch_rows_out(w.xres, CLIPMI_F32, y, dtype, n_prompts, L, Lc, D, s);
  if (f16res) return launch_cast_f16(w.xn, y, dtype, (int64_t)n_prompts * L * D, s);
  return launch_cast_f32(w.xres, y, dtype, (int64_t)n_prompts * L * D, s);
}

int clipmi_text_encoder(clipmi_model* m, const void* prompts, int dtype, const int32_t* eot, int n_prompts, int seq_rows,
                        const clipmi_prompt_hook* hook, float* out, void* workspace, size_t workspace_bytes, unsigned flags,
                        clipmi_stream_t stream) {
  TowerWs w;
  bool folded, f16res;
  int rc = text_prologue(m, n_prompts, seq_rows, hook, workspace, workspace_bytes, flags, &w, &folded, &f16res);
  if (rc) return rc;
  if (n_prompts == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(prompts && eot && out, CLIPMI_ERR_ARG, "text_encoder: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int L = live_rows(m, seq_rows), D = m->g.text_width;   // rows [L, context_length) of every prompt are never read
  if ((rc = launch_add_pos(prompts, dtype, m->tw.positional_embedding, w.xres, n_prompts, L, m->g.context_length, D, s))) return rc;
  if ((rc = launch_eot_rows(eot, w.idx + n_prompts, n_prompts, L, s))) return rc;
  if ((rc = run_text_blocks(m, w, n_prompts, L, hook, folded, f16res, s))) return rc;
  return run_text_tail(m, w, n_prompts, out, f16res, s);
}

int clipmi_encode_text(clipmi_model* m, const int64_t* ids, int n_prompts, int seq_rows, float* out, void* workspace, size_t workspace_bytes,
                       unsigned flags, clipmi_stream_t stream) {
  TowerWs w;
  bool folded, f16res;
  int rc = text_prologue(m, n_prompts, seq_rows, nullptr, workspace, workspace_bytes, flags, &w, &folded, &f16res);
  if (rc) return rc;
  if (n_prompts == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(ids && out, CLIPMI_ERR_ARG, "encode_text: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int L = live_rows(m, seq_rows), D = m->g.text_width;
  if ((rc = launch_embed_tokens(ids, m->tw.token_embedding, m->tw.positional_embedding, w.xres, w.idx, n_prompts, L, m->g.context_length, D,
                                m->g.vocab_size, s)))
    return rc;
  if ((rc = launch_eot_rows(w.idx, w.idx + n_prompts, n_prompts, L, s))) return rc;
  if ((rc = run_text_blocks(m, w, n_prompts, L, nullptr, folded, f16res, s))) return rc;
  return run_text_tail(m, w, n_prompts, out, f16res, s);
}

int clipmi_profile_block(clipmi_model* m, int batch, int iters, int only, void* workspace, size_t workspace_bytes, float* ms_out,
                         clipmi_stream_t stream) {
  CLIPMI_REQUIRE(m && ms_out && workspace, CLIPMI_ERR_ARG, "profile: null pointer");
  CLIPMI_REQUIRE(m->has_vision, CLIPMI_ERR_STATE, "vision weights not bound");
  CLIPMI_REQUIRE(batch > 0 && iters > 0 && only >= -1 && only < 5, CLIPMI_ERR_SHAPE, "profile: batch/iters must be positive, only in -1..4");
  hipStream_t s = (hipStream_t)stream;
  const int L = m->tokens0(), D = m->g.vision_width;
  const TowerWs w = carve(workspace, (int64_t)batch * L, D, batch, m->col_bytes(batch));
  CLIPMI_REQUIRE(workspace_bytes >= w.bytes, CLIPMI_ERR_WORKSPACE, "profile: workspace too small");
  const clipmi_block_weights& b = m->vblocks[0];
  const bool folded = fold_enabled(m, m->vblocks);
  int prc;
  const bool f16res = residual_f16_enabled(m, folded, true, 0u, &prc);
  // the row partials a residual GEMM of this shape leaves behind (what the consumers read in the tower)
  int parts = (D + 255) / 256 > LN_MAX_PARTS ? LN_MAX_PARTS : (D + 255) / 256;
  // One event pair around `iters` launches of a step queued back to back (after one untimed launch): the GPU stays under load for the whole
  // measurement.  (Until round 5 every launch was timed and synchronised on its own: on a box whose clock sags in the idle gap between two
  // synchronised launches that read 6-7 % above the same kernel inside the tower and above rocprofv3's average.)
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    set_error("profile: hipEventCreate failed");
    return CLIPMI_ERR_HIP;
  }
  int rc = CLIPMI_OK;
  for (int step = 0; step < 5 && rc == CLIPMI_OK; ++step) {
    ms_out[step] = 0.f;
    if (only >= 0 && only != step) continue;
    rc = run_block_step(step, b, w, batch, L, D, 0, folded, &parts, s, f16res);   // untimed warm-up launch
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && rc == CLIPMI_OK; ++i) rc = run_block_step(step, b, w, batch, L, D, 0, folded, &parts, s, f16res);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess) {
      set_error("profile: hipEventSynchronize failed");
      rc = CLIPMI_ERR_HIP;
      break;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms_out[step] = ms / (float)iters;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
}

}  // extern "C"
