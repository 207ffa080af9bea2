// fp16 MFMA GEMM with fused epilogues for gfx950:  out[M,N] = epi(A[M,K] @ W[N,K]^T).
//
// Replaces every nn.Linear / in_proj / out_proj / `x @ proj` on the path (reference clip/model.py:174-176,183,
// 422,611) and, with EPI_PATCH_POS, the GEMM half of conv1 + pos-emb add (clip/model.py:395-402).
//
// Structure (cdna_hip_programming.md §5, "minimum 2-phase"): 128(m) x 128(n) x 64(k) workgroup tile, 4 waves in a
// 2x2 grid, each wave 64x64 = 4x4 v_mfma_f32_16x16x32_f16 tiles.  Both operands are K-contiguous, staged
// global->LDS by global_load_lds_dwordx4 (no VGPR round trip) into two 32 KiB stages; the LDS image is linear per
// wave-instruction, the XOR bank swizzle is applied on the per-lane SOURCE address and again on the ds_read_b128
// address (rule 21).  W is the MFMA "A" operand and the activations the "B" operand, so each lane ends up with 4
// consecutive n for one m -> 8/16-byte epilogue accesses on row-major [M,N].
// Workgroup ids are remapped so that the workgroups resident on one XCD sweep the n-tiles of the same m-tile
// (the activation tile is then fetched once into that XCD's L2).
#include "gemm_common.h"


namespace clipmi {

namespace {

using namespace gemm;


int pick_band(int tiles_n, int bn, int K);


// Phase stamps exist only in the tuning build (make TUNING=1): the product kernels take no stamp pointer.


// ---------------------------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs (removes the two LayerNorm kernels of every residual block and their 232 MB pass).
//   LN(x) @ W^T + b  =  rstd * (x @ (gamma*W)^T)  -  rstd*mean * g  +  c,   g[n] = sum_k (gamma*W)[n,k],
//                                                                           c[n] = sum_k beta[k] W[n,k] + b[n]
// The producer of x (a BIAS_RESIDUAL epilogue, or the ln_pre LayerNorm kernel) also writes fp16(x) -- the operand of
// the consumer GEMM -- and per-row partial sums of x and x^2, one (sum, sumsq) pair per column tile of the producer
// (reduced over the workgroup's waves through LDS, then a plain store: no atomics, nothing to zero, bit-reproducible).
// The consumer adds the partials in order and forms the variance E[x^2] - mean^2 in double.  g is summed from the
// fp16-rounded folded weights the MFMA actually multiplies, so the mean term cancels as it does inside a LayerNorm.
// ---------------------------------------------------------------------------------------------------------------
// explicit fused multiply-adds: the same bits wherever this is inlined (ln_finalize_kernel, the tile prologues, the streamed
// kernel's epilogue), whatever the surrounding code lets the compiler contract
__device__ __forceinline__ void ln_params_from_sums(const KArgs& a, double s, double ss, float& rstd, float& mu_rstd) {
  const double mean = s * (double)a.ln_inv_d;
  double var = __builtin_fma(ss, (double)a.ln_inv_d, -(mean * mean));
  var = var > 0.0 ? var : 0.0;
  rstd = rsqrtf((float)var + a.ln_eps);
  mu_rstd = (float)mean * rstd;
}

__device__ __forceinline__ void ln_row_params(const KArgs& a, int m, float& rstd, float& mu_rstd) {
  const int mm = m < a.M ? m : a.M - 1;
  double s = 0.0, ss = 0.0;
  for (int p = 0; p < a.ln_parts; ++p) {
    const float2 st = *reinterpret_cast<const float2*>(a.ln_stats + 2 * ((int64_t)p * a.ln_M + (int64_t)mm * a.ln_rs));
    s += (double)st.x;
    ss += (double)st.y;
  }
  ln_params_from_sums(a, s, ss, rstd, mu_rstd);
}

// Tile configuration: BM x BN workgroup tile (m = activation rows, n = weight rows), WGM x WGN waves.
template <int BM_, int BN_, int WGM_, int WGN_, int OCC_>
struct Tile {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, OCC = OCC_;
  static constexpr int NW = WGM * WGN, NT = NW * 64;
  static constexpr int WTM = BM / WGM, WTN = BN / WGN;   // wave tile
  static constexpr int TM = WTM / 16, TN = WTN / 16;     // 16x16 MFMA tiles per wave
  static constexpr int XBYTES = BM * BK * 2, WBYTES = BN * BK * 2;
  static constexpr int STAGE = XBYTES + WBYTES, SMEM = 2 * STAGE;
  static constexpr int XI = BM * 8 / NT, WI = BN * 8 / NT;  // global_load_lds instructions per thread per stage
  static_assert(WTM % 16 == 0 && WTN % 16 == 0 && (BM * 8) % NT == 0 && (BN * 8) % NT == 0, "bad tile");
  static_assert((NT / 8) % 16 == 0, "staging swizzle assumes NT/8 rows per instruction is a multiple of 16");
};

// QuickGELU (clip/model.py:162-164): x * sigmoid(1.702 x) = x / (1 + 2^(-k x)), k = 1.702 log2(e): one v_exp_f32 + one v_rcp_f32 (1 ulp) instead of
// the IEEE division sequence.  Round 6: the activation is taken of the pre-activation ROUNDED TO fp16 -- the reference's own fp16 path has exactly this
// rounding point (clip/model.py:174-177: c_fc's output is an fp16 tensor before QuickGELU reads it) -- which lets gemm_stream_kernel keep a tile's
// pre-activations as 64 fp16 registers and apply the activation inside the NEXT tile's K loop, two registers per 16-MFMA compute part, where the vector
// pipe has nothing else to do (profiles/r06_gelu_in_compute_part.txt).  Every kernel goes through gelu_preact / quick_gelu_h (or, in that K loop, the
// same five operations as single instructions: gelu_uop): the same bits wherever the activation is applied.
//   t = -k h (one rounding); e = 2^t; d = 1 + e; r = 1 / d; out = fp16(h r)      h -> +big: e = 0, out = h; h -> -big: e = inf, r = 0, out = -0
constexpr float GELU_K = 2.4554669595930157f;
// v = acc * rstd + (bias - mean rstd g), explicit fused multiply-adds
__device__ __forceinline__ f16x4 gelu_preact(const f32x4& acc, float rs, float mrs, const f32x4& bias, const f32x4& g) {
  f32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf(acc[q], rs, __builtin_fmaf(-mrs, g[q], bias[q]));
  return f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
}
__device__ __forceinline__ float quick_gelu_h1(half_t h) {
  const float z = (float)h;
  const float t = z * -GELU_K;
  const float d = 1.0f + __builtin_amdgcn_exp2f(t);
  return z * __builtin_amdgcn_rcpf(d);
}
__device__ __forceinline__ f16x4 quick_gelu_h(const f16x4& h) {
  return f16x4{(half_t)quick_gelu_h1(h[0]), (half_t)quick_gelu_h1(h[1]), (half_t)quick_gelu_h1(h[2]), (half_t)quick_gelu_h1(h[3])};
}
// fp32 outputs (OUT_F32 callers: tests, fp32 hidden activations): nothing is rounded on the way
__device__ __forceinline__ float quick_gelu(float z) {
  return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -GELU_K));
}
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// One instruction of the activation of ONE packed register w = (h_lo, h_hi) -> (out_lo, out_hi), Q = 0 .. 10 in order; tl / th are its two
// temporaries.  gemm_stream_kernel places them between the MFMAs of a compute part (asm volatile: they stay where they are written), two registers
// interleaved so that no instruction reads the result of the one in front of it.
// MFMA with the accumulator tied to the destination (see gemm_stream_kernel); a function template, not a generic lambda: clang does not implicitly
// capture a variable that a generic lambda names only as an asm operand
template <bool ZERO>
__device__ __forceinline__ void mfma_tied(f32x4& acc, const f16x8& w, const f16x8& x) {
  if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(x));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(x));
}
template <int Q, int E>
__device__ __forceinline__ void gelu_uop(u32x4_t& w4, float& tl, float& th, float neg_k) {   // w = element E of w4
  if constexpr (Q == 0) asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(tl) : "v"(w4[E]), "s"(neg_k));
  if constexpr (Q == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(th) : "v"(w4[E]), "s"(neg_k));
  if constexpr (Q == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(tl));
  if constexpr (Q == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(th));
  if constexpr (Q == 4) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(tl));
  if constexpr (Q == 5) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(th));
  if constexpr (Q == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(tl));
  if constexpr (Q == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(th));
  if constexpr (Q == 8) asm volatile("v_fma_mix_f32 %0, %1, %0, 0 op_sel_hi:[1,0,0]" : "+v"(tl) : "v"(w4[E]));
  if constexpr (Q == 9) asm volatile("v_fma_mix_f32 %0, %1, %0, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(th) : "v"(w4[E]));
  if constexpr (Q == 10) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(w4[E]) : "v"(tl), "v"(th));
}

template <int E>
__device__ __forceinline__ void gelu_swap(u32x4_t& w4) {   // elements E and E + 2: odd 16-lane rows of the first against even rows of the second
  asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w4[E]), "+v"(w4[E + 2]));
}

// Tile traversal.  blockIdx % 8 labels the XCD (blocks are dealt round-robin over the 8 XCDs); each label gets a
// contiguous range of logical tile ids (bijective remap), and logical ids walk the tile grid in BANDS of `band`
// n-tiles: inside a band m is the slow index and n the fast one.  The ~32 tiles an XCD runs concurrently then share
// `band` weight panels (resident in that XCD's 4 MiB L2 for the whole band) and ~32/band activation panels, which
// stream through once per band -- instead of cycling through all of W (3.5-4.7 MB) for every m-tile.
__device__ __forceinline__ void tile_coords(const KArgs& a, int tiles_m, int& tile_m, int& tile_n) {
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = a.nwg >> 3, r = a.nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int per_band = tiles_m * a.band;
  const int b = wg / per_band;
  const int within = wg - b * per_band;
  const int rem = a.tiles_n - b * a.band;
  const int gw = rem < a.band ? rem : a.band;
  tile_m = within / gw;
  tile_n = b * a.band + (within - tile_m * gw);
}

// fp16 outputs: each wave transposes its tile through a private LDS patch (32 rows x 64 cols at a time) so that the
// global stores are 16 B per lane and 128 contiguous bytes per row, instead of 8-byte pieces of 32-byte row segments.
// The caller must have passed a workgroup barrier after the last main-loop LDS read.
template <typename T, int EPI, int CH>
__device__ __forceinline__ void epilogue_f16_staged(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int wave_m,
                                                    int wave_n, int lane, char* patch, const float2* lnp = nullptr) {
  constexpr int TM = T::TM, TN = T::TN;
  static_assert(T::WTN % 64 == 0 && TM % CH == 0 && (CH == 1 || CH == 2), "staged epilogue works on 64-column slices of the wave tile");
  constexpr int NH = T::WTN / 64;        // 64-column slices per wave tile (1 for the 64-wide wave tiles, 2 for 128)
  constexpr int ROWB = 64 * 2 + 16;      // 144 B: 16-B aligned rows, 2-way (cheap) bank conflicts on the 8-B writes
  const int r16 = lane & 15, g4 = lane >> 4;
  f32x4 bias[TN], lng[TN];
  const bool fold = a.ln_stats != nullptr;   // wave-uniform
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    bias[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    lng[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI != CLIPMI_EPI_NONE) {
      const int n = n0 + wave_n * T::WTN + i * 16 + g4 * 4;
      if (n < a.N) {
        bias[i] = *reinterpret_cast<const f32x4*>(a.bias + n);
        if (fold) lng[i] = *reinterpret_cast<const f32x4*>(a.ln_g + n);
      }
    }
  }
  half_t* out = static_cast<half_t*>(a.out);
  const int rrow = lane >> 3, rcol = lane & 7;
#pragma unroll
  for (int jc = 0; jc < TM / CH; ++jc) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int n_st = n0 + wave_n * T::WTN + h * 64 + rcol * 8;
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) {
        float rs = 1.f, mrs = 0.f;
        if (fold) {
          const int ml = wave_m * T::WTM + (jc * CH + jj) * 16 + r16;
          if (lnp) {   // (rstd, mean*rstd) of the tile's rows, put in LDS by the kernel prologue
            const float2 pr = lnp[ml];
            rs = pr.x;
            mrs = pr.y;
          } else {
            ln_row_params(a, m0 + ml, rs, mrs);
          }
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = h * 4 + ii;
          f32x4 v;
          if constexpr (EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
            const f16x4 o = quick_gelu_h(gelu_preact(acc[i][jc * CH + jj], rs, mrs, bias[i], lng[i]));
            v = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};   // (exact: rounded again below without change)
          } else {
            v = acc[i][jc * CH + jj] * rs + (bias[i] - mrs * lng[i]);
          }
          if constexpr (EPI == CLIPMI_EPI_BIAS_RESIDUAL16_RELU) {
            const int mr = m0 + wave_m * T::WTM + (jc * CH + jj) * 16 + r16;
            const int nr = n0 + wave_n * T::WTN + i * 16 + g4 * 4;
            if (mr < a.M && nr < a.N) {
              const f16x4 r = *reinterpret_cast<const f16x4*>(a.residual16 + (int64_t)mr * a.ldo + nr);
              v += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            }
          }
          if constexpr (EPI == CLIPMI_EPI_BIAS_RELU || EPI == CLIPMI_EPI_BIAS_RESIDUAL16_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *reinterpret_cast<f16x4*>(patch + (jj * 16 + r16) * ROWB + (ii * 16 + g4 * 4) * 2) =
              f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
      }
      // same wave, LDS is in order: the reads below see the writes above (and the next slice's writes follow these reads)
#pragma unroll
      for (int t = 0; t < 2 * CH; ++t) {
        const int row = t * 8 + rrow;
        const f16x8 val = *reinterpret_cast<const f16x8*>(patch + row * ROWB + rcol * 16);
        const int m = m0 + wave_m * T::WTM + jc * (16 * CH) + row;
        if (m < a.M && n_st < a.N) *reinterpret_cast<f16x8*>(out + (int64_t)m * a.ldo + n_st) = val;
      }
    }
  }
}

// Producer side of the LayerNorm fold: BIAS_RESIDUAL epilogue that, besides the fp32 read-modify-write of the residual
// stream, stores fp16(out) to x16 through the wave-private LDS transpose and writes this tile's row partials.
// Worked in chunks of 32 rows with a scheduling fence between chunks, so that the residual loads of later chunks are
// not hoisted over the whole epilogue (the 160-accumulator tile has no registers to spare).
// F16RES: the residual stream IS the fp16 copy (the reference's own GPU precision: clip/model.py:186-187 adds in fp16):
// the operand is read from x16, the sum is rounded to fp16 and written back in place, the row partials are taken from
// the ROUNDED values (what the consumer's MFMA will read), and no fp32 pass is made -- 154 MB per launch instead of 387.
template <typename T, bool F16RES = false>
__device__ __forceinline__ void epilogue_residual_fold(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int tile_n,
                                                       int wave_m, int wave_n, int lane, int wave, char* smem) {
  constexpr int TM = T::TM, TN = T::TN;
  constexpr int ROWB = T::WTN * 2 + 16;
  constexpr int PATCH = 32 * ROWB;
  static_assert(T::WTN == 64 && TM % 2 == 0, "fold epilogue assumes 64-column wave tiles");
  const int r16 = lane & 15, g4 = lane >> 4;
  char* patch = smem + wave * PATCH;
  float2* red = reinterpret_cast<float2*>(smem + T::NW * PATCH);   // [WGN][BM]
  __syncthreads();   // every wave is done with the main-loop LDS image
  float* out = static_cast<float*>(a.out);
  const int rrow = lane >> 3, rcol = lane & 7;
  const int n_st = n0 + wave_n * T::WTN + rcol * 8;
  f32x4 bias[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wave_n * T::WTN + i * 16 + g4 * 4;
    bias[i] = n < a.N ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int jc = 0; jc < TM / 2; ++jc) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int ml = wave_m * T::WTM + (jc * 2 + jj) * 16 + r16;
      const int m = m0 + ml;
      float rsum = 0.f, rsq = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + wave_n * T::WTN + i * 16 + g4 * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < a.M && n < a.N) {
          if constexpr (F16RES) {
            const f16x4 r = *reinterpret_cast<const f16x4*>(a.x16 + (int64_t)m * a.ldo + n);
            v = acc[i][jc * 2 + jj] + bias[i] + f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            fold_row_sums16(f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}, rsum, rsq);
          } else {
            v = acc[i][jc * 2 + jj] + bias[i] + *reinterpret_cast<const f32x4*>(a.residual + (int64_t)m * a.ldo + n);
            *reinterpret_cast<f32x4*>(out + (int64_t)m * a.ldo + n) = v;
            fold_row_sums(v, rsum, rsq);
          }
        }
        *reinterpret_cast<f16x4*>(patch + (jj * 16 + r16) * ROWB + (i * 16 + g4 * 4) * 2) =
            f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      }
      rsum = row4_sum(rsum);   // the 4 lanes of a row
      rsq = row4_sum(rsq);
      if (g4 == 0) red[wave_n * T::BM + ml] = make_float2(rsum, rsq);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {   // same wave, LDS in order: reads see the writes above
      const int row = t * 8 + rrow;
      const f16x8 val = *reinterpret_cast<const f16x8*>(patch + row * ROWB + rcol * 16);
      const int m = m0 + wave_m * T::WTM + jc * 32 + row;
      if (m < a.M && n_st < a.N) {
        if constexpr (CLIPMI_STORE_AUX & 2) __builtin_nontemporal_store(val, reinterpret_cast<f16x8*>(a.x16 + (int64_t)m * a.ldo + n_st));
        else *reinterpret_cast<f16x8*>(a.x16 + (int64_t)m * a.ldo + n_st) = val;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T::BM; t += T::NT) {
    const int m = m0 + t;
    if (m < a.M) {
      float sx = 0.f, sq = 0.f;
#pragma unroll
      for (int w = 0; w < T::WGN; ++w) {
        const float2 pr = red[w * T::BM + t];
        sx += pr.x;
        sq += pr.y;
      }
      *reinterpret_cast<float2*>(a.stats_out + 2 * ((int64_t)tile_n * a.M + m)) = make_float2(sx, sq);
    }
  }
}



// fp16-stream producer, second form (the default of gemm_f16_kernel): the residual operand comes in through LDS.
// The register-direct form above reads x16 as 8-byte pieces, 16 rows x 32 B per wave-instruction, in dependent 32-row
// chunks: phase stamps put that epilogue at 32 us per 320 x 256 tile (two tiles per CU: a quarter of the c_proj launch
// and half of the out-proj launch).  Here every wave DMAs the rows of its own 64-column sub-tile (128 B per row, 8 rows
// per buffer_load ... lds instruction: full lines) into a private region of the now idle stage buffers -- up to four
// 32-row chunks in flight, XOR-swizzled on the source address exactly like the main loop's operands -- adds in place
// (the lane that reads an element is the lane that writes it), and stores the rows back as 16 B per lane.  Counted
// vmcnt waits: the DMA of later chunks and the stores of earlier ones stay in flight while a chunk is worked on.
template <typename T>
struct FoldDma {
  static constexpr int NCH = T::TM / 2;                 // 32-row chunks per wave
  static constexpr int RD = NCH < 4 ? NCH : 4;          // chunks in flight (ring of private 4 KiB regions)
  static constexpr int CHB = 32 * 128;
  static constexpr int RED_OFF = T::NW * RD * CHB;
  static constexpr int LDS = RED_OFF + T::WGN * T::BM * (int)sizeof(float2);
  // VMEM operations a wave issues after the last DMA instruction of chunk c and before it needs chunk c:
  // the rest of the initial burst, then per chunk p worked on in between 4 stores (+ 4 DMA instructions if p re-arms the ring)
  static constexpr int younger(int c) {
    int n = c < RD ? (RD - 1 - c) * 4 : 0;
    for (int p = (c < RD ? 0 : c - RD + 1); p < c; ++p) n += 4 + (p + RD < NCH ? 4 : 0);
    return n;
  }
};

template <typename F, int C = 0>
__device__ __forceinline__ void wait_chunk(int c) {   // c is a constant after unrolling: exactly one counted wait survives
  if constexpr (C < F::NCH) {
    if (c == C) wait_vmcnt<F::younger(C)>();
    else wait_chunk<F, C + 1>(c);
  }
}

template <typename T>
__device__ __forceinline__ void epilogue_residual_fold16_dma(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int tile_n,
                                                             int wave_m, int wave_n, int lane, int wave, char* smem) {
  using F = FoldDma<T>;
  constexpr int TM = T::TM, TN = T::TN, NCH = F::NCH, RD = F::RD, CHB = F::CHB;
  static_assert(T::WTN == 64 && TM % 2 == 0 && TN == 4, "fold epilogue assumes 64-column wave tiles");
  const int r16 = lane & 15, g4 = lane >> 4;
  char* region = smem + wave * (RD * CHB);
  float2* red = reinterpret_cast<float2*>(smem + F::RED_OFF);   // [WGN][BM]
  const int wrow = wave_m * T::WTM;
  const int col0 = n0 + wave_n * 64, row0 = m0 + wrow;   // wave-uniform
  f32x4 bias[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = col0 + i * 16 + g4 * 4;
    bias[i] = n < a.N ? *reinterpret_cast<const f32x4*>(a.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the bias loads: nothing but this epilogue's own traffic is counted below
  __syncthreads();                                    // every wave is done with the main-loop LDS image
  // rows at or beyond M lie outside the descriptor and read as zero
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x16 + (int64_t)row0 * a.ldo + col0, ((int64_t)(a.M - row0) * a.ldo - col0) * 2);
  const int drow = lane >> 3, dslot = lane & 7;
  auto dma_chunk = [&](int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = c * 32 + q * 8 + drow;
      const int voff = (row * (int)a.ldo + ((dslot ^ ((row >> 1) & 7)) << 3)) * 2;
      CLIPMI_BUFFER_LOAD_LDS16(rs, region + (c % RD) * CHB + q * 1024, voff, 0);
    }
  };
#pragma unroll
  for (int c = 0; c < RD; ++c) dma_chunk(c);
  // lane-constant LDS offsets inside a chunk: element (row jj*16 + r16, columns i*16 + g4*4 .. +3) = data chunk 2i + (g4>>1),
  // stored at slot (chunk ^ ((row>>1)&7)); the coalesced read-back takes slot (lane&7) of row t*8 + (lane>>3) as it lies
  int eoff[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) eoff[jj] = (jj * 16 + r16) * 128 + (g4 & 1) * 8;
  const int esw = (r16 >> 1) & 7;   // ((jj*16 + r16) >> 1) & 7
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    char* reg = region + (c % RD) * CHB;
    __builtin_amdgcn_sched_barrier(0);
    wait_chunk<F>(c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int ml = wrow + (c * 2 + jj) * 16 + r16;
      const int m = m0 + ml;
      float rsum = 0.f, rsq = 0.f;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = col0 + i * 16 + g4 * 4;
        char* p = reg + eoff[jj] + (((2 * i + (g4 >> 1)) ^ esw) << 4);
        const f16x4 r = *reinterpret_cast<const f16x4*>(p);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < a.M && n < a.N) {
          v = acc[i][c * 2 + jj] + bias[i] + f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
          fold_row_sums16(f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}, rsum, rsq);
        }
        *reinterpret_cast<f16x4*>(p) = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      }
      rsum = row4_sum(rsum);   // the 4 lanes of a row
      rsq = row4_sum(rsq);
      if (g4 == 0) red[wave_n * T::BM + ml] = make_float2(rsum, rsq);
    }
    // same wave, LDS in order: the reads below see the writes above
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rl = t * 8 + drow;
      const f16x8 val = *reinterpret_cast<const f16x8*>(reg + rl * 128 + dslot * 16);
      const int m = row0 + c * 32 + rl;
      const int n_st = col0 + ((dslot ^ ((rl >> 1) & 7)) << 3);
      if (m < a.M && n_st < a.N) {
        if constexpr (CLIPMI_STORE_AUX & 2) __builtin_nontemporal_store(val, reinterpret_cast<f16x8*>(a.x16 + (int64_t)m * a.ldo + n_st));
        else *reinterpret_cast<f16x8*>(a.x16 + (int64_t)m * a.ldo + n_st) = val;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (c + RD < NCH) dma_chunk(c + RD);   // its region was read (lgkmcnt drained for the stores above) a moment ago
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T::BM; t += T::NT) {
    const int m = m0 + t;
    if (m < a.M) {
      float sx = 0.f, sq = 0.f;
#pragma unroll
      for (int w = 0; w < T::WGN; ++w) {
        const float2 pr = red[w * T::BM + t];
        sx += pr.x;
        sq += pr.y;
      }
      *reinterpret_cast<float2*>(a.stats_out + 2 * ((int64_t)tile_n * a.M + m)) = make_float2(sx, sq);
    }
  }
}

template <typename T, int EPI, bool OUT_F32>
__device__ __forceinline__ void epilogue_direct(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int wave_m,
                                                int wave_n, int lane);

// EPI_PATCH_POS with fp16 token rows (the fp16 residual stream: clip/model.py:395-401 on a convert_weights model holds the conv output and the
// positional sum in fp16): acc (+ pos[1 + patch] in fp32 when a.pos is given; the image tower passes none and lets embed_ln_kernel add it: 40
// dependent L2 loads per lane in this epilogue cost 13 us per launch), one rounding, 32 rows x 64 columns at a time through a wave-private LDS patch so that a
// store instruction writes 128 contiguous bytes of 8 token rows (16 B per lane).  Rows map patch m -> token row (m / patches) * tokens +
// m % patches + 1.  The caller has passed a workgroup barrier after the last main-loop LDS read.
template <typename T>
__device__ __forceinline__ void epilogue_patch_pos_f16(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int wave_m, int wave_n, int lane,
                                                       char* patch) {
  constexpr int TM = T::TM, TN = T::TN, ROWB = 64 * 2 + 16;
  static_assert(T::WTN == 64 && TM % 2 == 0 && TN == 4, "64-column wave tiles");
  const int r16 = lane & 15, g4 = lane >> 4;
  const int rrow = lane >> 3, rcol = lane & 7;
  half_t* out = static_cast<half_t*>(a.out);
  const int n_st = n0 + wave_n * 64 + rcol * 8;
#pragma unroll
  for (int jc = 0; jc < TM / 2; ++jc) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int m = m0 + wave_m * T::WTM + (jc * 2 + jj) * 16 + r16;
      const int mm = m < a.M ? m : a.M - 1;
      const int t = mm - (mm / a.patches) * a.patches + 1;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int n = n0 + wave_n * 64 + i * 16 + g4 * 4;
        f32x4 v = acc[i][jc * 2 + jj];
        if (a.pos && n < a.N) v += *reinterpret_cast<const f32x4*>(a.pos + (int64_t)t * a.N + n);   // (a.pos: kernel argument, uniform)
        *reinterpret_cast<f16x4*>(patch + (jj * 16 + r16) * ROWB + (i * 16 + g4 * 4) * 2) = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      }
    }
    // same wave, LDS is in order: the reads below see the writes above (and the next slice's writes follow these reads)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = q * 8 + rrow;
      const f16x8 val = *reinterpret_cast<const f16x8*>(patch + row * ROWB + rcol * 16);
      const int m = m0 + wave_m * T::WTM + jc * 32 + row;
      if (m < a.M && n_st < a.N) {
        const int b = m / a.patches;
        *reinterpret_cast<f16x8*>(out + ((int64_t)b * a.tokens + (m - b * a.patches + 1)) * a.ldo + n_st) = val;
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // one 32-row slice at a time: the accumulators die as they are converted
  }
}

template <typename T, int EPI, bool OUT_F32, bool DMA_RES = false>
__device__ __forceinline__ void epilogue(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int wave_m, int wave_n,
                                         int lane, int wave, char* smem, const float2* lnp = nullptr) {
  if constexpr (EPI == EPI_RESIDUAL_FOLD16 && DMA_RES) {
    epilogue_residual_fold16_dma<T>(acc, a, m0, n0, n0 / T::BN, wave_m, wave_n, lane, wave, smem);
    return;
  }
  if constexpr (EPI == EPI_RESIDUAL_FOLD || EPI == EPI_RESIDUAL_FOLD16) {
    epilogue_residual_fold<T, EPI == EPI_RESIDUAL_FOLD16>(acc, a, m0, n0, n0 / T::BN, wave_m, wave_n, lane, wave, smem);
    return;
  }
  if constexpr (!OUT_F32 && EPI == EPI_PATCH_POS) {   // (launch_gemm checks N % 8 == 0)
    __syncthreads();
    epilogue_patch_pos_f16<T>(acc, a, m0, n0, wave_m, wave_n, lane, smem + wave * (32 * 144));
    return;
  }
  if constexpr (!OUT_F32 && EPI != EPI_PATCH_POS) {
    if ((a.N & 7) == 0 && (a.ldo & 7) == 0) {   // wave-uniform
      __syncthreads();                           // every wave is done with the main-loop LDS image
      epilogue_f16_staged<T, EPI, 2>(acc, a, m0, n0, wave_m, wave_n, lane, smem + wave * (32 * 144), lnp);
      return;
    }
  }
  epilogue_direct<T, EPI, OUT_F32>(acc, a, m0, n0, wave_m, wave_n, lane);
}

template <typename T, int EPI, bool OUT_F32>
__device__ __forceinline__ void epilogue_direct(f32x4 (&acc)[T::TN][T::TM], const KArgs& a, int m0, int n0, int wave_m,
                                                int wave_n, int lane) {
  constexpr int TM = T::TM, TN = T::TN;
  const int r16 = lane & 15, g4 = lane >> 4;
  // ---- epilogue: acc[i][j][e] = C[m = m0 + wave_m*WTM + j*16 + (lane&15)][n = n0 + wave_n*WTN + i*16 + (lane>>4)*4 + e]
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wave_m * T::WTM + j * 16 + r16;
    if (m >= a.M) continue;
    int64_t orow = m;
    const float* posrow = nullptr;
    if constexpr (EPI == EPI_PATCH_POS) {
      const int b = m / a.patches;
      const int t = m - b * a.patches + 1;
      orow = (int64_t)b * a.tokens + t;
      posrow = a.pos + (int64_t)t * a.N;
    }
    float rs = 1.f, mrs = 0.f;
    if constexpr (EPI == CLIPMI_EPI_BIAS || EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
      if (a.ln_stats) ln_row_params(a, m, rs, mrs);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int n = n0 + wave_n * T::WTN + i * 16 + g4 * 4;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if constexpr (EPI == CLIPMI_EPI_BIAS) {
        f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
        if (a.ln_stats) b -= mrs * *reinterpret_cast<const f32x4*>(a.ln_g + n);
        v = v * rs + b;
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
        const f32x4 g = a.ln_stats ? *reinterpret_cast<const f32x4*>(a.ln_g + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (OUT_F32) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = quick_gelu(__builtin_fmaf(v[e], rs, __builtin_fmaf(-mrs, g[e], b[e])));
        } else {
          const f16x4 o = quick_gelu_h(gelu_preact(v, rs, mrs, b, g));
          v = f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
        }
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_RESIDUAL || EPI == CLIPMI_EPI_BIAS_RELU || EPI == CLIPMI_EPI_BIAS_RESIDUAL16_RELU) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
        v += b;
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_RESIDUAL16_RELU) {
        const f16x4 r = *reinterpret_cast<const f16x4*>(a.residual16 + orow * a.ldo + n);
        v += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_RELU || EPI == CLIPMI_EPI_BIAS_RESIDUAL16_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_RESIDUAL) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(a.residual + orow * a.ldo + n);
        v += rr;
      }
      if constexpr (EPI == EPI_PATCH_POS) {
        if (a.pos) v += *reinterpret_cast<const f32x4*>(posrow + n);
      }
      if constexpr (OUT_F32) {
        *reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + orow * a.ldo + n) = v;
      } else {
        f16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
        *reinterpret_cast<f16x4*>(static_cast<half_t*>(a.out) + orow * a.ldo + n) = h;
      }
    }
  }
}

template <typename T, int EPI, bool OUT_F32>
__global__ __launch_bounds__(T::NT, T::OCC) void gemm_f16_kernel(const KArgs a) {
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = T::TM, TN = T::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave % T::WGM, wave_n = wave / T::WGM;

  int tile_m, tile_n;
  tile_coords(a, (a.M + T::BM - 1) / T::BM, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging addresses: thread t, instruction i writes LDS 16-B slot p = i*NT + t of the tile;
  //      slot p holds row p>>3, data chunk (p&7) ^ ((row>>1)&7)  (XOR swizzle, applied on the source)
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  // buffer descriptors start at the tile's first row: rows at or beyond M / N are out of range and read as zero
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.A + (int64_t)m0 * a.lda, ((int64_t)(a.M - m0) * a.lda) * 2);
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.W + (int64_t)n0 * a.ldw, ((int64_t)(a.N - n0) * a.ldw) * 2);
  int xoff[T::XI], woff[T::WI];
#pragma unroll
  for (int i = 0; i < T::XI; ++i) xoff[i] = ((i * (NT / 8) + srow) * (int)a.lda + schunk * 8) * 2;
#pragma unroll
  for (int i = 0; i < T::WI; ++i) woff[i] = ((i * (NT / 8) + srow) * (int)a.ldw + schunk * 8) * 2;
  const int lds_wave_off = wave * 1024;  // 64 lanes x 16 B

  // Tall tiles are at the 256-VGPR limit: keeping the XI + WI per-thread source offsets live across the loop made hipcc
  // spill them and reload seven of them from scratch in front of EVERY barrier (~1 us per K-step).  They are rebuilt from
  // one base per operand with a v_add each (volatile asm, so that it is not hoisted back out of the loop).
  const int xstep = (NT / 8) * (int)a.lda * 2, wstep = (NT / 8) * (int)a.ldw * 2;
  auto row_off = [](int base, int add) {
    int r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(base), "s"(add));
    return r;
  };
  auto stage = [&](int buf, int kt) {
    char* xs = smem + buf * T::STAGE + lds_wave_off;
    char* ws = xs + T::XBYTES;
    const int k0 = kt * BK * 2;
    if constexpr (TM > 8) {
#pragma unroll
      for (int i = 0; i < T::XI; ++i) CLIPMI_BUFFER_LOAD_LDS16(xrs, xs + i * (NT * 16), row_off(xoff[0], i * xstep), k0);
#pragma unroll
      for (int i = 0; i < T::WI; ++i) CLIPMI_BUFFER_LOAD_LDS16(wrs, ws + i * (NT * 16), row_off(woff[0], i * wstep), k0);
    } else {
#pragma unroll
      for (int i = 0; i < T::XI; ++i) CLIPMI_BUFFER_LOAD_LDS16(xrs, xs + i * (NT * 16), xoff[i], k0);
#pragma unroll
      for (int i = 0; i < T::WI; ++i) CLIPMI_BUFFER_LOAD_LDS16(wrs, ws + i * (NT * 16), woff[i], k0);
    }
  };

  // ---- fragment read offsets (bytes inside an operand tile): lane reads row (lane&15) of its 16-row tile,
  //      data chunk ks*4 + (lane>>4), stored at chunk ^ ((row>>1)&7)
  const int r16 = lane & 15, g4 = lane >> 4;
  const int swz = (r16 >> 1) & 7;
  int foff[2];
  foff[0] = r16 * 128 + (((0 + g4) ^ swz) << 4);
  foff[1] = r16 * 128 + (((4 + g4) ^ swz) << 4);
  const int xbase = wave_m * T::WTM * 128;
  const int wbase = T::XBYTES + wave_n * T::WTN * 128;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
#ifdef CLIPMI_TUNING
#ifdef CLIPMI_TUNING
  const bool stamp = a.stamps != nullptr && tid == 0;
#endif
  if (stamp) {
    a.stamps[blockIdx.x * 8 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[blockIdx.x * 8 + 5] = (long long)__smid();
  }
#endif
  stage(0, 0);
  // LayerNorm-fold consumer: the (rstd, mean*rstd) pair of each of the tile's rows, computed once here (behind the first
  // stage's DMA latency) instead of per lane per row in the epilogue; visible after the loop's first barrier
  float2* lnp = nullptr;
  if constexpr (EPI == CLIPMI_EPI_BIAS || EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
    if (a.ln_stats) {   // block-uniform
      lnp = reinterpret_cast<float2*>(smem + T::SMEM);
      for (int t = tid; t < BM; t += NT) {
        float rs, mrs;
        ln_row_params(a, m0 + t, rs, mrs);
        lnp[t] = make_float2(rs, mrs);
      }
    }
  }
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; everyone finished reading the other buffer
#ifdef CLIPMI_TUNING
    if (stamp && kt == 0) {
      a.stamps[blockIdx.x * 8 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
      a.stamps[blockIdx.x * 8 + 6] = (long long)__builtin_amdgcn_s_memtime();   // shader-clock counter: in-kernel clock of the main loop
    }
#endif
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* st = smem + (kt & 1) * T::STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (TM > 8) {
        // tall wave tiles (160 x 64: 160 accumulator registers): the activation fragments are read in two halves that
        // share registers -- with all ten live next to the accumulators the kernel needs more than the 256 VGPRs two
        // waves per SIMD may have, and hipcc spills to scratch INSIDE this loop (observed: 2.2 us per K-step)
        constexpr int H0 = TM / 2;
        f16x8 wf[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f16x8*>(st + wbase + i * 2048 + foff[ks]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j0 = h ? H0 : 0, j1 = h ? TM : H0;
          f16x8 xf[TM - H0];
#pragma unroll
          for (int j = j0; j < j1; ++j) xf[j - j0] = *reinterpret_cast<const f16x8*>(st + xbase + j * 2048 + foff[ks]);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = j0; j < j1; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j - j0], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);   // keep the second half's reads behind the first half's MFMAs
        }
      } else {
      f16x8 xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const f16x8*>(st + xbase + j * 2048 + foff[ks]);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f16x8*>(st + wbase + i * 2048 + foff[ks]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      }
    }
  }

#ifdef CLIPMI_TUNING
  if (stamp) {
    a.stamps[blockIdx.x * 8 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memtime();
  }
#endif
  epilogue<T, EPI, OUT_F32, true>(acc, a, m0, n0, wave_m, wave_n, lane, wave, smem, lnp);
#ifdef CLIPMI_TUNING
  if (a.stamps != nullptr) {
    if (stamp) a.stamps[blockIdx.x * 8 + 3] = (long long)__builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp) a.stamps[blockIdx.x * 8 + 4] = (long long)__builtin_amdgcn_s_memrealtime();
  }
#endif
}

// n-tiles per traversal band.  Measured (profiles/r01_gemm_band_sweep.txt): 4 is best or tied on every tower
// shape (fc: 298 -> 275 us against the unbanded order); up to 6 n-tiles are kept as one band.
int pick_band(int tiles_n, int /*bn*/, int /*K*/) {
  const int forced = options().gemm_band.load(std::memory_order_relaxed);
  int g = forced > 0 ? forced : (tiles_n <= 6 ? tiles_n : 4);
  if (g < 1) g = 1;
  if (g > tiles_n) g = tiles_n;
  const int nb = (tiles_n + g - 1) / g;   // even out the bands (9 n-tiles, g = 4 -> 3 bands of 3)
  return (tiles_n + nb - 1) / nb;
}

template <typename T, int EPI, bool OUT_F32>
int launch_tile(KArgs k, hipStream_t s) {
  static DeviceOnce attr_once;
  auto fn = gemm_f16_kernel<T, EPI, OUT_F32>;
  constexpr int SMEM_MAIN = T::SMEM + T::BM * (int)sizeof(float2);   // + the LayerNorm-fold row parameters
  constexpr int SMEM_EPI = (EPI == EPI_RESIDUAL_FOLD16 && T::WTN == 64) ? FoldDma<T>::LDS : 0;   // the residual tile passes through LDS
  constexpr int SMEM = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  static_assert(SMEM <= 160 * 1024, "tile does not fit the CU's LDS");
  ensure_dynamic_lds(fn, SMEM, attr_once);
  const int tiles_m = (k.M + T::BM - 1) / T::BM;
  k.tiles_n = (k.N + T::BN - 1) / T::BN;
  k.band = pick_band(k.tiles_n, T::BN, k.K);
  const int64_t nwg = (int64_t)tiles_m * k.tiles_n;
  CLIPMI_REQUIRE(nwg < (1ll << 30), CLIPMI_ERR_SHAPE, "gemm: grid too large");
  k.nwg = (int)nwg;
  hipLaunchKernelGGL(fn, dim3(k.nwg), dim3(T::NT), SMEM, s, k);
  return check_launch("gemm_f16_kernel");
}


// (rstd, mean * rstd) of every row from the producer's row partials: ln_row_params once per row and GEMM
__global__ __launch_bounds__(256) void ln_finalize_kernel(const KArgs a, float2* __restrict__ rows) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  float rs, mrs;
  ln_row_params(a, m, rs, mrs);
  rows[m] = make_float2(rs, mrs);
}

// ---------------------------------------------------------------------------------------------------------------
// Streamed-epilogue persistent kernel (fp16-out epilogues: in-proj, c_fc): 256 x 256 tile, EIGHT waves of 128(m) x 64(n)
// (two per SIMD, 256 registers each), one workgroup per CU walking its tiles.
//
// Why: phase stamps of the 16-wave kernel (tools/gemm_stamps.py, round 2) put a 256 x 256 x 768 tile at 17.1 us of main
// loop (1.5 PFLOP/s) + 1.5 us of prologue + 2.6-3.6 us of epilogue + 1-4 us until the CU's next workgroup starts: a third
// of every tile is spent with the matrix pipe idle, and because every CU reaches its epilogue at the same moment the
// 32 MB of stores of one round of tiles hit HBM as one burst.  Here the epilogue leaves the critical path:
//   * after the K loop a wave only CONVERTS its 128 x 64 outputs (bias / LayerNorm fold) to fp16 and keeps them in registers (the accumulators
//     themselves are needed for the next tile at once).  QuickGELU (round 6): what is kept is the fp16 PRE-activation; the activation itself -- two
//     transcendentals per element -- is applied between the MFMAs of the NEXT tile's compute parts (gelu_uop, stream_gelu_op below);
//   * they are stored in 16-row slices, one per K-step, inside the NEXT tile's K loop, straight from the registers (two 16-byte buffer stores per
//     lane and slice after a v_permlane16_swap interleave -- itself issued inside that K loop since round 6) -- no LDS round trip, no wait: the
//     HBM write stream is continuous instead of bursty;
//   * the next tile's first stage AND its row / column parameters are DMA'd (buffer_load ... lds) during the LAST K-step of the current tile, as every
//     other stage is during the K-step before it (round 6; until then: between the slices of the conversion, with a vmcnt wait and a workgroup
//     barrier behind it); the row parameters (rstd, mean * rstd) are finalised from the producer's row partials at the tile start (RAW mode) or come
//     from ln_finalize_kernel.
// Ablations on MI355X (tools/stream_ablate.py): a first version that sent each slice through a wave-private LDS patch at
// the top of its K-step paid 0.9 us per slice (nothing else runs on the SIMD while both of its waves wait for that
// round trip), computed the parameters of the next tile with ordinary loads (their vmcnt wait also waits for the 64 KB
// stage) and ended 10 % SLOWER than the one-tile-per-workgroup kernel; without those three costs the loop runs at
// 19 us per tile at 100 % duty.
// With 8 waves a wave has 256 registers: 128 accumulators + 64 held outputs + 48 operand fragments fit (the 16-wave
// geometry has 128 per wave: 64 + 32 held + 24 fragments + addresses do not).  The K-steps that carry a store slice are
// unrolled (straight-line code: hipcc's waitcnt pass keeps counted lgkmcnt waits), the remaining ones run in a loop.
// Needs K >= 8 * 64 (K-steps 1 .. 6 and the last one carry the held slices of the previous tile), a matrix below 2 GiB per launch (one descriptor: launch_one cuts longer ones into row ranges), an 8-column-aligned fp16 output and, with the LayerNorm fold, the finalised row parameters.
// ---------------------------------------------------------------------------------------------------------------
#ifndef CLIPMI_STREAM_HD
#define CLIPMI_STREAM_HD 2
#endif
// build-time A/B of the cache policy of the two operands' LDS-DMA (profiles/r06_stream_dma_policy.txt): an activation panel is read by the four
// workgroups of a band at about the same time and never again in that band, a weight panel by every m-tile of the band
#ifndef CLIPMI_STREAM_BIAS_SWAP
#define CLIPMI_STREAM_BIAS_SWAP 1   // build-time A/B: the bias / plain epilogues' store interleave inside the next K loop (1) or in the tile change (0)
#endif
#ifndef CLIPMI_STREAM_A_AUX
#define CLIPMI_STREAM_A_AUX 0
#endif
#ifndef CLIPMI_STREAM_W_AUX
#define CLIPMI_STREAM_W_AUX 0
#endif
#ifndef CLIPMI_STREAM_PREFETCH
#define CLIPMI_STREAM_PREFETCH 1
#endif
using TStream = Tile<256, 256, 2, 4, 2>;
constexpr int STREAM_RAW_PARTS = 4;   // row-partial slots of gemm_stream_kernel's LDS table (D <= 1024 with 256-column producer tiles)
// the streamed kernel finalises the LayerNorm row partials itself (RAW mode) while they fit its LDS table; wider producers
// (more than STREAM_RAW_PARTS column tiles) go through one ln_finalize_kernel launch per folded GEMM, which needs a scratch row
inline bool stream_raw_ok(const KArgs& k) { return k.ln_parts <= STREAM_RAW_PARTS; }

// floor(n / d) for wave-uniform n through the multiplier mg = ceil(2^32 / d) the launcher prepared (exact while n d < 2^32: checked there);
// d = 1 has no 32-bit multiplier.  Two scalar instructions instead of the ~25 of a division by a run-time divisor -- and none of its vector ones
__device__ __forceinline__ int div_magic(int n, int d, uint32_t mg) { return d == 1 ? n : (int)__umulhi((uint32_t)n, mg); }

// Where the held slices of the previous tile get their activation (QuickGELU) and leave, by K-step index (0 .. 6; STREAM_LAST = the tile's last
// K-step, whatever its number).  The activation is a stream of 64 registers x 11 instructions (gelu_uop; two registers interleaved) dealt out over
// the MFMA slots of K-steps 0 .. 6 and of the last one, 11 instructions per 8 MFMAs: slice g is complete at the end of K-step g (the last K-step:
// g = 7) and leaves in phase 3 of the K-step behind it -- K-steps 1 .. 6, the last K-step (g = 6), right behind the loop (g = 7).
// Measured on c_fc of ViT-B/16 (profiles/r06_gelu_in_compute_part.txt): the 704 instructions cost the K loop ~1.6 us per tile (2.2 cycles each: three
// quarters of their stand-alone time are hidden) against 3.2 us of the tile change they leave.  A second schedule for K >= 12 K-steps -- one
// instruction per MFMA over K-steps 0 .. 10, every slice stored inside the loop -- measured the same within 0.3 % (c_fc 222.6 against 223.2 us in an
// interleaved A/B) and cost 256 registers and a scratch slot: not kept.
constexpr int STREAM_LAST = 100;
// The stream works on one held u32x4 at a time -- the 2 + 2 registers of a block pair (i, i + 1) of a slice, NOT yet interleaved for the 16-byte
// stores --: 22 instructions for registers (0, 1), 22 for (2, 3), then the two v_permlane16_swap that pack_slice would have issued in the tile change
// (0 against 2, 1 against 3: the activation is element-wise, so it commutes with the swap).  Inside a register pair the second register runs two steps
// behind the first, so that no two transcendentals are neighbours and nothing reads the result of the instruction in front of it.
constexpr int GELU_GROUP_OPS = 46, GELU_OPS = 16 * GELU_GROUP_OPS;
constexpr int gelu_pair_seq(int w) {   // instruction w = 0 .. 21 of a register pair -> (second register ? 16 : 0) | step 0 .. 10
  constexpr int seq[22] = {0, 1, 2, 16 + 0, 3, 16 + 1, 4, 16 + 2, 5, 16 + 3, 6, 16 + 4, 7, 16 + 5, 8, 16 + 6, 9, 16 + 7, 10, 16 + 8, 16 + 9, 16 + 10};
  return seq[w];
}
template <bool GELU>
constexpr int stream_store_slice(int ksi, int nheld) {
  if (!GELU) return (ksi >= 1 && ksi <= 6 && ksi - 1 < nheld) ? ksi - 1 : -1;
  return (ksi >= 1 && ksi <= 6) ? ksi - 1 : (ksi == STREAM_LAST ? 6 : -1);
}
constexpr int stream_gelu_op(int slot) {   // first instruction of the activation stream that MFMA slot `slot` of the K loop carries: 23 per 16 MFMAs
  const int n = slot * 23 / 16;
  return n < GELU_OPS ? n : GELU_OPS;
}
static_assert(stream_gelu_op(64) == 2 * GELU_GROUP_OPS && stream_gelu_op(512) == GELU_OPS, "a slice per K-step, the whole tile in eight");

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(const KArgs a, const float2* __restrict__ ln_rows) {
  using T = TStream;
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = T::TM, TN = T::TN;   // TM = 8, TN = 4
  // LDS behind the stages: 2 x (row parameters of a tile) | 2 x ([BN] bias | [BN] g).  Row parameters are either the finalised
  // (rstd, mean * rstd) pairs of ln_finalize_kernel (ln_rows != nullptr) or -- RAW mode, ln_rows == nullptr with a.ln_stats set, up
  // to STREAM_RAW_PARTS partials -- the producer's (sum, sumsq) row partials themselves, finalised at the tile start with the
  // arithmetic of ln_row_params: the launch of ln_finalize_kernel in front of every folded GEMM (4.8 us + a launch gap, 24 per
  // image-tower step) goes away for ~100 instructions per wave and tile.
  constexpr int LNP_PAR = STREAM_RAW_PARTS * BM * 8;
  constexpr int LNP_OFF = T::SMEM, COLP_OFF = LNP_OFF + 2 * LNP_PAR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave % T::WGM, wave_n = wave / T::WGM;
  const bool raw = ln_rows == nullptr && a.ln_stats != nullptr;   // kernel arguments: uniform
  const bool fold = ln_rows != nullptr || raw;

  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  const int xoff0 = (srow * (int)a.lda + schunk * 8) * 2, woff0 = (srow * (int)a.ldw + schunk * 8) * 2;
  const int xstep = (NT / 8) * (int)a.lda * 2, wstep = (NT / 8) * (int)a.ldw * 2;
  auto row_off = [](int base, int add) {
    int r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(base), "s"(add));
    return r;
  };
  const int lds_wave_off = wave * 1024;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int swz = (r16 >> 1) & 7;
  int foff[2];
  foff[0] = r16 * 128 + (((0 + g4) ^ swz) << 4);
  foff[1] = r16 * 128 + (((4 + g4) ^ swz) << 4);
  const int xbase = wave_m * T::WTM * 128;
  const int wbase = T::XBYTES + wave_n * T::WTN * 128;
  const int nk = a.K / BK;

  // virtual block id -> tile: the XCD label's contiguous range of logical ids, bands of a.band n-tiles, m slow and n fast inside a band (tile_coords);
  // the two divisions by multipliers from the launcher
  auto coords = [&](int vb, int& m0, int& n0) {
    const int xcd = vb & 7, q = a.nwg >> 3, r = a.nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int b = div_magic(wg, a.per_band, a.mg_per_band);
    const int within = wg - b * a.per_band;
    const bool last = (b + 1) * a.band > a.tiles_n;   // the ragged last band
    const int gw = last ? a.tiles_n - b * a.band : a.band;
    const int tm = div_magic(within, gw, last ? a.mg_gw_last : a.mg_band);
    m0 = tm * BM;
    n0 = (b * a.band + (within - tm * gw)) * BN;
  };
  // ---- ONE descriptor per matrix for the whole launch (round 6): a tile's first row / column goes into the offsets -- as a scalar added to the
  // lane's VGPR offset, the operand the hardware range-checks: rows at or beyond M and weight rows at or beyond N still read as zero, output rows
  // at or beyond M are still dropped.  The per-tile descriptors of rounds 2-5 cost a 64-bit multiply-add, a 64-bit clamp (vector compares: the
  // scalar unit has none) and four live SGPRs each at every tile change; the launcher checks that a whole matrix stays below 2 GiB.
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.A, (int64_t)a.M * a.lda * 2);
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.W, (int64_t)a.N * a.ldw * 2);
  constexpr int OUT_OF_RANGE = (int)0x80000000;   // as a tile offset: every lane's sum lies beyond any descriptor (lane offsets stay below 2^31)
  // one 1 KiB piece per wave of a stage (P = 0 .. XI-1: activations, XI .. XI+WI-1: weights): the K loop issues a stage's pieces
  // one per 4-MFMA step instead of all at once -- eight waves each issuing eight LDS-DMA instructions right behind the barrier
  // queued up behind the CU's one address path (64 KB at 64 B/clk) with no MFMA in flight: that, not the LDS reads, held the
  // first version of this loop at ~50 % of the matrix rate (in-kernel clock stamps: 46 k cycles per tile against 24.6 k of MFMAs)
  // xo / wo: byte offset of the tile's first activation / weight row (scalars)
  auto stage_piece = [&](auto p_tag, int buf, int kt, int xo, int wo) {
    constexpr int P = decltype(p_tag)::value;
    char* xs = smem + buf * T::STAGE + lds_wave_off;
    const int k0 = kt * BK * 2;
    if constexpr (P < T::XI) buffer_load_lds16_aux<CLIPMI_STREAM_A_AUX>(xrs, xs + P * (NT * 16), row_off(xoff0, xo + P * xstep), k0);
    else buffer_load_lds16_aux<CLIPMI_STREAM_W_AUX>(wrs, xs + T::XBYTES + (P - T::XI) * (NT * 16), row_off(woff0, wo + (P - T::XI) * wstep), k0);
  };
  static_assert(T::XI + T::WI == 8, "eight DMA pieces per wave and stage");
  auto stage = [&](int buf, int kt, int xo, int wo) {
    stage_piece(std::integral_constant<int, 0>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 1>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 2>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 3>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 4>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 5>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 6>{}, buf, kt, xo, wo);
    stage_piece(std::integral_constant<int, 7>{}, buf, kt, xo, wo);
  };
  // row / column parameters of a tile -> LDS by DMA, one 1 KiB piece per wave (waves 0-3), through whole-array descriptors like the operands:
  // columns at or beyond N and rows beyond the last partial plane read as zero; rows at or beyond M inside a plane read the next plane's first
  // rows -- finite numbers that only ever reach output rows the store descriptor drops.
  auto params = [&](int row0, int col0, int which) {
    char* lnp = smem + LNP_OFF + which * LNP_PAR;
    char* colp = smem + COLP_OFF + which * (2 * BN * 4);
    if (wave < 2) {
      if (raw) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.ln_stats, ((int64_t)(a.ln_parts - 1) * a.ln_M + a.M) * 8);
        for (int p = 0; p < a.ln_parts; ++p)   // [parts][ln_M] float2: 2 KiB of each partial belong to this tile's rows
          CLIPMI_BUFFER_LOAD_LDS16(rs, lnp + p * (BM * 8) + wave * 1024, row_off((wave * 64 + lane) * 16, (p * a.ln_M + row0) * 8), 0);
      } else if (fold) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(ln_rows, (int64_t)a.M * 8);
        CLIPMI_BUFFER_LOAD_LDS16(rs, lnp + wave * 1024, row_off((wave * 64 + lane) * 16, row0 * 8), 0);
      }
    } else if (wave == 2) {
      if constexpr (EPI != CLIPMI_EPI_NONE) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.bias, (int64_t)a.N * 4);
        CLIPMI_BUFFER_LOAD_LDS16(rs, colp, row_off(lane * 16, col0 * 4), 0);
      }
    } else if (wave == 3) {
      if (fold) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.ln_g, (int64_t)a.N * 4);
        CLIPMI_BUFFER_LOAD_LDS16(rs, colp + BN * 4, row_off(lane * 16, col0 * 4), 0);
      }
    }
  };
  // constant parts of the parameter tables (no fold: rstd = 1, mean * rstd = 0, g = 0; no bias: 0), both parities
  for (int t = tid; t < 2 * BM; t += NT)
    if (!fold) reinterpret_cast<float2*>(smem + LNP_OFF + (t / BM) * LNP_PAR)[t % BM] = make_float2(1.f, 0.f);
  for (int t = tid; t < 2 * 2 * BN; t += NT) {
    const bool is_g = (t / BN) & 1;
    if (is_g ? !fold : EPI == CLIPMI_EPI_NONE) reinterpret_cast<float*>(smem + COLP_OFF)[t] = 0.f;
  }

  int vb = blockIdx.x;
  int m0, n0;
  coords(vb, m0, n0);
#ifdef CLIPMI_TUNING
  // energy ablation (tuning build, knob 64): every tile reads the FIRST 256 activation rows -- the panel stays in L2, nothing of A
  // comes from beyond it (results wrong)
#define STREAM_A_ROW(m) ((a.knob & 64) ? 0 : (m))
#else
#define STREAM_A_ROW(m) (m)
#endif
  int txo = STREAM_A_ROW(m0) * (int)a.lda * 2, two = n0 * (int)a.ldw * 2;   // byte offsets of the tile's operand panels
  int first_buf = 0, par = 0;
  stage(first_buf, 0, txo, two);
  params(m0, n0, 0);

  // ---- outputs of the previous tile.  The first HD 16-row slices of a wave's 128 x 64 part are stored as soon as they are
  // converted (before the next tile's K loop starts); the other TM - HD slices are HELD as fp16 in registers and leave one
  // per K-step inside the next tile's K loop (K-steps 1 .. TM - HD).  Holding all eight slices (the first version) left 24
  // registers for operand fragments: every 16-MFMA block then opened with an un-hidden LDS round trip (ds_read x 6,
  // s_waitcnt lgkmcnt(1)) and the loop ran at 48 % of the matrix rate at 2.3 GHz (in-kernel clock stamps, tools/gemm_stamps.py).
  // QuickGELU (round 6): ALL eight slices are held, as fp16 PRE-activations, and get their activation between the MFMAs of the next tile's compute
  // parts (stream_gelu_op / stream_store_slice above).  What the tile change still has to do per element is the fold and a conversion (3
  // instructions) instead of those and the activation (8.5): 4.6 -> 1.4 us per c_fc tile with nothing else running on the CU.
  constexpr bool GELU = EPI == CLIPMI_EPI_BIAS_QUICKGELU;
  constexpr int HD = GELU ? 0 : CLIPMI_STREAM_HD, NHELD = TM - HD;
  static_assert(GELU ? NHELD == 8 : (NHELD >= 1 && NHELD <= 6), "K-steps 1 .. 6 carry held slices; QuickGELU: two more behind them");
  const float neg_k = -GELU_K;
  float ga_l, ga_h, gb_l, gb_h;   // the two temporaries of each of the two registers the activation stream is working on
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 held[2][NHELD];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int j = 0; j < NHELD; ++j) held[p][j] = u32x4{0u, 0u, 0u, 0u};
  // Stores go through the descriptor of the whole output matrix: rows at or beyond M are dropped by the hardware range check.  The whole byte
  // offset goes into the VGPR operand -- only that (not the scalar offset) takes part in the range check -- as (lane constant) + (scalar: tile,
  // wave, slice and column-block part); columns at or beyond N (last n-tile when N is not a multiple of 256) are sent out of range by a select
  // (no divergent branch around the stores), and so is everything while nothing is held yet (hn0 = N: no column passes).
  // A converted slice is 4 n-blocks x 4 consecutive columns (8 bytes) per lane; v_permlane16_swap on each block pair (i, i + 1)
  // -- odd 16-lane rows of block i against even rows of block i + 1 -- leaves every lane with 8 consecutive columns: lanes
  // g4 = 0, 2 get columns 16 i + 4 g4 .. + 7 of block i, lanes g4 = 1, 3 columns 16 (i + 1) + 4 (g4 - 1) .. + 7 of block i + 1.
  // Two 16-byte stores per slice (64 contiguous bytes per row and instruction) instead of four 8-byte ones: VMEM issue, not
  // bandwidth, is what these stores cost the K loop.
  half_t* out = static_cast<half_t*>(a.out);
  const __amdgpu_buffer_rsrc_t ors = make_rsrc(out, (int64_t)a.M * a.ldo * 2);
  int hn0 = a.N;             // first column of the tile the held slices belong to
  int tso = 0;               // byte offset of that tile's first element + this wave's part of it
  const int lcol = (g4 & 1) * 16 + (g4 >> 1) * 8;          // the lane's first column inside a block pair after the swap
  const int st_lane = (r16 * (int)a.ldo + lcol) * 2;
  const int slice_bytes = 16 * (int)a.ldo * 2;
  const int wave_soff = (wave_m * T::WTM * (int)a.ldo + wave_n * 64) * 2;   // scalar
  auto pack_slice = [&](const f16x4 (&v)[TN], u32x4 (&o)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const u32x2 lo = __builtin_bit_cast(u32x2, v[2 * p]), hi = __builtin_bit_cast(u32x2, v[2 * p + 1]);
      const auto r0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
      o[p] = u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
  };
  auto store_piece = [&](int j, int p, const u32x4& v) {   // rows 16 j .. 16 j + 15 of this wave's part of the tile, block pair p
    const int col = hn0 + wave_n * 64 + lcol + p * 32;
    const int in_range = row_off(st_lane, tso + j * slice_bytes + p * 64);
    const int voff = col < a.N ? in_range : (int)0xFFFFFFF0;
    if constexpr (CLIPMI_ABLATE & 1) asm volatile("" ::"v"(v), "v"(voff));   // (energy ablation: no output stores at all)
    else __builtin_amdgcn_raw_buffer_store_b128(v, ors, voff, 0, CLIPMI_STORE_AUX);
  };
  constexpr int SPS = 2;   // stores per slice and wave

  // ---- operand fragments: inline-asm LDS reads, pinned ahead of the MFMAs that use them (cdna_hip_programming.md §5.7 form
  // (ii); same construction as attend_dense_pf in attention.hip).  Per K-step and wave: 2 k-halves x 8 activation blocks
  // (B operand, xf) x 4 weight blocks (A operand, wf) = 64 MFMAs.  LDS returns in order: the counted waits below name how many
  // reads were issued after the awaited one.
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t xo[2] = {(uint32_t)(xbase + foff[0]), (uint32_t)(xbase + foff[1])};
  const uint32_t wo[2] = {(uint32_t)(wbase + foff[0]), (uint32_t)(wbase + foff[1])};

  f32x4 acc[TN][TM];
  // ---- K loop: PING-PONG between the two waves of a SIMD (waves w and w + 4; MI355X_MICROARCH.md "Two waves per SIMD").
  // A K-step is four phases per wave (k-half ks = p >> 1, row half jh = p & 1), each a LOAD part -- the phase's operand
  // fragments (4 activation blocks, plus the 4 weight blocks when jh == 0) by pinned LDS reads, this wave's share of the next
  // stage's LDS-DMA pieces (3 + 3 + 2 over phases 0..2) or a held slice's stores (phase 3), then the wait for the reads --
  // and a COMPUTE part of 16 MFMAs on registers only, with a workgroup barrier after each part.  Group 1 (waves 4-7) runs one
  // part behind group 0, so on every SIMD one wave issues MFMAs while its partner issues memory instructions: in the first
  // version each wave interleaved both, and eight VMEM instructions (~100 issue cycles each) per wave and K-step left the
  // matrix pipe idle a third of the time (build-time ablations: no DMA -26 %, no stores -19 %, no MFMAs only -19 %).
  // Hazards (slot = one part; group 0 is in slot 8k + 2p (load) / + 2p + 1 (compute) of K-step k, group 1 one slot later):
  //   WAR  stage k + 1 is DMA'd into the buffer of stage k - 1, whose last reads (group 1, phase 3) finished before the barrier
  //        that ends slot 8k - 1; the first piece is issued in slot 8k.
  //   RAW  the last pieces are issued in slot 8k + 5; every wave waits for its own pieces at the end of slot 8k + 7 (group 0:
  //        after its compute part, group 1: after its load part), the first read of stage k + 1 is in slot 8k + 8.
  // The LAST K-step of a tile stages the NEXT tile's first stage the same way (round 6: knext = 0 with the next tile's offsets; until then
  // those eight pieces went out between the slices of the epilogue and the tile change waited for them): by the time the epilogue is
  // converted the next K loop can start at once -- no vmcnt wait and no workgroup barrier between two tiles.
  // The fragment registers are single-buffered: a wave overwrites them in its load part, after its MFMAs of the previous
  // compute part have been issued.
  const int grp = wave >> 2;   // uniform
  auto kstep = [&](auto ksi_tag, auto first_tag, auto more_tag, int kt, int knext, int nxo, int nwo) {
    constexpr int KSI = decltype(ksi_tag)::value;         // K-step index 0 .. 10, STREAM_LAST, or -1 (a K-step of the run-time loop: carries nothing)
    constexpr int SLICE = KSI < 0 ? -1 : stream_store_slice<GELU>(KSI, NHELD);   // held slice stored in phase 3 (-1: none)
    // first MFMA slot of this K-step in the activation stream (-1: none)
    constexpr int GSLOT = !GELU || KSI < 0 ? -1 : (KSI == STREAM_LAST ? 7 * 64 : KSI * 64);
    constexpr bool FIRSTK = decltype(first_tag)::value;   // the accumulators start at 0
    constexpr bool MORE = decltype(more_tag)::value;      // a stage is DMA'd during this K-step: K-step knext of the tile at (nxo, nwo)
    constexpr int NST = (SLICE >= 0 && !(CLIPMI_ABLATE & 1)) ? SPS : 0;   // stores issued behind this K-step's DMA pieces
    const int buf = (first_buf + kt) & 1;
    const uint32_t sb = lds_base + (uint32_t)(buf * T::STAGE);
    uint32_t xa0 = sb + xo[0], xa1 = sb + xo[1], wa0 = sb + wo[0], wa1 = sb + wo[1];
    f16x8 wf[4], xf[4];
    auto phase = [&](auto p_tag) {
      constexpr int P = decltype(p_tag)::value;
      constexpr int KS = P >> 1, JH = P & 1;
      // ---- load part
      {
        const uint32_t xa = KS ? xa1 : xa0;
        ds_read128<(JH * 4 + 0) * 2048>(xf[0], xa);
        ds_read128<(JH * 4 + 1) * 2048>(xf[1], xa);
        ds_read128<(JH * 4 + 2) * 2048>(xf[2], xa);
        ds_read128<(JH * 4 + 3) * 2048>(xf[3], xa);
        if constexpr (JH == 0) {
          const uint32_t wa = KS ? wa1 : wa0;
          ds_read128<0>(wf[0], wa);
          ds_read128<2048>(wf[1], wa);
          ds_read128<4096>(wf[2], wa);
          ds_read128<6144>(wf[3], wa);
        }
      }
      if constexpr (MORE && !(CLIPMI_ABLATE & 2)) {
        if constexpr (P == 0) {
          stage_piece(std::integral_constant<int, 0>{}, buf ^ 1, knext, nxo, nwo);
          stage_piece(std::integral_constant<int, 1>{}, buf ^ 1, knext, nxo, nwo);
          stage_piece(std::integral_constant<int, 2>{}, buf ^ 1, knext, nxo, nwo);
        } else if constexpr (P == 1) {
          stage_piece(std::integral_constant<int, 3>{}, buf ^ 1, knext, nxo, nwo);
          stage_piece(std::integral_constant<int, 4>{}, buf ^ 1, knext, nxo, nwo);
          stage_piece(std::integral_constant<int, 5>{}, buf ^ 1, knext, nxo, nwo);
        } else if constexpr (P == 2) {
          stage_piece(std::integral_constant<int, 6>{}, buf ^ 1, knext, nxo, nwo);
          stage_piece(std::integral_constant<int, 7>{}, buf ^ 1, knext, nxo, nwo);
        }
      }
      if constexpr (P == 3 && NST > 0) {
        store_piece(HD + SLICE, 0, held[0][SLICE]);
        store_piece(HD + SLICE, 1, held[1][SLICE]);
      }
      if constexpr (JH == 0) lgkm_wait8<0>(xf[0], xf[1], xf[2], xf[3], wf[0], wf[1], wf[2], wf[3]);
      else lgkm_wait4<0>(xf[0], xf[1], xf[2], xf[3]);
      if constexpr (P == 3) {
        if (grp == 1) wait_vmcnt<NST>();   // this wave's pieces of the next stage have landed (see RAW above)
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- compute part: registers only
      __builtin_amdgcn_s_setprio(1);
      if constexpr (CLIPMI_ABLATE & 4) {
        asm volatile("" :: "v"(wf[0]), "v"(wf[1]), "v"(wf[2]), "v"(wf[3]), "v"(xf[0]), "v"(xf[1]), "v"(xf[2]), "v"(xf[3]));
      } else {
        // MFMAs as asm with the accumulator TIED to the destination: left to the builtin, hipcc's allocator sends the results of
        // the first k-half to 64 fresh registers and brings them back in the second (v_mfma v[192:195], .., v[12:15] ...), which
        // this kernel does not have -- it spilled the held outputs to scratch.  Back-to-back MFMAs on different accumulators, or
        // accumulating in place, need no wait states; nothing but MFMAs reads an accumulator before the tile's epilogue.
        auto mfma = [&](auto m_tag) {
          constexpr int M = decltype(m_tag)::value, j = M >> 2, i = M & 3;
          mfma_tied<(FIRSTK && KS == 0)>(acc[i][JH * 4 + j], wf[i], xf[j]);
        };
        if constexpr (GSLOT >= 0) {
          // instruction n of the stream: register pair n / 22, registers 2 k (even n) and 2 k + 1 (odd n) alternate, step (n % 22) / 2
          auto uop = [&](auto n_tag) {
            constexpr int N = decltype(n_tag)::value, G = N / GELU_GROUP_OPS, W = N % GELU_GROUP_OPS;   // group G = block pair G % 2 of slice G / 2
            if constexpr (W >= 44) {
              gelu_swap<W - 44>(held[G % 2][G / 2]);
            } else {
              constexpr int C = gelu_pair_seq(W % 22), E = 2 * (W / 22) + (C >> 4), Q = C & 15;
              if constexpr (C >> 4) gelu_uop<Q, E>(held[G % 2][G / 2], gb_l, gb_h, neg_k);
              else gelu_uop<Q, E>(held[G % 2][G / 2], ga_l, ga_h, neg_k);
            }
          };
          auto group = [&](auto m_tag) {
            constexpr int M = decltype(m_tag)::value;
            mfma(m_tag);
            constexpr int S0 = stream_gelu_op(GSLOT + P * 16 + M), S1 = stream_gelu_op(GSLOT + P * 16 + M + 1);
            static_assert(S1 - S0 <= 2, "at most two instructions of the activation behind an MFMA");
            if constexpr (S1 - S0 >= 1) uop(std::integral_constant<int, S0>{});
            if constexpr (S1 - S0 == 2) uop(std::integral_constant<int, S0 + 1>{});
          };
          group(std::integral_constant<int, 0>{});  group(std::integral_constant<int, 1>{});  group(std::integral_constant<int, 2>{});
          group(std::integral_constant<int, 3>{});  group(std::integral_constant<int, 4>{});  group(std::integral_constant<int, 5>{});
          group(std::integral_constant<int, 6>{});  group(std::integral_constant<int, 7>{});  group(std::integral_constant<int, 8>{});
          group(std::integral_constant<int, 9>{});  group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
          group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{}); group(std::integral_constant<int, 14>{});
          group(std::integral_constant<int, 15>{});
        } else if constexpr (CLIPMI_STREAM_BIAS_SWAP && !GELU && KSI >= 0 && KSI < NHELD) {
          // bias / plain epilogues: held slice KSI (stored in the next K-step) is interleaved for its 16-byte stores here -- one v_permlane16_swap
          // per compute part, behind the eighth MFMA -- instead of in the tile change (pack_slice: 24 swaps of 18 cycles per wave and tile)
          mfma(std::integral_constant<int, 0>{});  mfma(std::integral_constant<int, 1>{});  mfma(std::integral_constant<int, 2>{});
          mfma(std::integral_constant<int, 3>{});  mfma(std::integral_constant<int, 4>{});  mfma(std::integral_constant<int, 5>{});
          mfma(std::integral_constant<int, 6>{});  mfma(std::integral_constant<int, 7>{});
          gelu_swap<(P & 1)>(held[P >> 1][KSI]);
          mfma(std::integral_constant<int, 8>{});
          mfma(std::integral_constant<int, 9>{});  mfma(std::integral_constant<int, 10>{}); mfma(std::integral_constant<int, 11>{});
          mfma(std::integral_constant<int, 12>{}); mfma(std::integral_constant<int, 13>{}); mfma(std::integral_constant<int, 14>{});
          mfma(std::integral_constant<int, 15>{});
        } else {
          mfma(std::integral_constant<int, 0>{});  mfma(std::integral_constant<int, 1>{});  mfma(std::integral_constant<int, 2>{});
          mfma(std::integral_constant<int, 3>{});  mfma(std::integral_constant<int, 4>{});  mfma(std::integral_constant<int, 5>{});
          mfma(std::integral_constant<int, 6>{});  mfma(std::integral_constant<int, 7>{});  mfma(std::integral_constant<int, 8>{});
          mfma(std::integral_constant<int, 9>{});  mfma(std::integral_constant<int, 10>{}); mfma(std::integral_constant<int, 11>{});
          mfma(std::integral_constant<int, 12>{}); mfma(std::integral_constant<int, 13>{}); mfma(std::integral_constant<int, 14>{});
          mfma(std::integral_constant<int, 15>{});
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (P == 3) {
        if (grp == 0) wait_vmcnt<NST>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
  };

#ifdef CLIPMI_TUNING
  const bool stamp = a.stamps != nullptr && tid == 0;
#endif
  wait_vmcnt<0>();                 // the first tile's stage 0 and parameters
  __builtin_amdgcn_s_barrier();
  while (true) {
#ifdef CLIPMI_TUNING
    if (stamp) {
      a.stamps[vb * 8 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
      a.stamps[vb * 8 + 5] = (long long)blockIdx.x;
    }
#endif
    // RAW mode: the tile's row partials have landed (behind the last K-step of the previous tile): thread t owns row t, reduces its partials
    // with the arithmetic of ln_row_params and leaves (rstd, mean * rstd) in slot 0 of the table, in place.  Read in the epilogue, a
    // whole K loop of workgroup barriers later.  Rows past M hold other rows' numbers and are never stored.
    if (raw && wave < BM / 64) {
      int row = lane;
      asm volatile("" : "+v"(row));   // rebuilt per tile: as a loop invariant the table address cost a VGPR the K loop does not have (it went to scratch)
      row += wave * 64;
      float2* rp = reinterpret_cast<float2*>(smem + LNP_OFF + par * LNP_PAR) + row;
      double ps = 0.0, pss = 0.0;
      for (int p = 0; p < a.ln_parts; ++p) {
        const float2 st = rp[p * BM];
        ps += (double)st.x;
        pss += (double)st.y;
      }
      float rs, mrs;
      ln_params_from_sums(a, ps, pss, rs, mrs);
      rp[0] = make_float2(rs, mrs);
    }
    constexpr std::false_type no{};
    constexpr std::true_type yes{};
    using I = std::integral_constant<int, -1>;
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 starts one part later
    kstep(std::integral_constant<int, 0>{}, yes, yes, 0, 1, txo, two);
#ifdef CLIPMI_TUNING
    if (stamp) {
      a.stamps[vb * 8 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
      a.stamps[vb * 8 + 6] = (long long)__builtin_amdgcn_s_memtime();
    }
#endif
    kstep(std::integral_constant<int, 1>{}, no, yes, 1, 2, txo, two);
    kstep(std::integral_constant<int, 2>{}, no, yes, 2, 3, txo, two);
    kstep(std::integral_constant<int, 3>{}, no, yes, 3, 4, txo, two);
    kstep(std::integral_constant<int, 4>{}, no, yes, 4, 5, txo, two);
    kstep(std::integral_constant<int, 5>{}, no, yes, 5, 6, txo, two);
    kstep(std::integral_constant<int, 6>{}, no, yes, 6, 7, txo, two);
    for (int kt = 7; kt < nk - 1; ++kt) kstep(I{}, no, yes, kt, kt + 1, txo, two);   // K >= 8 K-steps (checked by the launcher)

    // ---- the next tile, before this one's last K-step: its first stage and its parameters are DMA'd during that K-step.  Nothing left to
    // do: the offsets point beyond the descriptors, the pieces fetch nothing.
    const int cm0 = m0, cn0 = n0;
    [[maybe_unused]] const int cvb = vb;
    const int nvb = vb + gridDim.x;
    const bool has_next = nvb < a.nwg;
    txo = OUT_OF_RANGE;
    two = OUT_OF_RANGE;
    if (has_next) {
      vb = nvb;
      coords(vb, m0, n0);
      txo = STREAM_A_ROW(m0) * (int)a.lda * 2;
      two = n0 * (int)a.ldw * 2;
      params(m0, n0, par ^ 1);
    }
    kstep(std::integral_constant<int, STREAM_LAST>{}, no, yes, nk - 1, 0, txo, two);
    if (grp == 0) __builtin_amdgcn_s_barrier();   // ... and group 0 waits out group 1's last compute part
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the asm MFMAs' results are read by compiler-scheduled VALU code from here on
    if constexpr (GELU) {   // the last held slice of the PREVIOUS tile got its activation a moment ago
      store_piece(7, 0, held[0][7]);
      store_piece(7, 1, held[1][7]);
    }
    first_buf = ((first_buf + nk - 1) & 1) ^ 1;         // where the next tile's first stage has just landed
#ifdef CLIPMI_TUNING
    if (stamp) {
      a.stamps[cvb * 8 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
      a.stamps[cvb * 8 + 7] = (long long)__builtin_amdgcn_s_memtime();
    }
#endif
    // ---- element-wise epilogue (row / column parameters from LDS, DMA'd a whole tile ago): slices 0 .. HD-1 are stored at
    // once, slices HD .. TM-1 go into `held`.  All parameters of the tile are read up front (the fragment registers are free
    // now: 8 column vectors + 8 row pairs, one LDS round trip instead of one per slice).
    hn0 = cn0;
    tso = (cm0 * (int)a.ldo + cn0) * 2 + wave_soff;
    {
      int le = lane;
      asm volatile("" : "+v"(le));   // keeps the lane-derived offsets below out of the K loop's live ranges
      const int er16 = le & 15, eg4 = le >> 4;
      const float2* lnp = reinterpret_cast<const float2*>(smem + LNP_OFF + par * LNP_PAR);
      const float* colp = reinterpret_cast<const float*>(smem + COLP_OFF) + par * 2 * BN;
      f32x4 bb[TN], gg[TN];
      // the row pairs: all eight up front, or -- QuickGELU, whose 64 held registers leave no room for them -- one slice ahead of their use
      constexpr int NPR = GELU ? 2 : TM;
      float2 pr[NPR];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int nl = wave_n * 64 + eg4 * 4 + i * 16;
        bb[i] = *reinterpret_cast<const f32x4*>(colp + nl);
        gg[i] = *reinterpret_cast<const f32x4*>(colp + BN + nl);
      }
#pragma unroll
      for (int j = 0; j < (GELU ? 1 : TM); ++j) pr[j] = lnp[wave_m * T::WTM + j * 16 + er16];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
#ifdef CLIPMI_TUNING
        if (a.knob & 8) break;
#endif
        if constexpr (GELU) {
          if (j + 1 < TM) pr[(j + 1) & 1] = lnp[wave_m * T::WTM + (j + 1) * 16 + er16];
        }
        const float rs = pr[GELU ? (j & 1) : j].x, mrs = pr[GELU ? (j & 1) : j].y;
        f16x4 cv[TN];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          if constexpr (EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
            cv[i] = gelu_preact(acc[i][j], rs, mrs, bb[i], gg[i]);   // held as it is: the activation follows inside the next K loop
          } else {
            const f32x4 v = acc[i][j] * rs + (bb[i] - mrs * gg[i]);
            cv[i] = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          }
        }
        u32x4 pk[2];
        if (j >= HD && (GELU || CLIPMI_STREAM_BIAS_SWAP)) {   // held slices as they are: the next K loop interleaves them for the stores (QuickGELU: behind the activation)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const u32x2 lo = __builtin_bit_cast(u32x2, cv[2 * p]), hi = __builtin_bit_cast(u32x2, cv[2 * p + 1]);
            pk[p] = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        } else {
          pack_slice(cv, pk);
        }
        if (j < HD) {
          store_piece(j, 0, pk[0]);
          store_piece(j, 1, pk[1]);
        } else {
          held[0][j - HD] = pk[0];
          held[1][j - HD] = pk[1];
        }
        __builtin_amdgcn_sched_barrier(0);   // one 16-row slice at a time: the accumulators die as they are converted
      }
    }
    par ^= 1;
#ifdef CLIPMI_TUNING
    if (stamp) a.stamps[cvb * 8 + 3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if (!has_next) break;
  }
  // the last tile's held slices: nothing left to hide them behind (QuickGELU: nor their activation)
#pragma unroll
  for (int j = 0; j < NHELD; ++j) {
    if constexpr (GELU) {
      f16x4 cvl[TN];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        cvl[2 * p] = quick_gelu_h(__builtin_bit_cast(f16x4, u32x2{held[p][j][0], held[p][j][1]}));
        cvl[2 * p + 1] = quick_gelu_h(__builtin_bit_cast(f16x4, u32x2{held[p][j][2], held[p][j][3]}));
      }
      u32x4 pkl[2];
      pack_slice(cvl, pkl);
      held[0][j] = pkl[0];
      held[1][j] = pkl[1];
    } else if constexpr (CLIPMI_STREAM_BIAS_SWAP) {
      gelu_swap<0>(held[0][j]); gelu_swap<1>(held[0][j]);
      gelu_swap<0>(held[1][j]); gelu_swap<1>(held[1][j]);
    }
    store_piece(HD + j, 0, held[0][j]);
    store_piece(HD + j, 1, held[1][j]);
  }
}

#undef STREAM_A_ROW

template <int EPI>
int launch_stream(KArgs k, float2* ln_rows, hipStream_t s) {
  using T = TStream;
  constexpr int SMEM = T::SMEM + 2 * STREAM_RAW_PARTS * T::BM * 8 + 2 * 2 * T::BN * 4;
  static_assert(SMEM <= 160 * 1024, "stream kernel LDS");
  static DeviceOnce attr_once;
  auto fn = gemm_stream_kernel<EPI>;
  ensure_dynamic_lds(fn, SMEM, attr_once);
  // the XCD label of a virtual block id must not change across rounds: a grid that is a multiple of 8.  A device (partition) with fewer than 8
  // CUs never gets here: stream_offsets_ok() -- part of launch_one's `fits` and required again below -- sends it to the tile kernels.
  const int n_cu = device_cus() & ~7;
  const int tiles_m = (k.M + T::BM - 1) / T::BM;
  k.tiles_n = (k.N + T::BN - 1) / T::BN;
  k.band = pick_band(k.tiles_n, T::BN, k.K);
  const int64_t nwg = (int64_t)tiles_m * k.tiles_n;
  CLIPMI_REQUIRE(nwg < (1ll << 30), CLIPMI_ERR_SHAPE, "gemm: grid too large");
  CLIPMI_REQUIRE(stream_offsets_ok(k) && stream_whole_matrix_ok(k), CLIPMI_ERR_SHAPE, "gemm: operands too large for the 32-bit offsets of the streamed kernel");
  k.nwg = (int)nwg;
  // tile traversal by multiplication (div_magic): bands of k.band n-tiles, the last one possibly narrower
  const auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint32_t)d - 1) / (uint32_t)d) : 0u; };
  k.per_band = tiles_m * k.band;
  const int gw_last = k.tiles_n % k.band ? k.tiles_n % k.band : k.band;
  k.mg_per_band = magic(k.per_band);
  k.mg_band = magic(k.band);
  k.mg_gw_last = magic(gw_last);
  CLIPMI_REQUIRE(nwg * (int64_t)(k.per_band > k.band ? k.per_band : k.band) < (1ll << 32), CLIPMI_ERR_SHAPE, "gemm: tile grid too large for the streamed kernel's traversal");
  // RAW mode: the kernel finalises the row partials itself; with more partials than its LDS table holds they are reduced to
  // (rstd, mean * rstd) once per GEMM by ln_finalize_kernel
  const bool raw = k.ln_stats && stream_raw_ok(k);
  if (k.ln_stats && !raw) {
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((k.M + 255) / 256), dim3(256), 0, s, k, ln_rows);
    const int rc = check_launch("ln_finalize_kernel");
    if (rc) return rc;
  }
  const int grid = k.nwg < n_cu ? k.nwg : n_cu;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(T::NT), SMEM, s, k, static_cast<const float2*>(k.ln_stats && !raw ? ln_rows : nullptr));
  return check_launch("gemm_stream_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// One-tile-per-workgroup kernel with the PING-PONG main loop of gemm_stream_kernel (round 2): eight waves, two per SIMD,
// T::BM x 256 tile (the 320 x 256 tile of the residual GEMMs: 160 accumulators + 36 fragment registers per wave).  A K-step is
// four phases per wave (k-half ks = p >> 1, row half jh = p & 1): a LOAD part -- the phase's operand fragments by pinned LDS
// reads (TM / 2 activation blocks, plus the 4 weight blocks when jh == 0) and, in phases 0..2, a third of this wave's LDS-DMA
// pieces of the next stage, then the wait for the reads -- and a COMPUTE part of 2 TM MFMAs on registers only, a workgroup barrier
// after each part; waves 4-7 run one part behind waves 0-3.  Hazards as in gemm_stream_kernel (WAR: the first piece of stage
// k + 1 goes out after the barrier that ends the last reads of stage k - 1; RAW: every wave waits for its own pieces at the end of
// the K-step's last slot, one barrier before the first read).  Prologue (stage 0, LayerNorm-fold row parameters) and epilogues
// are those of gemm_f16_kernel.  Why: that kernel's compiler-scheduled loop (vmcnt(0) + __syncthreads per K-step, all DMA
// issued behind the barrier) keeps the matrix pipe 68-75 % busy on this tile; this one 77+ %.
// ---------------------------------------------------------------------------------------------------------------
// IM2COL (the patch embedding, clip/model.py:395-397): the activation operand is the fp16 NCHW image itself.  conv1 has stride = kernel = P,
// so column k = c P^2 + ky P + kx of patch row m = (b, py, px) is pixel (b, c, py P + ky, px P + kx): with P in {8, 16, 32} a 16-byte LDS slot
// (8 consecutive k) is 8 consecutive pixels of one image row and a 64-deep K-step is 64 / P whole row segments of one channel.  The per-lane
// source offset of a DMA piece is then (patch origin) + (row segment, first pixel of the lane's slot) -- one register per piece, set up once
// -- and the scalar offset of a K-step is its (channel, first row).  No im2col matrix exists; the loader moves the bytes a dense operand would.
template <typename T, int EPI, bool OUT_F32, bool IM2COL = false>
__global__ __launch_bounds__(T::NT, T::OCC) void gemm_pp_kernel(const KArgs a) {
  constexpr int BM = T::BM, NT = T::NT, TM = T::TM, TN = T::TN, H = TM / 2;
  static_assert(T::NW == 8 && TN == 4 && TM % 2 == 0 && T::WTN == 64, "ping-pong loop: eight waves of (16 TM) x 64");
  constexpr int NP = T::XI + T::WI;                       // LDS-DMA pieces per wave and stage
  constexpr int PPP = (NP + 2) / 3;                       // ... per load part (phases 0..2)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave % T::WGM, wave_n = wave / T::WGM;
  const int grp = wave >> 2;   // uniform: waves w and w + 4 share a SIMD

  int tile_m, tile_n;
  tile_coords(a, a.nwg / a.tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * T::BN;

  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  const __amdgpu_buffer_rsrc_t xrs = IM2COL ? make_rsrc(a.A, (int64_t)(a.M / a.patches) * 3 * a.im_R * a.im_R * 2)
                                            : make_rsrc(a.A + (int64_t)m0 * a.lda, ((int64_t)(a.M - m0) * a.lda) * 2);
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.W + (int64_t)n0 * a.ldw, ((int64_t)(a.N - n0) * a.ldw) * 2);
  const int xoff0 = (srow * (int)a.lda + schunk * 8) * 2, woff0 = (srow * (int)a.ldw + schunk * 8) * 2;
  const int xstep = (NT / 8) * (int)a.lda * 2, wstep = (NT / 8) * (int)a.ldw * 2;
  auto row_off = [](int base, int add) {
    int r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(base), "s"(add));
    return r;
  };
  // IM2COL: byte offset of the lane's 8 pixels of K-step 0 for each of its activation pieces (piece P = tile rows P * 64 + srow); rows at or
  // beyond M point past the descriptor and read as zero.  The lane's data chunk `schunk` is row segment schunk / (P / 8), pixels (schunk % (P / 8)) * 8 ..
  // INVARIANT (tail rows): "read as zero" leans on the raw-buffer range check seeing voffset 0xFFFFFF00 + the K-step's scalar offset as out of range,
  // which holds while num_records < 0x7FFFFF00 and the sum does not wrap (K-step offsets stay far below 2^24 for any CLIP image).  It is NOT relied on
  // for results: rows >= M are never stored, and no epilogue of this mode (EPI_PATCH_POS only) reduces over rows.  A future IM2COL epilogue that
  // does (row statistics, a fold) must clamp tail rows to a valid row instead, as epilogue_patch_pos_f16 does with its output rows.
  int xim[IM2COL ? T::XI : 1];
  int im_spc = 1, im_rps = 1;   // K-steps per channel, image rows per K-step
  if constexpr (IM2COL) {
    const int cps = a.im_P >> 3;
    const int lane_k = ((schunk / cps) * a.im_R + (schunk % cps) * 8) * 2;
#pragma unroll
    for (int i = 0; i < T::XI; ++i) {
      const int m = m0 + i * (NT / 8) + srow;
      const int b = m / a.patches, p = m - b * a.patches;
      const int py = p / a.im_G, px = p - py * a.im_G;
      xim[i] = m < a.M ? (((b * 3) * a.im_R + py * a.im_P) * a.im_R + px * a.im_P) * 2 + lane_k : (int)0xFFFFFF00;
    }
    im_spc = (a.im_P * a.im_P) >> 6;
    im_rps = 64 / a.im_P;
  }
  const int lds_wave_off = wave * 1024;
  auto stage_piece = [&](auto p_tag, int buf, int kt) {
    constexpr int P = decltype(p_tag)::value;
    if constexpr (P < NP) {
      char* xs = smem + buf * T::STAGE + lds_wave_off;
      const int k0 = kt * BK * 2;
      if constexpr (P < T::XI) {
        if constexpr (IM2COL) {
          const int c = kt / im_spc, ky0 = (kt - c * im_spc) * im_rps;   // scalar
          CLIPMI_BUFFER_LOAD_LDS16(xrs, xs + P * (NT * 16), xim[P], ((c * a.im_R + ky0) * a.im_R) * 2);
        } else {
          CLIPMI_BUFFER_LOAD_LDS16(xrs, xs + P * (NT * 16), row_off(xoff0, P * xstep), k0);
        }
      } else {
        CLIPMI_BUFFER_LOAD_LDS16(wrs, xs + T::XBYTES + (P - T::XI) * (NT * 16), row_off(woff0, (P - T::XI) * wstep), k0);
      }
    }
  };

  const int r16 = lane & 15, g4 = lane >> 4;
  const int swz = (r16 >> 1) & 7;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t f0 = (uint32_t)(r16 * 128 + (((0 + g4) ^ swz) << 4)), f1 = (uint32_t)(r16 * 128 + (((4 + g4) ^ swz) << 4));
  const uint32_t xb = (uint32_t)(wave_m * T::WTM * 128), wb = (uint32_t)(T::XBYTES + wave_n * T::WTN * 128);
  const int nk = a.K / BK;

#ifdef CLIPMI_TUNING
  const bool stamp = a.stamps != nullptr && tid == 0;
  if (stamp) {
    a.stamps[blockIdx.x * 8 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[blockIdx.x * 8 + 5] = (long long)__smid();
  }
#endif
  // ---- prologue: stage 0, and (LayerNorm-fold consumers) the tile's row parameters, behind the DMA latency
  stage_piece(std::integral_constant<int, 0>{}, 0, 0); stage_piece(std::integral_constant<int, 1>{}, 0, 0);
  stage_piece(std::integral_constant<int, 2>{}, 0, 0); stage_piece(std::integral_constant<int, 3>{}, 0, 0);
  stage_piece(std::integral_constant<int, 4>{}, 0, 0); stage_piece(std::integral_constant<int, 5>{}, 0, 0);
  stage_piece(std::integral_constant<int, 6>{}, 0, 0); stage_piece(std::integral_constant<int, 7>{}, 0, 0);
  stage_piece(std::integral_constant<int, 8>{}, 0, 0);
  static_assert(NP <= 9, "at most nine pieces per wave and stage");
  float2* lnp = nullptr;
  if constexpr (EPI == CLIPMI_EPI_BIAS || EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
    if (a.ln_stats) {   // block-uniform
      lnp = reinterpret_cast<float2*>(smem + T::SMEM);
      for (int t = tid; t < BM; t += NT) {
        float rs, mrs;
        ln_row_params(a, m0 + t, rs, mrs);
        lnp[t] = make_float2(rs, mrs);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef CLIPMI_TUNING
  if (stamp) {
    a.stamps[blockIdx.x * 8 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[blockIdx.x * 8 + 6] = (long long)__builtin_amdgcn_s_memtime();
  }
#endif

  f32x4 acc[TN][TM];
  auto kstep = [&](auto first_tag, auto more_tag, int kt) {
    constexpr bool FIRSTK = decltype(first_tag)::value;
    constexpr bool MORE = decltype(more_tag)::value;
    const int buf = kt & 1;
    const uint32_t sb = lds_base + (uint32_t)(buf * T::STAGE);
    uint32_t xa0 = sb + xb + f0, xa1 = sb + xb + f1, wa0 = sb + wb + f0, wa1 = sb + wb + f1;
    f16x8 wf[4], xf[H];
    auto phase = [&](auto p_tag) {
      constexpr int P = decltype(p_tag)::value;
      constexpr int KS = P >> 1, JH = P & 1;
      // ---- load part
      {
        const uint32_t xa = KS ? xa1 : xa0;
        ds_read128<(JH * H + 0) * 2048>(xf[0], xa);
        ds_read128<(JH * H + 1) * 2048>(xf[1], xa);
        ds_read128<(JH * H + 2) * 2048>(xf[2], xa);
        ds_read128<(JH * H + 3) * 2048>(xf[3], xa);
        if constexpr (H == 5) ds_read128<(JH * H + 4) * 2048>(xf[H - 1], xa);
        if constexpr (JH == 0) {
          const uint32_t wa = KS ? wa1 : wa0;
          ds_read128<0>(wf[0], wa);
          ds_read128<2048>(wf[1], wa);
          ds_read128<4096>(wf[2], wa);
          ds_read128<6144>(wf[3], wa);
        }
      }
      if constexpr (MORE && P < 3 && !(CLIPMI_ABLATE & 2)) {
        stage_piece(std::integral_constant<int, P * PPP + 0>{}, buf ^ 1, kt + 1);
        stage_piece(std::integral_constant<int, P * PPP + 1>{}, buf ^ 1, kt + 1);
        stage_piece(std::integral_constant<int, P * PPP + 2>{}, buf ^ 1, kt + 1);
        static_assert(PPP == 3, "three pieces per load part");
      }
      if constexpr (JH == 0) lgkm_wait4<0>(wf[0], wf[1], wf[2], wf[3]);
      lgkm_wait_x<0, H>(xf);
      if constexpr (P == 3) {
        if (grp == 1) wait_vmcnt<0>();   // this wave's pieces of the next stage have landed
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- compute part: registers only, accumulators tied to the destination (see gemm_stream_kernel)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < H; ++j) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          if constexpr (CLIPMI_ABLATE & 4) {   // (energy ablation: the loop without its MFMAs; the accumulators are only defined)
            if constexpr (FIRSTK && KS == 0) asm volatile("" : "=v"(acc[i][JH * H + j]) : "v"(wf[i]), "v"(xf[j]));
            else asm volatile("" : "+v"(acc[i][JH * H + j]) : "v"(wf[i]), "v"(xf[j]));
          } else if constexpr (FIRSTK && KS == 0)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc[i][JH * H + j]) : "v"(wf[i]), "v"(xf[j]));
          else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][JH * H + j]) : "v"(wf[i]), "v"(xf[j]));
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (P == 3) {
        if (grp == 0) wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    phase(std::integral_constant<int, 0>{});
    phase(std::integral_constant<int, 1>{});
    phase(std::integral_constant<int, 2>{});
    phase(std::integral_constant<int, 3>{});
  };

  constexpr std::false_type no{};
  constexpr std::true_type yes{};
  if (grp == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 start one part later
  kstep(yes, yes, 0);                            // nk >= 2 (checked by the launcher)
  for (int kt = 1; kt < nk - 1; ++kt) kstep(no, yes, kt);
  kstep(no, no, nk - 1);
  if (grp == 0) __builtin_amdgcn_s_barrier();   // ... and waves 0-3 wait out the last compute part of waves 4-7
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the asm MFMAs' results are read by compiler-scheduled VALU code from here on
#ifdef CLIPMI_TUNING
  if (stamp) {
    a.stamps[blockIdx.x * 8 + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    a.stamps[blockIdx.x * 8 + 7] = (long long)__builtin_amdgcn_s_memtime();
  }
#endif
  epilogue<T, EPI, OUT_F32, true>(acc, a, m0, n0, wave_m, wave_n, lane, wave, smem, lnp);
#ifdef CLIPMI_TUNING
  if (a.stamps != nullptr) {
    if (stamp) a.stamps[blockIdx.x * 8 + 3] = (long long)__builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp) a.stamps[blockIdx.x * 8 + 4] = (long long)__builtin_amdgcn_s_memrealtime();
  }
#endif
}

template <typename T, int EPI, bool OUT_F32, bool IM2COL = false>
int launch_pp(KArgs k, hipStream_t s) {
  static DeviceOnce attr_once;
  auto fn = gemm_pp_kernel<T, EPI, OUT_F32, IM2COL>;
  constexpr int SMEM_MAIN = T::SMEM + T::BM * (int)sizeof(float2);   // + the LayerNorm-fold row parameters
  constexpr int SMEM_EPI = (EPI == EPI_RESIDUAL_FOLD16 && T::WTN == 64) ? FoldDma<T>::LDS : 0;
  constexpr int SMEM = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  static_assert(SMEM <= 160 * 1024, "tile does not fit the CU's LDS");
  ensure_dynamic_lds(fn, SMEM, attr_once);
  const int tiles_m = (k.M + T::BM - 1) / T::BM;
  k.tiles_n = (k.N + T::BN - 1) / T::BN;
  k.band = pick_band(k.tiles_n, T::BN, k.K);
  const int64_t nwg = (int64_t)tiles_m * k.tiles_n;
  CLIPMI_REQUIRE(nwg < (1ll << 30), CLIPMI_ERR_SHAPE, "gemm: grid too large");
  k.nwg = (int)nwg;
  hipLaunchKernelGGL(fn, dim3(k.nwg), dim3(T::NT), SMEM, s, k);
  return check_launch("gemm_pp_kernel");
}

// (Round 4 built this loop on a four-slot ring of HALF stages -- (BM + BN) rows x 64 B, 16-row DMA pieces, three half stages in flight behind a
// counted vmcnt -- to let a piece fly one and a half K-steps instead of one: bit-identical, 5-7 % slower on every GEMM of the block.  Record:
// profiles/r04_cproj_half_stage_ring.txt.)

using T128 = Tile<128, 128, 2, 2, 2>;      // 4 waves of 64x64, 64 KiB LDS, 2 workgroups / CU
using T256w16 = Tile<256, 256, 4, 4, 4>;   // 16 waves of 64x64, 128 KiB LDS, 1 workgroup / CU, 4 waves / SIMD
using T256w8 = Tile<256, 256, 2, 4, 2>;    // 8 waves of 128x64 (the implicit-GEMM convolution)
using T320w8 = Tile<320, 256, 2, 4, 2>;    // 8 waves of 160x64: 474 tiles at M=50432, N=768 (1.85 rounds instead of 2.31)

// Tile choice among the one-tile-per-workgroup kernels: minimise  rounds x tile area x workgroups-per-CU x penalty  over the three
// configurations that won the interleaved A/B runs on MI355X (tools/gemm_ab.py, profiles/r01_gemm_ab.txt):
//   1   256 x 256, 16 waves of 64 x 64   half the L2->LDS bytes per flop of the 128^2 tile; best whenever its tile count fills
//                                         the CUs evenly
//   10  320 x 256,  8 waves of 160 x 64  ping-pong main loop (gemm_pp_kernel); for tile counts that quantise badly at 256 rows:
//                                         N = 768 at M = 50432 is 591 tiles = 2.31 rounds with (1) but 474 = 1.85 rounds here
//   0   128 x 128,  4 waves, 2 WG / CU   small problems (final projections, tiny batches)
// rounds = ceil(tiles / (CUs x workgroups per CU)); penalties are the measured per-flop slowdowns relative to (1).
// The two persistent kernels sit on top of this choice (launch_one): 13 = gemm_stream_kernel (fp16-out epilogues), 16 =
// gemm_rstream_kernel (fp16-stream residual epilogue).  Option gemm_variant (CLIPMI_GEMM_VARIANT) forces one of {0, 1, 10, 13, 16}
// where the shape allows it -- a test aid: the records of the kernel families that were measured and removed (ring, register
// pipeline, persistent 16-wave, deferred stores, 128 x 128 wave tiles) are profiles/r01_gemm_variants.txt and r02_*_ab.txt.
int pick_variant(const KArgs& k) {
  const int forced = options().gemm_variant.load(std::memory_order_relaxed);   // -1 unless a test / tuning run forces one
  if (forced == 0 || forced == 1 || forced == 13 || forced == 16) return forced;
  if (forced == 10) return k.K >= 2 * BK ? 10 : 1;
  struct Cand { int id, bm, bn, per_cu; double penalty; };
  static const Cand cands[] = {{1, 256, 256, 1, 1.00}, {10, 320, 256, 1, 1.03}, {0, 128, 128, 2, 1.12}};
  const int cus = device_cus();
  int best = 1;
  double best_cost = 1e300;
  for (const Cand& c : cands) {
    if (c.id == 10 && k.K < 2 * BK) continue;   // the ping-pong loop needs two K-steps
    const int64_t tiles = (int64_t)((k.M + c.bm - 1) / c.bm) * ((k.N + c.bn - 1) / c.bn);
    const int64_t slots = (int64_t)cus * c.per_cu;
    const int64_t rounds = (tiles + slots - 1) / slots;
    double cost = (double)rounds * c.bm * c.bn * c.per_cu * c.penalty;
    // short K (the text tower's width-512 GEMMs): the per-tile fixed cost weighs more, and the 320-row tile has 20 % fewer
    // tiles -- measured +4 % on the whole text tower at 4000-16000 prompts, equal at 1000 (tools/text_bench.py)
    if (c.id == 10 && k.K <= 512) cost *= 0.93;
    if (cost < best_cost) { best_cost = cost; best = c.id; }
  }
  return best;
}

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM 3x3 convolution (stride 1, pad 1) on NHWC fp16 activations -- ModifiedResNet's Bottleneck.conv2
// (clip/model.py:20), SURVEY f-4.  out[(b,y,x), n] = relu(bias[n] + sum_{ky,kx,c} x[b, y+ky-1, x+kx-1, c] * W[n, (ky*3+kx)*C + c]).
// Same tile machinery as gemm_f16_kernel; the only difference is the activation operand: K-step kt covers channels
// [c0, c0+64) of ONE tap (C % 64 == 0), so a thread's LDS-DMA source is its pixel row shifted by ((ky-1)*W + (kx-1)) rows,
// and a tap that falls outside the image is pointed past the end of the buffer descriptor, where the hardware returns
// zeros -- the padding costs no instruction and no im2col matrix (9x the activation bytes) is ever written.
// ---------------------------------------------------------------------------------------------------------------
struct ConvArgs {
  int H, W, C;
};

template <typename T, int EPI>
__global__ __launch_bounds__(T::NT, T::OCC) void gemm_conv3x3_kernel(const KArgs a, const ConvArgs cv) {
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, TM = T::TM, TN = T::TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave % T::WGM, wave_n = wave / T::WGM;

  int tile_m, tile_n;
  tile_coords(a, (a.M + T::BM - 1) / T::BM, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.A, (int64_t)a.M * cv.C * 2);                      // the whole activation tensor
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.W + (int64_t)n0 * a.ldw, ((int64_t)(a.N - n0) * a.ldw) * 2);
  // per DMA instruction of this thread: byte offset of its pixel row and the 9-bit mask of taps that stay inside the image
  int xrow[T::XI], xmask[T::XI], woff[T::WI];
#pragma unroll
  for (int i = 0; i < T::XI; ++i) {
    const int m = m0 + i * (NT / 8) + srow;
    const int x = m % cv.W, y = (m / cv.W) % cv.H;
    int mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      mask |= (m < a.M && yy >= 0 && yy < cv.H && xx >= 0 && xx < cv.W) ? (1 << t) : 0;
    }
    xmask[i] = mask;
    xrow[i] = (m * cv.C + schunk * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < T::WI; ++i) woff[i] = ((i * (NT / 8) + srow) * (int)a.ldw + schunk * 8) * 2;
  const int lds_wave_off = wave * 1024;
  const int cpk = cv.C / BK;   // K-steps per tap

  auto stage = [&](int buf, int kt) {
    char* xs = smem + buf * T::STAGE + lds_wave_off;
    char* ws = xs + T::XBYTES;
    const int tap = kt / cpk, c0 = (kt - tap * cpk) * BK;                       // scalar
    const int shift = (((tap / 3 - 1) * cv.W + (tap % 3 - 1)) * cv.C + c0) * 2;  // bytes, may be negative
#pragma unroll
    for (int i = 0; i < T::XI; ++i) {
      const int voff = ((xmask[i] >> tap) & 1) ? xrow[i] + shift : (int)0xFFFFFFF0;   // past the descriptor: reads as zero
      CLIPMI_BUFFER_LOAD_LDS16(xrs, xs + i * (NT * 16), voff, 0);
    }
    const int k0 = kt * BK * 2;
#pragma unroll
    for (int i = 0; i < T::WI; ++i) CLIPMI_BUFFER_LOAD_LDS16(wrs, ws + i * (NT * 16), woff[i], k0);
  };

  const int r16 = lane & 15, g4 = lane >> 4;
  const int swz = (r16 >> 1) & 7;
  int foff[2];
  foff[0] = r16 * 128 + (((0 + g4) ^ swz) << 4);
  foff[1] = r16 * 128 + (((4 + g4) ^ swz) << 4);
  const int xbase = wave_m * T::WTM * 128;
  const int wbase = T::XBYTES + wave_n * T::WTN * 128;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* st = smem + (kt & 1) * T::STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 xf[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const f16x8*>(st + xbase + j * 2048 + foff[ks]);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f16x8*>(st + wbase + i * 2048 + foff[ks]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
  }
  epilogue<T, EPI, false>(acc, a, m0, n0, wave_m, wave_n, lane, wave, smem);
}

template <typename T, int EPI>
int launch_conv_tile(KArgs k, const ConvArgs& cv, hipStream_t s) {
  static DeviceOnce attr_once;
  auto fn = gemm_conv3x3_kernel<T, EPI>;
  ensure_dynamic_lds(fn, T::SMEM, attr_once);
  const int tiles_m = (k.M + T::BM - 1) / T::BM;
  k.tiles_n = (k.N + T::BN - 1) / T::BN;
  k.band = pick_band(k.tiles_n, T::BN, k.K);
  const int64_t nwg = (int64_t)tiles_m * k.tiles_n;
  CLIPMI_REQUIRE(nwg < (1ll << 30), CLIPMI_ERR_SHAPE, "conv3x3: grid too large");
  k.nwg = (int)nwg;
  hipLaunchKernelGGL(fn, dim3(k.nwg), dim3(T::NT), T::SMEM, s, k, cv);
  return check_launch("gemm_conv3x3_kernel");
}

template <typename Tl, int EPI, bool OUT_F32>
int launch_by_variant(int variant, const KArgs& k, hipStream_t s) {
  switch (variant) {
    case 10: return launch_pp<T320w8, EPI, OUT_F32>(k, s);
    case 0: return launch_tile<T128, EPI, OUT_F32>(k, s);
    default: return launch_tile<T256w16, EPI, OUT_F32>(k, s);
  }
}

// the convolution epilogues (ModifiedResNet, SURVEY f-4) only come in the three cost-model configurations
template <int EPI, bool OUT_F32>
int launch_basic(const KArgs& k, hipStream_t s) {
  int variant = pick_variant(k);
  if (variant == 13 || variant == 16) variant = 1;
  return launch_by_variant<void, EPI, OUT_F32>(variant, k, s);
}

template <int EPI, bool OUT_F32>
int launch_one(const KArgs& k, hipStream_t s, int* parts_out, float2* ln_rows = nullptr) {
  int variant = pick_variant(k);
  const bool forced = options().gemm_variant.load(std::memory_order_relaxed) >= 0;
  // fp16-out GEMMs with at least 1.4 rounds of 256 x 256 tiles and K >= 8 K-steps: the streamed-epilogue persistent kernel
  // (the cost model only ranks the one-tile-per-workgroup kernels)
  if constexpr (!OUT_F32 && (EPI == CLIPMI_EPI_NONE || EPI == CLIPMI_EPI_BIAS || EPI == CLIPMI_EPI_BIAS_QUICKGELU)) {
    const bool fits_shape = (k.N & 7) == 0 && (k.ldo & 7) == 0 && k.K >= 8 * BK && (!k.ln_stats || ln_rows || k.ln_parts <= STREAM_RAW_PARTS) && k.ln_rs == 1 &&
                            stream_offsets_ok(k);
    const bool fits = fits_shape && stream_whole_matrix_ok(k);
    // from 1.4 rounds of tiles on (round 6; 2 rounds before the tile change was reworked): the text towers' width-512 GEMMs at 12 000-16 000 rows -- fc 376 / 504 tiles:
    // 37.8 against 39.6 us, 40.6 against 46.0; in-proj 378 tiles: 33.0 against 34.3; 282 tiles (1.1 rounds): 31.8 against 23.4, stays with the tile kernels
    // (profiles/r06_text_gemm_sweep.txt)
    const bool pays = 5 * (int64_t)((k.M + 255) / 256) * ((k.N + 255) / 256) >= 7 * (int64_t)device_cus();
    const bool wanted = variant == 13 || (!forced && pays && options().gemm_stream.load(std::memory_order_relaxed) == 1);
    if (fits_shape && !fits && wanted) {
      // a matrix beyond the 2 GiB one descriptor addresses (a text tower of several hundred thousand token rows): consecutive launches over row ranges
      // that fit, each a multiple of the tile height
      const int64_t per_row = 2 * (k.lda > k.ldo ? k.lda : k.ldo);
      int64_t rows = (((1ll << 31) - (1ll << 25)) / per_row - 256) & ~255ll;
      const int64_t max_tiles = (1ll << 16) / ((k.N + 255) / 256) * 256;      // ... and keep the tile ids inside the traversal's multipliers
      if (rows > max_tiles) rows = max_tiles;
      KArgs c = k;
      c.M = (int)rows;
      if (rows >= 256 && stream_whole_matrix_ok(c)) {
        for (int64_t r0 = 0; r0 < k.M; r0 += rows) {
          c = k;
          c.M = (int)(k.M - r0 < rows ? k.M - r0 : rows);
          c.A = k.A + r0 * k.lda;
          c.out = static_cast<half_t*>(k.out) + r0 * k.ldo;
          if (k.ln_stats) c.ln_stats = k.ln_stats + 2 * r0;
          const int rc = launch_stream<EPI>(c, ln_rows ? ln_rows + r0 : nullptr, s);
          if (rc) return rc;
        }
        return CLIPMI_OK;
      }
    }
    if (fits && wanted) {
      // A ragged last row of tiles that opens a round of its own -- ViT-L/14@336 at 64 images: c_fc is 145 x 16 tiles = 9.06 rounds of 256 persistent
      // workgroups, the tenth for 64 rows; ViT-L/14 at 128 images: 8.06 and 6.05 rounds -- goes to the one-tile-per-workgroup kernels as a launch of
      // its own (a few tens of tiles, ~10 us) and the persistent kernel runs whole rounds (profiles/r05_gemm_remainder.txt).
      const int n_cu = device_cus() & ~7;
      const int64_t tm = (k.M + 255) / 256, tn = (k.N + 255) / 256;
      const int rem = k.M % 256;
      const int split_mode = options().gemm_split_rows.load(std::memory_order_relaxed);
      bool split = !forced && rem != 0 && rem <= 128 && tm > 2 && n_cu > 0 && split_mode >= 1 &&
                   ((tm - 1) * tn + n_cu - 1) / n_cu < (tm * tn + n_cu - 1) / n_cu;
      int64_t head_rows = k.M - rem;
      // (A SHORT last round of whole tiles -- c_fc of ViT-B/16 at 256 images: 2364 tiles = 9.23 rounds -- sent to the tile kernels the same way was measured
      // in round 6 and is not done: c_fc 227.3 -> 229.8 us, profiles/r06_stream_small_steps.txt; the last tiles of a persistent launch run on an emptying chip.)
      if (!split) return launch_stream<EPI>(k, ln_rows, s);
      KArgs head = k, tail = k;
      head.M = (int)head_rows;
      int rc = launch_stream<EPI>(head, ln_rows, s);
      if (rc) return rc;
      tail.M = k.M - (int)head_rows;
      tail.A = k.A + (int64_t)head.M * k.lda;
      tail.out = static_cast<half_t*>(k.out) + (int64_t)head.M * k.ldo;
      if (k.ln_stats) tail.ln_stats = k.ln_stats + 2 * (int64_t)head.M;
      int tv = pick_variant(tail);
      if (tv == 13 || tv == 16) tv = 1;
      return launch_by_variant<void, EPI, OUT_F32>(tv, tail, s);
    }
  }
  if constexpr (EPI == EPI_RESIDUAL_FOLD16) {
    // fp16-stream residual GEMMs whose row ranges split into tall tiles: the persistent row-range kernel.  By default for K <= 1536
    // only: interleaved A/B on two boxes (profiles/r03_residual_gemm_study.txt) has it 2.4-3.9 % ahead on out-proj (12 K-steps: the
    // hidden tile ends are a fifth of the launch) and level with the tile kernel on c_proj (48 K-steps: what the tile ends return,
    // the five K-steps that carry the residual take back)
    const bool want = variant == 16 || (!forced && k.K <= 1536 && options().gemm_rstream.load(std::memory_order_relaxed) == 1);
    if (want && rstream_fits(k)) {
      *parts_out = (k.N + 255) / 256;
      return launch_rstream(k, s);
    }
  }
  if (variant == 13 || variant == 16) variant = 1;
  if (k.x16) {   // producer fold: one row partial per 256-column tile (128-column tiles: two), at most LN_MAX_PARTS of them
    const int bn = variant == 0 ? 128 : 256;
    if ((k.N + bn - 1) / bn > LN_MAX_PARTS) variant = 1;
    CLIPMI_REQUIRE((k.N + 255) / 256 <= LN_MAX_PARTS, CLIPMI_ERR_SHAPE, "gemm: N=%d has too many column tiles for the LayerNorm fold", k.N);
    *parts_out = (k.N + (variant == 0 ? 128 : 256) - 1) / (variant == 0 ? 128 : 256);
  }
  return launch_by_variant<void, EPI, OUT_F32>(variant, k, s);
}

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  CLIPMI_REQUIRE(a.A && a.W && (a.out || a.residual_f16), CLIPMI_ERR_ARG, "gemm: null operand");
  CLIPMI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, CLIPMI_ERR_SHAPE, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  CLIPMI_REQUIRE(a.K % BK == 0, CLIPMI_ERR_SHAPE, "gemm: K=%d must be a multiple of %d", a.K, BK);
  CLIPMI_REQUIRE(a.N % 4 == 0, CLIPMI_ERR_SHAPE, "gemm: N=%d must be a multiple of 4", a.N);
  CLIPMI_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 4 == 0, CLIPMI_ERR_SHAPE,
                 "gemm: leading dimensions must keep 16-byte alignment (lda=%lld ldw=%lld ldo=%lld)", (long long)a.lda,
                 (long long)a.ldw, (long long)a.ldo);
  CLIPMI_REQUIRE(a.lda >= a.K && a.ldw >= a.K && a.ldo >= a.N, CLIPMI_ERR_SHAPE, "gemm: leading dimension too small");
  CLIPMI_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.out % 16 == 0) && ((uintptr_t)a.x16 % 16 == 0),
                 CLIPMI_ERR_ARG, "gemm: operands must be 16-byte aligned");
  const bool f32 = a.out_dtype == CLIPMI_F32;
  CLIPMI_REQUIRE(f32 || a.out_dtype == CLIPMI_F16, CLIPMI_ERR_ARG, "gemm: bad out_dtype %d", a.out_dtype);

  KArgs k;
  k.A = a.A; k.lda = a.lda; k.W = a.W; k.ldw = a.ldw; k.bias = a.bias; k.residual = static_cast<const float*>(a.residual);
  k.residual16 = static_cast<const half_t*>(a.residual);
  k.out = a.out; k.ldo = a.ldo; k.M = a.M; k.N = a.N; k.K = a.K;
  k.pos = a.pos; k.patches = a.patches; k.tokens = a.tokens;
  k.im_R = a.im_R; k.im_P = a.im_P; k.im_G = a.im_P > 0 ? a.im_R / a.im_P : 0;
  k.ln_stats = a.ln_stats; k.ln_parts = a.ln_parts; k.ln_g = a.ln_g; k.ln_inv_d = a.ln_dim > 0 ? 1.0f / (float)a.ln_dim : 0.f;
  k.ln_eps = a.ln_eps; k.ln_M = a.ln_plane > 0 ? (int)a.ln_plane : a.M; k.ln_rs = a.ln_row_stride; k.x16 = a.x16; k.stats_out = a.stats_out;
  CLIPMI_REQUIRE(a.ln_row_stride >= 1 && a.ln_plane >= 0 && a.ln_plane < (1ll << 31), CLIPMI_ERR_ARG, "gemm: bad LayerNorm statistics stride");
#ifdef CLIPMI_TUNING
  k.stamps = g_tuning_stamps.load(std::memory_order_relaxed);   // clipmi_tuning_set_stamps (tools/gemm_stamps.py), tuning build only
  k.knob = g_tuning_knob.load(std::memory_order_relaxed);
#endif
  CLIPMI_REQUIRE(!a.ln_stats || (a.ln_g && a.ln_dim > 0 && a.ln_parts >= 1 && a.ln_parts <= LN_MAX_PARTS &&
                                 (a.epilogue == CLIPMI_EPI_BIAS || a.epilogue == CLIPMI_EPI_BIAS_QUICKGELU)),
                 CLIPMI_ERR_ARG, "gemm: LayerNorm fold needs ln_g, ln_dim, 1..%d partials and a BIAS / BIAS_QUICKGELU epilogue", LN_MAX_PARTS);
  CLIPMI_REQUIRE((!a.x16 && !a.stats_out) || (a.x16 && a.stats_out && a.parts_out && a.epilogue == CLIPMI_EPI_BIAS_RESIDUAL &&
                                              a.N % 8 == 0 && a.ldo % 8 == 0),
                 CLIPMI_ERR_ARG, "gemm: x16/stats_out/parts_out come together, only with BIAS_RESIDUAL and N %% 8 == 0");
  k.tiles_n = 0; k.nwg = 0;

  switch (a.epilogue) {
    case CLIPMI_EPI_NONE:
      return f32 ? launch_one<CLIPMI_EPI_NONE, true>(k, s, a.parts_out) : launch_one<CLIPMI_EPI_NONE, false>(k, s, a.parts_out, reinterpret_cast<float2*>(a.ln_rows));
    case CLIPMI_EPI_BIAS:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      return f32 ? launch_one<CLIPMI_EPI_BIAS, true>(k, s, a.parts_out) : launch_one<CLIPMI_EPI_BIAS, false>(k, s, a.parts_out, reinterpret_cast<float2*>(a.ln_rows));
    case CLIPMI_EPI_BIAS_QUICKGELU:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      return f32 ? launch_one<CLIPMI_EPI_BIAS_QUICKGELU, true>(k, s, a.parts_out) : launch_one<CLIPMI_EPI_BIAS_QUICKGELU, false>(k, s, a.parts_out, reinterpret_cast<float2*>(a.ln_rows));
    case CLIPMI_EPI_BIAS_RESIDUAL:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      if (a.residual_f16) {
        CLIPMI_REQUIRE(a.x16, CLIPMI_ERR_ARG, "gemm: residual_f16 needs the fp16 stream (x16)");
        return launch_one<EPI_RESIDUAL_FOLD16, true>(k, s, a.parts_out);
      }
      CLIPMI_REQUIRE(a.residual && (uintptr_t)a.residual % 16 == 0, CLIPMI_ERR_ARG, "gemm: residual missing/unaligned");
      CLIPMI_REQUIRE(f32, CLIPMI_ERR_ARG, "gemm: the residual stream is fp32");
      if (a.x16) return launch_one<EPI_RESIDUAL_FOLD, true>(k, s, a.parts_out);
      return launch_one<CLIPMI_EPI_BIAS_RESIDUAL, true>(k, s, a.parts_out);
    case CLIPMI_EPI_BIAS_RELU:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      return f32 ? launch_basic<CLIPMI_EPI_BIAS_RELU, true>(k, s) : launch_basic<CLIPMI_EPI_BIAS_RELU, false>(k, s);
    case CLIPMI_EPI_BIAS_RESIDUAL16_RELU:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      CLIPMI_REQUIRE(a.residual && (uintptr_t)a.residual % 8 == 0 && !f32, CLIPMI_ERR_ARG, "gemm: BIAS_RESIDUAL16_RELU needs an fp16 residual and fp16 output");
      return launch_basic<CLIPMI_EPI_BIAS_RESIDUAL16_RELU, false>(k, s);
    case EPI_PATCH_POS:
      CLIPMI_REQUIRE((a.pos || a.im_P) && a.patches > 0 && a.tokens > a.patches, CLIPMI_ERR_ARG, "gemm: bad patch epilogue");
      if (a.im_P) {   // implicit im2col: the ping-pong 320 x 256 kernel with the image as its activation operand
        CLIPMI_REQUIRE((a.im_P == 8 || a.im_P == 16 || a.im_P == 32) && a.im_R % a.im_P == 0 && a.K == 3 * a.im_P * a.im_P &&
                           a.patches == (a.im_R / a.im_P) * (a.im_R / a.im_P) && a.M % a.patches == 0 && a.N % 8 == 0 && a.ldo % 8 == 0 &&
                           (int64_t)(a.M / a.patches) * 3 * a.im_R * a.im_R * 2 < 0x7FFFFF00ll,
                       CLIPMI_ERR_SHAPE, "gemm: implicit im2col needs P in {8, 16, 32}, K = 3 P^2, whole images below 2 GB, N %% 8 == 0");
        return f32 ? launch_pp<T320w8, EPI_PATCH_POS, true, true>(k, s) : launch_pp<T320w8, EPI_PATCH_POS, false, true>(k, s);
      }
      CLIPMI_REQUIRE(f32, CLIPMI_ERR_ARG, "gemm: the im2col-matrix patch epilogue writes fp32 rows");
      return launch_one<EPI_PATCH_POS, true>(k, s, a.parts_out);
    default:
      set_error("gemm: unknown epilogue %d", a.epilogue);
      return CLIPMI_ERR_ARG;
  }
}

int launch_conv3x3(const half_t* x, const half_t* w, const float* bias, half_t* out, int B, int H, int W, int C, int Cout, int relu,
                   hipStream_t s) {
  if (B == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(x && w && bias && out, CLIPMI_ERR_ARG, "conv3x3: null pointer");
  CLIPMI_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 64 == 0 && Cout > 0 && Cout % 8 == 0, CLIPMI_ERR_SHAPE,
                 "conv3x3: B=%d H=%d W=%d C=%d Cout=%d unsupported (C %% 64 == 0, Cout %% 8 == 0)", B, H, W, C, Cout);
  CLIPMI_REQUIRE((int64_t)B * H * W * C * 2 < 0xFFFFFFF0ll && (int64_t)B * H * W < (1ll << 31), CLIPMI_ERR_SHAPE, "conv3x3: activation tensor too large for 32-bit offsets");
  CLIPMI_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)bias % 16 == 0, CLIPMI_ERR_ARG,
                 "conv3x3: pointers must be 16-byte aligned");
  KArgs k{};
  k.A = x; k.lda = C; k.W = w; k.ldw = 9 * (int64_t)C; k.bias = bias; k.out = out; k.ldo = Cout;
  k.M = B * H * W; k.N = Cout; k.K = 9 * C;
  const ConvArgs cv{H, W, C};
  if (Cout <= 128) {
    return relu ? launch_conv_tile<T128, CLIPMI_EPI_BIAS_RELU>(k, cv, s) : launch_conv_tile<T128, CLIPMI_EPI_BIAS>(k, cv, s);
  }
  return relu ? launch_conv_tile<T256w8, CLIPMI_EPI_BIAS_RELU>(k, cv, s) : launch_conv_tile<T256w8, CLIPMI_EPI_BIAS>(k, cv, s);
}

}  // namespace clipmi
