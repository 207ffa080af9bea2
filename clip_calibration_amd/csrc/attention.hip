// Multi-head self attention after the packed in-projection (reference clip/model.py:181-183 ->
// nn.MultiheadAttention -> scaled_dot_product_attention; SURVEY a-5a), head_dim 64, fp16 in/out, fp32 softmax.
//
// Three kernels by sequence length (launch_attention): attention_vision_kernel (193..200 tokens, non-causal: the image towers at
// 224 px), attention_persist_kernel (up to 224 tokens: the text tower and the other vision lengths) and attention_stream_kernel
// (257 / 577 tokens: ViT-L/14).  In all of them a (sequence, head) item is worked by one workgroup, each wave owning one 32-query
// tile; K and V of a key block (NKT*32 keys) are staged ONCE per workgroup into LDS by LDS-DMA (buffer_load ... lds; K: 128-B
// rows, XOR swizzle for ds_read_b128 row reads; V: 128-B rows, a second XOR so that ds_read_b64_tr_b16 transposed reads spread
// over the bank row).  Q fragments come from global memory, or by DMA too in the vision kernel.
//
//   S^T tile = K_tile(32 keys x 64) * Q^T           v_mfma_f32_32x32x16_f16, A = K rows, B = Q rows
//   softmax over keys                                 keys live in the 16 accumulator registers x NKT tiles of a lane
//                                                     (its query is the lane's column) -> in-register max/sum, one
//                                                     cross-half shuffle; online rescale across key blocks
//   O^T tile += V^T(32 d x 16 keys) * P^T             the S^T accumulators, packed to fp16, ARE the B operand
//                                                     (cdna_hip_programming.md §3 "accumulator tile as the next
//                                                     MFMA's operand"); A = V^T via transposed LDS reads
#include <cstring>
#include <type_traits>

#include "common.h"

namespace clipmi {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ f16x4 tr_read(const char* p) {
  const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
  return __builtin_bit_cast(f16x4, t);
}

// One key block (NKT*32 keys already in LDS) against this wave's 32 queries: S^T = K Q^T, online softmax over
// groups of GROUP key tiles, O^T += V^T P^T.  kread / vread are the lane's LDS read bases for the block.
//
// The softmax is VALU-issue bound (the first version spent ~44 issue cycles per score), so the per-score work is cut
// to: one v_max (raw scores; the 1/8 * log2(e) factor is folded into the exponent FMA), one v_fma + one v_exp, and
// the fp16 pack.  Masking code exists only in tiles that can contain a dead key (wave-uniform test); fully dead
// causal tiles are skipped.  Row sums come out of the matrix pipe: a third "V^T" tile of all ones accumulates
// sum_k P[k][q] in lacc alongside O, from the same fp16-rounded P that multiplies V.
template <int NKT, int GROUP, int DENSE = 0>
__device__ __forceinline__ void attend_block(const char* const (&kread)[4], const char* const (&vread)[2], const f16x8 (&qf)[4],
                                             int kb0, int L, int causal, int q0, int q, int hh, float& m_run,
                                             f32x16 (&oacc)[2], f32x16& lacc) {
  constexpr float C = 0.125f * LOG2E;  // softmax(s/8) = 2^(C s - C max)
  const f32x16 zero16 = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f16x8 ones = f16x8{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
#pragma unroll
  for (int g0 = 0; g0 < NKT; g0 += GROUP) {
    f32x16 s[GROUP];
    bool live[GROUP];   // wave-uniform: tile has at least one key this wave may attend to
    // ---- S^T = K Q^T
#pragma unroll
    for (int t = 0; t < GROUP; ++t) {
      const int kt = g0 + t;
      live[t] = false;
      if (kt < NKT) {
        const int k_lo = kb0 + kt * 32;
        // DENSE (vision towers: no causal mask, L > (NKT-1)*32): every tile is live and only the last can be partial,
        // so the whole group is one basic block and hipcc can run the LDS reads ahead of the MFMAs
        // DENSE == 2 (text tower, 3 tiles): every tile is computed and masked per lane -- a few wasted MFMAs buy
        // straight-line code.
        live[t] = DENSE ? true : ((k_lo < L) && !(causal && k_lo > q0 + 31));
        if (live[t]) {
          // the first MFMA of the chain takes a constant zero as its accumulator input: no VALU-written register is an MFMA source here
          // (CLIPMI_VALU_TO_MFMA_FENCE, common.h)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const f16x8 kf = *reinterpret_cast<const f16x8*>(kread[ks] + kt * 4096);
            if (ks == 0) s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0], zero16, 0, 0, 0);
            else s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[t], 0, 0, 0);
          }
          // DENSE == 3: a key block that lies entirely inside the sequence (multi-block sequences): nothing to mask
          const bool partial = DENSE == 3 ? false : DENSE == 1 ? (kt == NKT - 1) : (DENSE == 2 ? true : ((k_lo + 32 > L) || (causal && k_lo + 31 > q0)));   // wave-uniform
          if (partial) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int key = k_lo + (e & 3) + 8 * (e >> 2) + 4 * hh;
              const bool dead = DENSE == 1 ? (key >= L) : ((key >= L) || (causal && key > q));
              s[t][e] = dead ? NEG_BIG : s[t][e];
            }
          }
        }
      }
    }
    // ---- group max of the raw scores
    float mloc = NEG_BIG;
#pragma unroll
    for (int t = 0; t < GROUP; ++t) {
      if (g0 + t < NKT && live[t]) {
#pragma unroll
        for (int e = 0; e < 16; e += 2) mloc = fmaxf(fmaxf(s[t][e], s[t][e + 1]), mloc);
      }
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C);
    const float mc = m_new * C;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
    lacc[0] *= alpha;   // every row of the ones-tile holds the same sum; only register 0 is read
    CLIPMI_VALU_TO_MFMA_FENCE3(oacc[0], oacc[1], lacc);   // VALU-written accumulators are MFMA sources (SrcC) below
    // ---- P = 2^(C s - C m); O^T += V^T P^T; l += 1^T P^T
#pragma unroll
    for (int t = 0; t < GROUP; ++t) {
      const int kt = g0 + t;
      if (kt < NKT && live[t]) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          f16x8 pf;
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[j] = (half_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[t][8 * ss + j], C, -mc));
          CLIPMI_VALU_TO_MFMA_FENCE(pf);
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const f16x4 lo = tr_read(vread[dt] + kt * 4096 + ss * 2048);
            const f16x4 hi = tr_read(vread[dt] + kt * 4096 + ss * 2048 + 1024);
            f16x8 vf = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            CLIPMI_VALU_TO_MFMA_FENCE(vf);   // the two halves may have been moved together by VALU copies
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[dt], 0, 0, 0);
          }
          f16x8 one_rows = ones;
          CLIPMI_VALU_TO_MFMA_FENCE(one_rows);   // the constant may be re-materialised by a v_mov right in front of its use
          lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(one_rows, pf, lacc, 0, 0, 0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// attend_block with the LDS fragment reads PINNED AHEAD of their consumers (round 2).
//
// The ISA of attend_block as hipcc schedules it reads ONE fragment into the same four registers in front of every MFMA:
//     ds_read_b128 v[4:7] ; s_waitcnt lgkmcnt(0) ; v_mfma ... v[4:7] ; ds_read_b128 v[4:7] ; s_waitcnt lgkmcnt(0) ; v_mfma ...
// so each of the 28 S MFMAs and most of the 42 P.V MFMAs of a query tile pays a whole LDS round trip (~100+ cycles in
// front of a 32-cycle instruction; amdgpu_waves_per_eu does not change the schedule).  Here the reads are inline asm
// (cdna_hip_programming.md §5.7, form (ii): "=v" loads, later a wait statement that names every destination "+v" -- the
// data dependency keeps the consumers below the wait): the K fragments of key tile t+1 are in flight while tile t's four
// MFMAs run (two register sets), the first K tile of the next softmax group is read before the P.V phase of the current
// one, and the four transposed V reads of a 16-key step are issued before that step's exponent work, one step ahead.
// LDS operations return in order, so the counted lgkmcnt(N) waits below only assume that the N youngest operations are
// the ones issued after the awaited set; an LDS operation hipcc adds on its own (the cross-half shuffle) makes a wait
// stricter, never weaker.  Dense single-block form only (vision towers: no causal mask), with (NKT-1)*32 < L <= (NKT-1)*32 + 8:
// the last key tile holds at most 8 live keys (193..200 tokens at NKT = 7).
// ---------------------------------------------------------------------------------------------------------------
// CLIPMI_ATTN_ABLATE (build-time, diagnostic builds only: results are wrong with any bit set; tools/attn_ablate.sh): 1 no v_exp, 2 no P.V /
// row-sum MFMAs, 4 no S MFMAs, 8 query waves 4-6 idle (one query wave per SIMD), 16 no row-sum MFMA,
// 32 no max phase, 64 no LDS fragment reads, 128 operands read as if every (sequence, head) held K | V | Q contiguously (what a head-major
// in-projection output would give the loader), 256 no output stores,
// 512 operand DMA marked non-temporal, 1024 output stores non-temporal, 2048 output stores write-through (sc0 sc1)
#ifndef CLIPMI_ATTN_ABLATE
#define CLIPMI_ATTN_ABLATE 0
#endif
#ifdef CLIPMI_TUNING
#define CLIPMI_ATTN_STAMP(slot, dep) do { asm volatile("" :: "v"(dep)); if (sp) sp[slot] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CLIPMI_ATTN_STAMP(slot, dep) do { } while (0)
#endif
#if CLIPMI_ATTN_ABLATE & 64
#define CLIPMI_DS_READ_B128(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr), "n"(off))
#define CLIPMI_DS_READ_TR16_B64(dst, addr, off) asm volatile("" : "=v"(dst) : "v"(addr), "n"(off))
#else
#define CLIPMI_DS_READ_B128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define CLIPMI_DS_READ_TR16_B64(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#endif

template <int N>
__device__ __forceinline__ void lds_wait4(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4h(f16x4& a, f16x4& b, f16x4& c, f16x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}

template <int KT>
__device__ __forceinline__ void read_k(f16x8 (&dst)[4], const uint32_t (&ka)[4]) {
  CLIPMI_DS_READ_B128(dst[0], ka[0], KT * 4096);
  CLIPMI_DS_READ_B128(dst[1], ka[1], KT * 4096);
  CLIPMI_DS_READ_B128(dst[2], ka[2], KT * 4096);
  CLIPMI_DS_READ_B128(dst[3], ka[3], KT * 4096);
}
template <int OFF>
__device__ __forceinline__ void read_v(f16x4 (&dst)[4], const uint32_t (&va)[2]) {
  CLIPMI_DS_READ_TR16_B64(dst[0], va[0], OFF);
  CLIPMI_DS_READ_TR16_B64(dst[1], va[0], OFF + 1024);
  CLIPMI_DS_READ_TR16_B64(dst[2], va[1], OFF);
  CLIPMI_DS_READ_TR16_B64(dst[3], va[1], OFF + 1024);
}

// ka[ks] / va[dt]: the lane's LDS byte addresses (key tile 0) of its K fragment ks and of its transposed V reads for d-tile dt.
// (Measured and removed in round 3, same bits: all exponent work of a group before its P.V MFMAs as one block, and those
// segments separated by workgroup barriers with waves 4-6 one segment behind -- profiles/r02_attention_segments_ab.txt.)
template <int NKT, int GROUP>
__device__ __forceinline__ void attend_dense_pf(const uint32_t (&ka)[4], const uint32_t (&va)[2], const uint32_t (&qa)[4], int L,
                                                int hh, f32x16 (&oacc)[2], f32x16& lacc, long long* sp = nullptr) {
  constexpr float C = 0.125f * LOG2E;
  const f16x8 ones = f16x8{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
  const f32x16 zero16 = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f16x8 kf[2][4];
  f16x4 vf[2][4];   // [buffer][dt * 2 + (lo | hi)]
  float m_run = NEG_BIG;
  f16x8 qf[4];      // this wave's 32 query rows (B operand of S^T = K Q^T), read like a K tile: qa = the lane's LDS addresses
  read_k<0>(qf, qa);
  read_k<0>(kf[0], ka);

  auto group = [&](auto g0_tag) {
    constexpr int G0 = decltype(g0_tag)::value;
    constexpr int G = NKT - G0 < GROUP ? NKT - G0 : GROUP;
    constexpr bool MORE = G0 + G < NKT;
    f32x16 s[G];
    // ---- S^T = K Q^T, the next tile's fragments in flight
    auto s_tile = [&](auto t_tag) {
      constexpr int T = decltype(t_tag)::value;
      constexpr int KT = G0 + T;
      constexpr int CUR = KT & 1;
      if constexpr (T + 1 < G) {
        read_k<KT + 1>(kf[CUR ^ 1], ka);
        lds_wait4<4>(kf[CUR][0], kf[CUR][1], kf[CUR][2], kf[CUR][3]);
        if constexpr (KT == 0) lds_wait4<4>(qf[0], qf[1], qf[2], qf[3]);   // older than the K reads: already back; names the registers
      } else {
        lds_wait4<0>(kf[CUR][0], kf[CUR][1], kf[CUR][2], kf[CUR][3]);
      }
      if constexpr (CLIPMI_ATTN_ABLATE & 4) {
        asm volatile("" :: "v"(kf[CUR][0]), "v"(kf[CUR][1]), "v"(kf[CUR][2]), "v"(kf[CUR][3]));
        s[T] = zero16;
        asm volatile("" : "+v"(s[T]));
      } else {
        s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[CUR][0], qf[0], zero16, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[CUR][ks], qf[ks], s[T], 0, 0, 0);
      }
      if constexpr (KT == NKT - 1) {   // the only tile that can hold keys at or beyond L
        // L <= (NKT - 1) * 32 + 8 (the caller's contract: 193..200 tokens): registers e >= 4 of this tile are keys >= L in EVERY
        // lane (key = 32 KT + (e & 3) + 8 (e >> 2) + 4 hh) -- P = 0 exactly, so they get no mask, no max, no exponent, and the
        // tile's second 16-key step no MFMAs (adding 0 . V leaves the accumulators bit for bit as they are)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = KT * 32 + e + 4 * hh;
          s[T][e] = key >= L ? NEG_BIG : s[T][e];
        }
      }
    };
    s_tile(std::integral_constant<int, 0>{});
    if constexpr (G > 1) s_tile(std::integral_constant<int, 1>{});
    if constexpr (G > 2) s_tile(std::integral_constant<int, 2>{});
    if constexpr (G > 3) s_tile(std::integral_constant<int, 3>{});
    static_assert(G <= 4, "softmax groups of at most four key tiles");
    CLIPMI_ATTN_STAMP(G0 == 0 ? 4 : 6, s[G - 1][0]);
    // the reads that the P.V phase (and the next group's first S tile) open with: behind the exponent work by the time they are needed
    if constexpr (MORE) read_k<G0 + G>(kf[(G0 + G) & 1], ka);
    read_v<G0 * 4096>(vf[0], va);
    // ---- group max of the raw scores, online rescale (nothing to rescale in the first group)
    float mloc = NEG_BIG;
    if constexpr (CLIPMI_ATTN_ABLATE & 32) {
      mloc = 40.f;
    } else {
#pragma unroll
      for (int t = 0; t < G; ++t)
#pragma unroll
        for (int e = 0; e < ((G0 + t == NKT - 1) ? 4 : 16); e += 2) mloc = fmaxf(fmaxf(s[t][e], s[t][e + 1]), mloc);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    }
    const float m_new = G0 == 0 ? mloc : fmaxf(m_run, mloc);
    if constexpr (G0 > 0) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
      lacc[0] *= alpha;
      CLIPMI_VALU_TO_MFMA_FENCE3(oacc[0], oacc[1], lacc);   // VALU-written accumulators are MFMA sources (SrcC) below
    }
    const float mc = m_new * C;
    m_run = m_new;
    // ---- P = 2^(C s - C m); O^T += V^T P^T; l += 1^T P^T
    auto pv_step = [&](auto t_tag, auto ss_tag) {
      constexpr int T = decltype(t_tag)::value, SS = decltype(ss_tag)::value;
      constexpr int STEP = T * 2 + SS, CUR = STEP & 1;
      constexpr int NSTEPS = 2 * G - ((G0 + G == NKT) ? 1 : 0);   // the tail tile's second 16-key step is all dead keys: skipped
      if constexpr (STEP >= NSTEPS) return;
      constexpr bool LAST = STEP == NSTEPS - 1;
      constexpr bool TAILT = G0 + T == NKT - 1;
      if constexpr (!LAST) read_v<(G0 + (STEP + 1) / 2) * 4096 + ((STEP + 1) & 1) * 2048>(vf[CUR ^ 1], va);
      f16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pf[j] = (TAILT && 8 * SS + j >= 4) ? (half_t)0.f
                : (CLIPMI_ATTN_ABLATE & 1) ? (half_t)__builtin_fmaf(s[T][8 * SS + j], C, -mc)
                                           : (half_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[T][8 * SS + j], C, -mc));
      if constexpr (!LAST) lds_wait4h<4>(vf[CUR][0], vf[CUR][1], vf[CUR][2], vf[CUR][3]);
      else lds_wait4h<0>(vf[CUR][0], vf[CUR][1], vf[CUR][2], vf[CUR][3]);
      CLIPMI_VALU_TO_MFMA_FENCE(pf);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const f16x4 lo = vf[CUR][dt * 2], hi = vf[CUR][dt * 2 + 1];
        f16x8 v8 = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        CLIPMI_VALU_TO_MFMA_FENCE(v8);   // the two halves may have been moved together by VALU copies
        if constexpr (CLIPMI_ATTN_ABLATE & 2) {
          asm volatile("" :: "v"(v8), "v"(pf));
          if (G0 == 0 && STEP == 0) { oacc[dt] = zero16; asm volatile("" : "+v"(oacc[dt])); }
        } else if (G0 == 0 && STEP == 0) oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v8, pf, zero16, 0, 0, 0);
        else oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v8, pf, oacc[dt], 0, 0, 0);
      }
      if constexpr (CLIPMI_ATTN_ABLATE & (2 | 16)) {
        asm volatile("" :: "v"(pf));
        if (G0 == 0 && STEP == 0) { lacc = zero16; lacc[0] = 1.f; asm volatile("" : "+v"(lacc)); }
      } else {
        f16x8 one_rows = ones;
        CLIPMI_VALU_TO_MFMA_FENCE(one_rows);   // the constant may be re-materialised by a v_mov right in front of its use
        if (G0 == 0 && STEP == 0) lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(one_rows, pf, zero16, 0, 0, 0);
        else lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(one_rows, pf, lacc, 0, 0, 0);
      }
    };
    auto pv_tile = [&](auto t_tag) {
      pv_step(t_tag, std::integral_constant<int, 0>{});
      pv_step(t_tag, std::integral_constant<int, 1>{});
    };
    pv_tile(std::integral_constant<int, 0>{});
    if constexpr (G > 1) pv_tile(std::integral_constant<int, 1>{});
    if constexpr (G > 2) pv_tile(std::integral_constant<int, 2>{});
    if constexpr (G > 3) pv_tile(std::integral_constant<int, 3>{});
  };
  group(std::integral_constant<int, 0>{});
  CLIPMI_ATTN_STAMP(5, oacc[0][0]);
  if constexpr (NKT > GROUP) group(std::integral_constant<int, GROUP>{});
  static_assert(NKT <= 2 * GROUP, "at most two softmax groups");
}

// O^T tile (64 d x 32 queries) -> the 32 query rows of `out`, 128 B each.  A lane holds 8 groups of 4 consecutive d of ITS query
// (group g = dt * 4 + rr: d = 8 g + 4 hh + e), the lane 32 further on the other 4 of every group.  One v_permlane32_swap per
// register and group pair (cdna_hip_programming.md T21) gives the lower half-wave d = 8k .. 8k+7 of the even groups and the
// upper half-wave those of the odd groups: four 16-byte stores per lane instead of eight 8-byte ones (the tail is store-ISSUE
// bound).  Every lane of the wave must call this (the swap is a cross-lane exchange); `valid` masks the stores only.
__device__ __forceinline__ void store_out(half_t* orow, const f32x16 (&oacc)[2], float l_run, int hh, bool valid = true) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const float inv = 1.0f / l_run;
  u32x2 o2[8];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(oacc[dt][rr * 4 + e] * inv);
      o2[dt * 4 + rr] = __builtin_bit_cast(u32x2, o);
    }
  char* row = reinterpret_cast<char*>(orow) + (hh ? 16 : 0);
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    u32x2 a = o2[k], b = o2[k + 1];
    auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    if (valid) *reinterpret_cast<u32x4*>(row + 16 * k) = u32x4{r0[0], r1[0], r0[1], r1[1]};
  }
}

// The same tile as full 128-byte lines: store_out's four instructions each write 32 bytes of 32 different rows, so every output line
// reaches the L2 as four partial writes, and the 77 MB of output cost the vision kernel a quarter of its launch for a quarter of its
// bytes (profiles/r03_attention_ablation.txt).  Here the wave parks its tile in LDS -- `stage` = the byte address of 32 rows x 128 B
// that only this wave touches (its own Q rows of the item: read at the start of the item, dead since), 16-byte pieces XOR-swizzled by
// ((row >> 1) & 7) like the K rows -- and reads it back with 8 consecutive lanes per row: one store instruction = 8 complete rows.
// Rows at or beyond `nrows` (relative to the tile) are neither parked nor stored.  Same bits as store_out.
template <bool NT>
__device__ __forceinline__ void store_out_lines(half_t* tile_row0, int64_t row_stride_halves, uint32_t stage, const f32x16 (&oacc)[2],
                                                float l_run, int lane, int nrows) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const int r32 = lane & 31, hh = lane >> 5;
  const float inv = 1.0f / l_run;
  u32x2 o2[8];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      f16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)(oacc[dt][rr * 4 + e] * inv);
      o2[dt * 4 + rr] = __builtin_bit_cast(u32x2, o);
    }
  const uint32_t wrow = stage + (uint32_t)(r32 * 128);
  const int wswz = (r32 >> 1) & 7;
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    u32x2 a = o2[k], b = o2[k + 1];
    auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
    const u32x4 piece = u32x4{r0[0], r1[0], r0[1], r1[1]};          // 16-byte piece (k + hh) of row r32
    if (r32 < nrows) asm volatile("ds_write_b128 %0, %1" :: "v"(wrow + (uint32_t)(((k + hh) ^ wswz) << 4)), "v"(piece) : "memory");
  }
  const int p = lane & 7, rl = lane >> 3;
  u32x4 line[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = rl + 8 * j;
    asm volatile("ds_read_b128 %0, %1" : "=v"(line[j]) : "v"(stage + (uint32_t)(r * 128 + ((p ^ ((r >> 1) & 7)) << 4))) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(line[0]), "+v"(line[1]), "+v"(line[2]), "+v"(line[3]) :: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = rl + 8 * j;
    u32x4* dst = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(tile_row0 + r * row_stride_halves) + 16 * p);
    if (r < nrows) {
      if constexpr (NT || (CLIPMI_ATTN_ABLATE & 1024)) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(dst), "v"(line[j]) : "memory");
      else if constexpr (CLIPMI_ATTN_ABLATE & 2048) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(line[j]) : "memory");
      else *dst = line[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent variant for sequences that fit one key block (L <= NKT*32; every CLIP tower at 224 px and the text
// tower): one workgroup per CU walks the (sequence, head) items; K/V of item i+1 are DMA'd into the second half of a
// double-buffered LDS image (and its Q fragments loaded into a spare register set) while item i is computed, so
// the HBM/L2 latency of the operands never sits on the critical path.  Per item: vmcnt(0) (loads issued a whole item
// ago) -> barrier -> issue loads for the next item -> compute -> store.
// ---------------------------------------------------------------------------------------------------------------
template <int NKT, int GROUP, int DENSE>
__global__ __launch_bounds__(512, 2) void attention_persist_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                                   int L, int H, int causal, int n_items) {
  constexpr int KEYS = NKT * 32;
  constexpr int OPB = KEYS * 128;      // one operand image
  constexpr int BUF = 2 * OPB;         // K + V
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthr >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  const int q0 = wave * 32;
  const bool active = q0 < L;
  const int q = q0 + r32;
  const int qc = q < L ? q : L - 1;

  const int kswz = (r32 >> 1) & 7;
  int kro[4], vro[2];   // lane-constant byte offsets inside a buffer
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kro[ks] = r32 * 128 + (((2 * ks + hh) ^ kswz) << 4);
  {
    const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int fq = (qq >> 1) & 1;
    const int lane_base = OPB + hh * 512 + qq * 128 + ((((lane >> 4) & 1) * 2 + (pp >> 1)) << 4) + (pp & 1) * 8;
    vro[0] = lane_base + fq * 64;
    vro[1] = lane_base + (1 - fq) * 64;
  }

  auto item_base = [&](int item) {
    const int n = item / H, h = item - n * H;
    return qkv + (int64_t)n * L * ld + h * 64;
  };
  auto stage = [&](int item, int buf) {
    const int h = item % H;
    const half_t* base = item_base(item);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, ((int64_t)L * ld - h * 64) * 2);
    char* Ks = smem + buf * BUF;
    for (int it = wave; it < KEYS / 8; it += nwaves) {
      const int pw = it * 64;
      const int p = pw + lane;
      const int row = p >> 3, cs = p & 7;
      const int koff = (row * (int)ld + D) * 2;      // rows >= L are outside the descriptor: zero
      CLIPMI_BUFFER_LOAD_LDS16(rs, Ks + pw * 16, koff + ((cs ^ ((row >> 1) & 7)) << 4), 0);
      CLIPMI_BUFFER_LOAD_LDS16(rs, Ks + OPB + pw * 16, koff + D * 2 + ((cs ^ (((row >> 1) & 1) << 2)) << 4), 0);
    }
  };
  auto load_q = [&](int item, f16x8 (&dst)[4]) {
    const half_t* base = item_base(item);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst[ks] = *reinterpret_cast<const f16x8*>(base + (int64_t)qc * ld + ks * 16 + hh * 8);
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  f16x8 qf[4], qn[4];
  stage(item, 0);
  load_q(item, qf);
  int buf = 0;
  for (; item < n_items; item += gridDim.x, buf ^= 1) {
    const int next = item + gridDim.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // this item's K/V visible to all waves; everybody is done with the other buffer
    // qf was loaded one item ago and is complete (vmcnt(0) above).  Make that visible to hipcc: without this fence it
    // waits vmcnt(0) at the first MFMA that reads qf -- AFTER the prefetch below has been issued -- and the prefetch
    // of the next item's K/V/Q ends up on the critical path.
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    if (next < n_items) {
      stage(next, buf ^ 1);
      load_q(next, qn);
    }
    if (active) {
      const char* b = smem + buf * BUF;
      const char* const kread[4] = {b + kro[0], b + kro[1], b + kro[2], b + kro[3]};
      const char* const vread[2] = {b + vro[0], b + vro[1]};
      f32x16 oacc[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
      float m_run = NEG_BIG;
      f32x16 lacc;
#pragma unroll
      for (int e = 0; e < 16; ++e) lacc[e] = 0.f;
      attend_block<NKT, GROUP, DENSE>(kread, vread, qf, 0, L, causal, q0, q, hh, m_run, oacc, lacc);
      if (q < L) {
        const int n = item / H, h = item - n * H;
        store_out(out + ((int64_t)n * L + q) * D + h * 64, oacc, lacc[0], hh);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Vision kernel (non-causal, 193 <= L <= 200: ViT-B/16 and B/32-style towers at 224 px, 197 tokens, 199 with MaPLe's
// prompts): the persistent kernel above with EVERY operand of the next item -- K, V and now Q too -- brought in by
// LDS-DMA from dedicated loader wave(s), so that a query wave's instruction stream holds no VMEM load at all.
//
// Why (tools/attn_stamps.py + the ISA, round 2): in the kernel above a query wave issues its share of the K/V prefetch and
// its own next Q fragments right after the item's barrier, and hipcc then waits for them at once -- it copies the freshly
// loaded Q registers into their home registers behind an s_waitcnt vmcnt(0), and because VMEM retires in order that wait
// also covers the K/V DMA issued just before.  Every item therefore paid the whole load latency (1.3-1.9 us of a 5.7 us
// item) in front of its first MFMA: nothing was prefetched.  With no VMEM load left in the query waves there is nothing
// for the compiler to wait on (stores need no wait, raw s_barrier), and the loaders run a whole item ahead.  (Query waves
// that issue LDS-DMA themselves do not work either: hipcc puts an s_waitcnt vmcnt(0) in front of the ds_read_b64_tr_b16
// of the P.V product -- the builtin carries no address-space-precise memory operand, so it may alias the pending DMA.)
//
// LDS (dynamic; the tail pad zeroed once): 2 buffers x (K | V | Q) x 200 rows x 128 B = 153,600 B + 3 KiB tail pad.  The seventh key /
// query tile spans rows 192-223; rows 200-223 of an array fall into the NEXT array.  For K and V that is always finite data
// written by this item's DMA (the V array behind K, the Q array behind V; rows at or beyond L lie outside the DMA descriptor
// and read as zero): keys >= L are masked in the scores and their P = 0 annihilates whatever V row is read.  Behind the Q
// array lie the other buffer's K rows (in flight, or untouched during the very first item) or the tail pad: they are read only
// as QUERY rows >= 200 -- columns of S^T and O^T that no other column depends on and that are never stored.
// ---------------------------------------------------------------------------------------------------------------
constexpr int VROWS = 200;
constexpr int VARR = VROWS * 128;            // one operand image
constexpr int VBUF = 3 * VARR;               // K | V | Q
constexpr int VSMEM = 2 * VBUF + 24 * 128;   // + tail pad for the overrun of the last array

#ifdef CLIPMI_TUNING
#define CLIPMI_VISION_STAMPS_PARAM , long long* stamps   // diagnostic build: [item][wave 0..7 (7 = loader)][8]
#define CLIPMI_VISION_STAMPS_ARG , stamps
#else
#define CLIPMI_VISION_STAMPS_PARAM
#define CLIPMI_VISION_STAMPS_ARG
#endif
template <bool NT>   // NT: output rows stored non-temporal
__device__ __forceinline__ void attention_vision_body(const half_t* __restrict__ qkv, half_t* __restrict__ out, int L, int H, int n_items CLIPMI_VISION_STAMPS_PARAM) {
  constexpr int NKT = 7, GROUP = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..6 query waves, 7.. loader waves
  const int r32 = lane & 31, hh = lane >> 5;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  // Only the tail pad is never written by the DMA (every item rewrites rows 0 .. 199 of its three arrays, zeros beyond L): it is read as the
  // query rows >= 200 of buffer 1's last tile, whose outputs are never stored -- zeroed all the same, so that no NaN pattern left by an
  // earlier kernel enters an MFMA.  The first item's barrier orders these stores before any read.  (Until round 3 the whole 157 KB image was
  // zeroed here, with a workgroup barrier, in front of the first DMA: ~1 us per launch.)
  if (tid * 16 < VSMEM - 2 * VBUF) *reinterpret_cast<f32x4*>(smem + 2 * VBUF + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};

  int item = blockIdx.x;
  if (item >= n_items) return;

  if (wave >= 7) {
    // ---- loader wave: ALL DMA of item (i + 1) while the query waves work on item i -- 25 groups of 8 rows x 128 B per
    // operand.  Per-lane source offsets are precomputed (a group's swizzle depends on its parity only), so a DMA costs one
    // v_add + the M0 update + the instruction itself.  (Two loader waves, raised loader priority: measured equal, removed.)
    const int lr = lane >> 3, cs = lane & 7;
    const int swv = (cs ^ (((lr >> 1) & 1) << 2)) << 4;                                        // V: chunk ^ (((row >> 1) & 1) << 2)
    const int swk[2] = {(cs ^ (lr >> 1)) << 4, (cs ^ (4 + (lr >> 1))) << 4};                    // K, Q: chunk ^ ((row >> 1) & 7), by group parity
    constexpr bool FAKE = (CLIPMI_ATTN_ABLATE & 128) != 0;
    const int lane_row = lr * (FAKE ? 64 : (int)ld) * 2;
    const int gstep = 8 * (FAKE ? 64 : (int)ld) * 2;                                            // one group further
    auto radd = [](int base, int add) {
      int r;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(base), "s"(add));
      return r;
    };
    auto stage = [&](int it_, int buf) {
      const int n = it_ / H, h = it_ - n * H;
      const half_t* base = FAKE ? qkv + (int64_t)it_ * 3 * L * 64 : qkv + (int64_t)n * L * ld + h * 64;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, FAKE ? (int64_t)3 * L * 64 * 2 : ((int64_t)L * ld - h * 64) * 2);   // rows >= L: outside, read as zero
      const int koff = FAKE ? L * 64 * 2 : D * 2;
      char* B = smem + buf * VBUF;
#pragma unroll
      for (int g = 0; g < VROWS / 8; ++g) {
        const int kq = lane_row + swk[g & 1];
        constexpr int AUX = (CLIPMI_ATTN_ABLATE & 512) ? 2 : 0;   // nt
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CLIPMI_LDS_PTR(B + g * 1024), 16, radd(kq + koff, g * gstep), 0, 0, AUX);                        // K
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CLIPMI_LDS_PTR(B + VARR + g * 1024), 16, radd(lane_row + swv + 2 * koff, g * gstep), 0, 0, AUX);  // V
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CLIPMI_LDS_PTR(B + 2 * VARR + g * 1024), 16, radd(kq, g * gstep), 0, 0, AUX);                      // Q
      }
    };
    stage(item, 0);
    int buf = 0;
    for (; item < n_items; item += gridDim.x, buf ^= 1) {
#ifdef CLIPMI_TUNING
      const bool stamp = stamps != nullptr && lane == 0;
      long long* sp = stamps + ((size_t)item * 8 + 7) * 8;
      if (stamp) sp[0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this item's operands have landed
#ifdef CLIPMI_TUNING
      if (stamp) sp[1] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
      __builtin_amdgcn_s_barrier();                       // ... and every query wave is done with the other buffer
#ifdef CLIPMI_TUNING
      if (stamp) sp[2] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
      const int next = item + gridDim.x;
      if (next < n_items) stage(next, buf ^ 1);
#ifdef CLIPMI_TUNING
      if (stamp) sp[3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    }
    return;
  }

  // ---- query waves: no VMEM load in their instruction stream
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int q0 = wave * 32;
  const int kswz = (r32 >> 1) & 7;
  int kro[4], vro[2], qro[4];   // lane-constant byte offsets inside a buffer
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kro[ks] = r32 * 128 + (((2 * ks + hh) ^ kswz) << 4);
    qro[ks] = 2 * VARR + (q0 + r32) * 128 + (((2 * ks + hh) ^ kswz) << 4);   // (q0 >> 1) & 7 == 0: the swizzle of row q is kswz
  }
  {
    const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int fq = (qq >> 1) & 1;
    const int lane_base = VARR + hh * 512 + qq * 128 + ((((lane >> 4) & 1) * 2 + (pp >> 1)) << 4) + (pp & 1) * 8;
    vro[0] = lane_base + fq * 64;
    vro[1] = lane_base + (1 - fq) * 64;
  }
  int buf = 0;
  for (; item < n_items; item += gridDim.x, buf ^= 1) {
#ifdef CLIPMI_TUNING
    const bool stamp = stamps != nullptr && lane == 0;
    long long* sp = stamps + ((size_t)item * 8 + wave) * 8;
    if (stamp) sp[0] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_barrier();   // the loaders' vmcnt(0) came first: this item's K / V / Q are in LDS
#ifdef CLIPMI_TUNING
    if (stamp) sp[1] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    f32x16 oacc[2];
    f32x16 lacc;
    {   // fragment reads pinned ahead of their MFMAs (attend_dense_pf)
      const uint32_t lb = lds_base + (uint32_t)(buf * VBUF);
      const uint32_t ka[4] = {lb + (uint32_t)kro[0], lb + (uint32_t)kro[1], lb + (uint32_t)kro[2], lb + (uint32_t)kro[3]};
      const uint32_t va[2] = {lb + (uint32_t)vro[0], lb + (uint32_t)vro[1]};
      const uint32_t qa[4] = {lb + (uint32_t)qro[0], lb + (uint32_t)qro[1], lb + (uint32_t)qro[2], lb + (uint32_t)qro[3]};
#ifdef CLIPMI_TUNING
      if ((CLIPMI_ATTN_ABLATE & 8) && wave >= 4) continue;
      attend_dense_pf<NKT, GROUP>(ka, va, qa, L, hh, oacc, lacc, stamp ? sp : nullptr);
#else
      attend_dense_pf<NKT, GROUP>(ka, va, qa, L, hh, oacc, lacc);
#endif
    }
#ifdef CLIPMI_TUNING
    asm volatile("" :: "v"(oacc[0][0]), "v"(oacc[1][15]), "v"(lacc[0]));
    if (stamp) sp[2] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (CLIPMI_ATTN_ABLATE & 256) {
      asm volatile("" :: "v"(oacc[0]), "v"(oacc[1]), "v"(lacc[0]));
    } else {
      // the wave's own Q rows of this item are dead (qf was read before the first MFMA): its output tile goes through them as full lines.
      // Rows >= VROWS of the last tile would lie in the other buffer's K rows (being written by the loader): never parked (nrows <= 5 there).
      const int n = item / H, h = item - n * H;
      const int nrows = L - q0 < 32 ? L - q0 : 32;
      store_out_lines<NT>(out + ((int64_t)n * L + q0) * D + h * 64, D, lds_base + (uint32_t)(buf * VBUF + 2 * VARR + q0 * 128), oacc, lacc[0], lane, nrows);
    }
#ifdef CLIPMI_TUNING
    if (stamp) sp[3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
  }
}

__global__ __launch_bounds__(512, 2) void attention_vision_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int L, int H, int n_items CLIPMI_VISION_STAMPS_PARAM) {
  attention_vision_body<false>(qkv, out, L, H, n_items CLIPMI_VISION_STAMPS_ARG);
}
__global__ __launch_bounds__(512, 2) void attention_vision_nt_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int L, int H, int n_items CLIPMI_VISION_STAMPS_PARAM) {
  attention_vision_body<true>(qkv, out, L, H, n_items CLIPMI_VISION_STAMPS_ARG);
}

// Measured on MI355X, B = 256 (tools/block_ab2.py, profiles/r02_attention_ab.txt): persistent kernel without a loader wave
// 81-83 us, this kernel 74-78 us; two loader waves, compiler-placed fragment reads and the segmented forms were A/B arms of
// round 2 (same bits, not faster) and are gone.  Round 3 (profiles/r03_attention_ablation.txt, r03_attention_store_policy.txt): the
// launch is bounded by its 310 MB, not by either pipe; with full-line non-temporal output stores 64-66 us (NT = true, the default).
template <bool NT>
int launch_vision(const half_t* qkv, half_t* out, int N, int L, int H, hipStream_t s) {
  static DeviceOnce attr_once;
  auto fn = NT ? attention_vision_nt_kernel : attention_vision_kernel;
  ensure_dynamic_lds(fn, VSMEM, attr_once);
  const int n_cu = device_cus();
  const int n_items = N * H;
  const int grid = n_items < n_cu ? n_items : n_cu;
#ifdef CLIPMI_TUNING
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), VSMEM, s, qkv, out, L, H, n_items, g_tuning_stamps.load(std::memory_order_relaxed));
#else
  hipLaunchKernelGGL(fn, dim3(grid), dim3(512), VSMEM, s, qkv, out, L, H, n_items);   // 7 query waves + the loader
#endif
  return check_launch("attention_vision_kernel");
}

// (Round 4 built the block's in-projection INTO this kernel -- a persistent workgroup per (image, head) computing q | k | v into this LDS image
// itself; bit-identical to the two launches, 14-24 % slower than them: the 208 x 192 item GEMM is too small to amortise its barriers and pulls
// 1.28 x the L2 -> LDS bytes per flop of the 256 x 256 tile.  Record with stamps and ablations: profiles/r04_qkv_attention_fusion.txt.)

// ---------------------------------------------------------------------------------------------------------------
// Streaming variant for sequences longer than one key block (ViT-L/14: 257 tokens, ViT-L/14@336: 577): the key
// blocks (NKT*32 keys) of one (sequence, head) pass through a TWO-slot LDS ring -- block kb+1 is DMA'd while block kb
// is computed, so the staging latency that the single-buffer kernel above pays once per block (load -> wait ->
// barrier -> compute, only hidden by the co-resident workgroup) is off the critical path.  With NKT = 4 the ring is
// 64 KiB, so two workgroups still share a CU and interleave their MFMA and softmax phases.
// ---------------------------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(512, 2) void attention_stream_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                                  int L, int H, int causal, int nkb) {
  constexpr int KEYS = NKT * 32;
  constexpr int OPB = KEYS * 128;      // one operand image
  constexpr int BUF = 2 * OPB;         // K + V
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = nthr >> 6;
  const int r32 = lane & 31, hh = lane >> 5;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  const int n = blockIdx.x / H, h = blockIdx.x - n * H;
  const half_t* base = qkv + (int64_t)n * L * ld + h * 64;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, ((int64_t)L * ld - h * 64) * 2);

  const int q0 = (blockIdx.y * nwaves + wave) * 32;
  const bool active = q0 < L;  // wave-uniform
  const int q = q0 + r32;
  const int qc = q < L ? q : L - 1;
  f16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(base + (int64_t)qc * ld + ks * 16 + hh * 8);

  const int kswz = (r32 >> 1) & 7;
  int kro[4], vro[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kro[ks] = r32 * 128 + (((2 * ks + hh) ^ kswz) << 4);
  {
    const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int fq = (qq >> 1) & 1;
    const int lane_base = OPB + hh * 512 + qq * 128 + ((((lane >> 4) & 1) * 2 + (pp >> 1)) << 4) + (pp & 1) * 8;
    vro[0] = lane_base + fq * 64;
    vro[1] = lane_base + (1 - fq) * 64;
  }
  auto stage = [&](int kb, int buf) {
    char* Ks = smem + buf * BUF;
    for (int it = wave; it < KEYS / 8; it += nwaves) {   // one wave-instruction = 8 rows x 128 B
      const int pw = it * 64;
      const int p = pw + lane;
      const int row = p >> 3, cs = p & 7;
      const int koff = ((kb * KEYS + row) * (int)ld + D) * 2;   // rows >= L are outside the descriptor: zero
      CLIPMI_BUFFER_LOAD_LDS16(rs, Ks + pw * 16, koff + ((cs ^ ((row >> 1) & 7)) << 4), 0);
      CLIPMI_BUFFER_LOAD_LDS16(rs, Ks + OPB + pw * 16, koff + D * 2 + ((cs ^ (((row >> 1) & 1) << 2)) << 4), 0);
    }
  };

  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
  float m_run = NEG_BIG;
  f32x16 lacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) lacc[e] = 0.f;

  stage(0, 0);
  for (int kb = 0; kb < nkb; ++kb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // block kb visible to all waves; everybody is done with the other slot (block kb-1)
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));   // complete since the wait above (see the persistent kernel)
    if (kb + 1 < nkb) stage(kb + 1, (kb + 1) & 1);
    if (active) {
      const char* b = smem + (kb & 1) * BUF;
      const char* const kread[4] = {b + kro[0], b + kro[1], b + kro[2], b + kro[3]};
      const char* const vread[2] = {b + vro[0], b + vro[1]};
      const int kb0 = kb * KEYS;
      if (!causal && kb0 + KEYS <= L)   // block-uniform: full block -> straight-line code
        attend_block<NKT, NKT, 3>(kread, vread, qf, kb0, L, causal, q0, q, hh, m_run, oacc, lacc);
      else
        attend_block<NKT, NKT, 0>(kread, vread, qf, kb0, L, causal, q0, q, hh, m_run, oacc, lacc);
    }
  }
  if (active && q < L) store_out(out + ((int64_t)n * L + q) * D + h * 64, oacc, lacc[0], hh);
}

template <int NKT>
int launch_stream(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s) {
  constexpr int SMEM = 2 * 2 * NKT * 32 * 128;
  static DeviceOnce attr_once;
  auto fn = attention_stream_kernel<NKT>;
  ensure_dynamic_lds(fn, SMEM, attr_once);
  const int nqt = (L + 31) / 32;
  const int qsplit = (nqt + 7) / 8;
  const int nw = (nqt + qsplit - 1) / qsplit < 4 ? 4 : (nqt + qsplit - 1) / qsplit;   // query tiles spread evenly over the splits
  const int nkb = (L + NKT * 32 - 1) / (NKT * 32);
  hipLaunchKernelGGL(fn, dim3(N * H, qsplit), dim3(nw * 64), SMEM, s, qkv, out, L, H, causal, nkb);
  return check_launch("attention_stream_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Ring kernel (round 5) for non-causal sequences longer than one key block: ViT-L/14 (257 tokens) and ViT-L/14@336 (577 tokens; BASELINE
// configs[4], clip/model.py:181-183 at the geometry of clip/clip.py:38).  The streaming kernel above spends ~900 cycles per (32 queries x
// 32 keys) tile where the matrix pipe needs 256-320: compiler-placed fragment reads (an LDS round trip in front of every MFMA), one
// workgroup per CU, K | V re-staged per query split behind a load -> wait -> barrier -> compute chain.  Here:
//
//   * persistent workgroups (one per CU: RNW = 11 compute waves + a loader wave = three waves on every SIMD, <= 168 registers) walk the
//     (sequence, head) items; a compute wave works on the 32-query tile it holds in registers (the round-3 vision form: S^T = K Q^T, a lane owns a
//     query column, softmax in registers, P is directly the B operand of O^T += V^T P^T), with every LDS fragment read pinned ahead of its MFMA;
//   * the keys of an item pass by in BLOCKS of 128 (K | V: 32 KiB) through a THREE-slot LDS ring filled by LDS-DMA from the loader wave (a
//     piece issued from a wave in the middle of MFMA / exponent work cost ~200 cycles, from a dedicated wave ~20): at step g it issues block
//     g + 2, so a block has two whole steps to land; one workgroup barrier per block;
//   * an item's query tiles are worked in PASSES of RNW (577 tokens: 19 tiles = 11 + 8, 257 tokens: 9 tiles = one pass); the 32 rows x 128 B of a
//     wave's Q tile come by DMA into its own 4 KiB of LDS a whole pass ahead.  A last pass whose tiles can ALL be shared (R tiles x 2 or 4 waves
//     <= RNW) is SPLIT over the waves by key tile within every block, and the partial (max, sum, O) of a tile are merged through the ring slot
//     that falls free at the pass end (the plan is built on the host: RingPlan); a pass lasts as long as its largest share, so nothing else is split;
//   * outputs straight from registers (store_out).
//
// LDS: 3 x 32 KiB ring + 44 KiB of Q tiles = 140 KiB.  vmcnt discipline (loader): a block is 32 pieces; `s_waitcnt vmcnt(32)` at step g leaves
// at most the 32 youngest pieces in flight -- block g + 1's -- so block g has landed; the Q pieces of a step are issued BEFORE its block pieces
// and are never among the 32 youngest when a pass start needs them (needs >= 2 blocks per item: L > 128).
// History and measurements: profiles/r05_vitl_attention.txt, r05_ring_waves.txt (8 / 10 / 11 waves; LDS counters instead of the barrier and a
// pre-scaled-Q form measured and removed), r05_ring_ablate.txt (what the launch is made of).
// ---------------------------------------------------------------------------------------------------------------
constexpr int RTPB = 4;                     // key tiles per block
constexpr int RKEYS = RTPB * 32;            // keys per block
constexpr int RIMG = RKEYS * 128;           // one operand image of a block
constexpr int RSLOT = 2 * RIMG;             // K | V
constexpr int RNSLOT = 3;
#ifndef CLIPMI_RING_WAVES
#define CLIPMI_RING_WAVES 11
#endif
constexpr int RNW = CLIPMI_RING_WAVES;      // compute waves = query tiles per pass; the loader is wave RNW.  11 + 1 = 12 waves: three on every SIMD, 168 VGPRs
constexpr int RTHREADS = (RNW + 1) * 64;
constexpr int RQ = RNW * 32 * 128;          // the Q tiles of a pass
constexpr int RSMEM = RNSLOT * RSLOT + RQ;  // 96 KiB + 44 KiB
static_assert(RNW >= 4 && RNW <= 12 && RSMEM <= 160 * 1024, "ring attention: 4..12 compute waves, LDS <= 160 KiB");
constexpr int RSPOT = 8192 + 512;           // a partial: O (32 registers x 64 lanes x 4 B) + m + l per lane
constexpr int RMAXPASS = 8;                 // 8 passes x RNW tiles x 32 rows = 2816 tokens with RNW = 11 (the dispatcher's limit below)

// 32-bit words only: a wave reads its entry by a wave-uniform index, which must stay a scalar load (a byte array in the kernel arguments is
// read with global_load_sbyte + s_waitcnt vmcnt(0), and that wait would drain the DMA ring at every step).
struct RingPlan {
  int n_pass, n_blocks, n_rounds;   // passes per item; key blocks per item; merge rounds of the split (last) pass (0: not split)
  // LAST pass, wave w: bits 0-7 query tile + 1 (0: idle) | 8-9 first key tile of every block | 10-12 key tiles per block | 13-16 the wave that
  // owns the tile's result (== w: this wave stores it) | 17-18 merge round (partner waves) | 19-20 scratch spot.  Earlier passes: tile RNW p + w, whole blocks.
  unsigned info[RNW];
};
__host__ __device__ inline int ring_qt(unsigned i) { return (int)(i & 255u) - 1; }
__host__ __device__ inline int ring_first(unsigned i) { return (int)((i >> 8) & 3u); }
__host__ __device__ inline int ring_count(unsigned i) { return (int)((i >> 10) & 7u); }
__host__ __device__ inline int ring_leader(unsigned i) { return (int)((i >> 13) & 15u); }
__host__ __device__ inline int ring_round(unsigned i) { return (int)((i >> 17) & 3u); }
__host__ __device__ inline int ring_spot(unsigned i) { return (int)((i >> 19) & 3u); }

// One block's share of a wave: CNT key tiles (LDS images at ka / va, first key = k_lo) against the wave's 32 queries.  One softmax
// group per call; fragment reads pinned as in attend_dense_pf; only the item's very last key tile can hold keys >= L (wave-uniform test).
#ifdef CLIPMI_TUNING
#define CLIPMI_RING_STAMP(slot, dep) do { asm volatile("" :: "v"(dep)); if (sp) sp[slot] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CLIPMI_RING_STAMP(slot, dep) do { } while (0)
#endif
// One softmax group of CNT (1 or 2) key tiles (LDS images at ka / va, first key = k_lo) against the wave's 32 queries; a block of four tiles is
// two groups, so that a wave alternates a matrix-bound S phase with a vector-bound P.V phase twice per block and the two waves of a SIMD, half a
// phase apart (see the kernel), overlap one's MFMAs with the other's exponentials.  Fragment reads pinned as in attend_dense_pf; only the item's
// very last key tile can hold keys >= L (wave-uniform test).  `ones`: the all-ones A operand of the row-sum MFMA, defined ONCE per wave by inline
// asm (the compiler cannot re-materialise it in front of its use, so it needs no fence); one fence per 16-key step covers P and both V tiles.
// The row sums stay in `lacc` (ones-tile MFMA accumulator) for the WHOLE pass and are read by the vector pipe once, at the pass end, behind
// CLIPMI_MFMA_TO_VALU_FENCE3 (common.h): a per-group `l += lacc[0]` right behind the group's last MFMAs -- hipcc put `s_nop 10` between them -- is
// the second co-residency hazard of this code base (52-160 of 200 LayerNorm launches wrong beside the kernel, 0 with more wait states).
// CLIPMI_RING_ABLATE (build-time, diagnostic builds only: results are wrong with any bit set; make ring_ablate, tools/lib_ab.py):
// 1 no exponent work (P = the raw score bits) | 2 no MFMAs (S and P.V) | 4 no LDS fragment reads | 8 no maximum | 16 no fma in front of v_exp |
// 32 no per-block barrier (loader and compute waves run free: races, timing only) | 64 no LDS-DMA at all (compute on whatever the LDS holds)
#ifndef CLIPMI_RING_ABLATE
#define CLIPMI_RING_ABLATE 0
#endif
template <int CNT>
__device__ __forceinline__ void ring_attend(const uint32_t (&ka)[4], const uint32_t (&va)[2], const f16x8 (&qf)[4], const f16x8& ones, int k_lo, int L,
                                            int hh, float& m_run, f32x16 (&oacc)[2], f32x16& lacc, long long* sp = nullptr) {
  constexpr float C = 0.125f * LOG2E;
  const f32x16 zero16 = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f16x8 kf[2][4];
  f16x4 vf[2][4];
  f32x16 s[CNT];
  static_assert(CNT == 1 || CNT == 2, "one or two key tiles per softmax group");
  constexpr bool NO_LDS = (CLIPMI_RING_ABLATE & 4) != 0, NO_MFMA = (CLIPMI_RING_ABLATE & 2) != 0;
  if constexpr (NO_LDS) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kf[i][j] = qf[j];
        vf[i][j] = f16x4{qf[j][0], qf[j][1], qf[j][2], qf[j][3]};
        asm volatile("" : "+v"(kf[i][j]), "+v"(vf[i][j]));
      }
  } else {
    read_k<0>(kf[0], ka);
  }
  auto s_tile = [&](auto t_tag) {
    constexpr int T = decltype(t_tag)::value;
    constexpr int CUR = T & 1;
    if constexpr (NO_LDS) {
    } else if constexpr (T + 1 < CNT) {
      read_k<T + 1>(kf[CUR ^ 1], ka);
      lds_wait4<4>(kf[CUR][0], kf[CUR][1], kf[CUR][2], kf[CUR][3]);
    } else {
      lds_wait4<0>(kf[CUR][0], kf[CUR][1], kf[CUR][2], kf[CUR][3]);
    }
    if constexpr (NO_MFMA) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[T][e] = (float)kf[CUR][e & 3][e >> 2] + (float)qf[e & 3][e >> 2];
    } else {
      s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[CUR][0], qf[0], zero16, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) s[T] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[CUR][ks], qf[ks], s[T], 0, 0, 0);
    }
    if (k_lo + T * 32 + 32 > L) {   // wave-uniform: the item's last key tile
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = k_lo + T * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        s[T][e] = key >= L ? NEG_BIG : s[T][e];
      }
    }
  };
  s_tile(std::integral_constant<int, 0>{});
  if constexpr (CNT > 1) s_tile(std::integral_constant<int, 1>{});
  CLIPMI_RING_STAMP(3, s[CNT - 1][0]);
  if constexpr (!NO_LDS) read_v<0>(vf[0], va);   // the P.V phase opens with these: behind the maximum and the rescale by the time they are needed
  float mloc = NEG_BIG;
  if constexpr (CLIPMI_RING_ABLATE & 8) {
    mloc = s[0][0];
  } else {
#pragma unroll
    for (int t = 0; t < CNT; ++t)
#pragma unroll
      for (int e = 0; e < 16; e += 2) mloc = fmaxf(fmaxf(s[t][e], s[t][e + 1]), mloc);
    mloc = fmaxf(mloc, swap32_f(mloc));
  }
  // The reference point of the exponentials moves only when some query's maximum has grown by more than RING_SLACK (2^8 in P): with 32 queries per wave
  // "some lane saw a new maximum" holds in nearly every group (SQ_INSTS_VALU: 96 vector instructions per tile where the straight path has 56 -- the 48
  // accumulator multiplies of this branch), while a maximum that jumps by e^5.5 after the first group is rare.  In between P = 2^((s - m_run) C) <= 256:
  // exact in fp16 to the same 2^-11, sums and O in fp32; O / l does not depend on the reference point.  A lane's m_run is its reference, not its maximum.
  constexpr float RING_SLACK = 8.0f / C;
  const float m_new = fmaxf(m_run, mloc);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(m_new > m_run + RING_SLACK) != 0ull, 0)) {   // wave-uniform; every lane then moves to its own maximum
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
#pragma unroll
    for (int e = 0; e < 16; ++e) lacc[e] *= alpha;   // the WHOLE tuple (an element write into an MFMA tuple makes hipcc copy it at every join)
    CLIPMI_VALU_TO_MFMA_FENCE3(oacc[0], oacc[1], lacc);   // VALU-written accumulators are MFMA sources (SrcC) below
    m_run = m_new;
  }
  const float mc = m_run * C;
  CLIPMI_RING_STAMP(4, mc);
  auto pv_step = [&](auto t_tag, auto ss_tag) {
    constexpr int T = decltype(t_tag)::value, SS = decltype(ss_tag)::value;
    constexpr int STEP = T * 2 + SS, CUR = STEP & 1;
    constexpr bool LAST = STEP == 2 * CNT - 1;
    if constexpr (!LAST && !NO_LDS) read_v<((STEP + 1) / 2) * 4096 + ((STEP + 1) & 1) * 2048>(vf[CUR ^ 1], va);
    f16x8 pf;
    if constexpr (CLIPMI_RING_ABLATE & 1) {
      const f32x4 raw = f32x4{s[T][8 * SS], s[T][8 * SS + 1] + mc, s[T][8 * SS + 2], s[T][8 * SS + 3]};
      pf = __builtin_bit_cast(f16x8, raw);
    } else if constexpr (CLIPMI_RING_ABLATE & 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (half_t)__builtin_amdgcn_exp2f(s[T][8 * SS + j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (half_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[T][8 * SS + j], C, -mc));
    }
    if constexpr (NO_LDS) {
    } else if constexpr (!LAST) lds_wait4h<4>(vf[CUR][0], vf[CUR][1], vf[CUR][2], vf[CUR][3]);
    else lds_wait4h<0>(vf[CUR][0], vf[CUR][1], vf[CUR][2], vf[CUR][3]);
    f16x8 v8[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const f16x4 lo = vf[CUR][dt * 2], hi = vf[CUR][dt * 2 + 1];
      v8[dt] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    CLIPMI_VALU_TO_MFMA_FENCE3(pf, v8[0], v8[1]);   // ONE fence: P comes from conversions, the V halves may have been moved together by VALU copies
    if constexpr (NO_MFMA) {
      oacc[0][STEP] += (float)pf[0] * (float)v8[0][0] + (float)pf[2] * (float)v8[0][5];
      oacc[1][STEP] += (float)pf[4] * (float)v8[1][0] + (float)pf[6] * (float)v8[1][5];
      lacc[STEP] += (float)pf[1] + (float)pf[3] + (float)pf[5] + (float)pf[7];
    } else {
      oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v8[0], pf, oacc[0], 0, 0, 0);
      oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v8[1], pf, oacc[1], 0, 0, 0);
      lacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, lacc, 0, 0, 0);
    }
  };
  pv_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  pv_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  if constexpr (CNT > 1) {
    pv_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
    pv_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  }
}

// The lane's LDS byte offsets of its four K (or Q) fragments and its two transposed-V read bases inside a (K | V) block image.
__device__ __forceinline__ void frag_offsets(int lane, int (&kro)[4], int (&vro)[2]) {
  const int r32 = lane & 31, hh = lane >> 5;
  const int kswz = (r32 >> 1) & 7;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kro[ks] = r32 * 128 + (((2 * ks + hh) ^ kswz) << 4);
  const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
  const int fq = (qq >> 1) & 1;
  const int lane_base = RIMG + hh * 512 + qq * 128 + ((((lane >> 4) & 1) * 2 + (pp >> 1)) << 4) + (pp & 1) * 8;
  vro[0] = lane_base + fq * 64;
  vro[1] = lane_base + (1 - fq) * 64;
}

// diagnostic build: stamps[((workgroup * 64 + step) * (RNW + 1) + wave) * 8 + k], s_memtime (shader cycles) of lane 0, workgroups 0..7, steps 0..63.
// compute waves 0 .. RNW-1: 0 loop top | 2 past the block's barrier | 3 S tiles of the LAST group issued | 4 its maximum + rescale done | 5 block computed | 6 step end
// (pass end: merged + stored);  loader (wave RNW): 0 loop top | 1 block g landed | 2 past the barrier | 7 DMA of block g + 2 (and the next Q tiles) issued
__global__ __launch_bounds__(RTHREADS) void attention_ring_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int L, int H, int n_items,
                                                             const RingPlan plan CLIPMI_VISION_STAMPS_PARAM) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..RNW-1 compute, RNW loader
  const int r32 = lane & 31, hh = lane >> 5;
  const int D = H * 64;
  const int ld = 3 * D;                                   // halves per qkv row (< 2^31 / L: checked by the launcher)
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int P = plan.n_pass, NB = plan.n_blocks;
  const int gstride = gridDim.x;
  const int my_items = (int)blockIdx.x < n_items ? (n_items - 1 - (int)blockIdx.x) / gstride + 1 : 0;
  const int n_steps = my_items * P * NB;
  if (n_steps == 0) return;

  // ---- cursors over the stream of (item, pass, block) steps, each with its (sequence, head) kept incrementally (no division per step)
  struct Cur { int item, n, h, p, b; };
  const int dn = gstride / H, dh = gstride - dn * H;
  auto next_item = [&](Cur& u) {
    u.item += gstride;
    u.n += dn;
    u.h += dh;
    if (u.h >= H) { u.h -= H; ++u.n; }
  };
  auto advance = [&](Cur& u) {
    if (++u.b == NB) { u.b = 0; if (++u.p == P) { u.p = 0; next_item(u); } }
  };
  Cur c{(int)blockIdx.x, (int)blockIdx.x / H, (int)blockIdx.x % H, 0, 0};

  if (wave == RNW) {
    // ================= loader wave: every LDS-DMA piece of the workgroup (a dedicated wave issues a piece in tens of cycles; a wave in the middle of
    // MFMA / exponent work was measured at ~200 per piece, a fifth of the step: profiles/r05_vitl_attention.txt) =================
    // A piece = 8 rows x 128 B; lane -> row lr of the piece, 16-byte chunk cs.  Lane-constant byte offsets by piece parity (the swizzle of a row
    // depends on (row >> 1) & 7 = ((lr >> 1) + 4 (piece & 1)) & 7); a piece adds one scalar (its rows' offset) per operand.
    const int lr = lane >> 3, cs = lane & 7;
    const int lane_row = lr * ld * 2;
    const int swv = (cs ^ (((lr >> 1) & 1) << 2)) << 4;                                      // V: chunk ^ (((row >> 1) & 1) << 2)
    const int swk[2] = {(cs ^ (lr >> 1)) << 4, (cs ^ (4 + (lr >> 1))) << 4};                  // K, Q: chunk ^ ((row >> 1) & 7), by piece parity
    const int piece_bytes = 8 * ld * 2;
    auto radd = [](int base, int add) {
      int r;
      asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(base), "s"(add));
      return r;
    };
    auto item_rsrc = [&](const Cur& u) {
      return make_rsrc(qkv + (int64_t)u.n * L * ld + u.h * 64, ((int64_t)L * ld - u.h * 64) * 2);   // rows >= L: outside the descriptor, read as zero
    };
    auto dma_block = [&](const Cur& u, int slot) {   // 32 pieces: K | V of 128 keys
      if constexpr (CLIPMI_RING_ABLATE & 64) return;
      const __amdgpu_buffer_rsrc_t rs = item_rsrc(u);
      char* B = smem + slot * RSLOT;
      const int boff = u.b * RKEYS * ld * 2;
#pragma unroll
      for (int pc = 0; pc < RKEYS / 8; ++pc) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CLIPMI_LDS_PTR(B + pc * 1024), 16, radd(lane_row + swk[pc & 1] + D * 2, boff + pc * piece_bytes), 0, 0, 0);          // K
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CLIPMI_LDS_PTR(B + RIMG + pc * 1024), 16, radd(lane_row + swv + 2 * D * 2, boff + pc * piece_bytes), 0, 0, 0);     // V
      }
    };
    auto dma_q = [&](const Cur& u) {   // the Q tiles of pass u.p, tile of wave w into region w: 4 pieces per wave
      if constexpr (CLIPMI_RING_ABLATE & 64) return;
      const __amdgpu_buffer_rsrc_t rs = item_rsrc(u);
      char* Q = smem + RNSLOT * RSLOT;
      for (int w = 0; w < RNW; ++w) {
        int qt = u.p == P - 1 ? ring_qt(plan.info[w]) : u.p * RNW + w;
        qt = qt < 0 ? 0 : qt;
        const int toff = qt * 32 * ld * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, CLIPMI_LDS_PTR(Q + w * 4096 + j * 1024), 16, radd(lane_row + swk[j & 1], toff + j * piece_bytes), 0, 0, 0);
      }
    };
    Cur f = c, q = c;
    auto next_pass = [&](Cur& u) {
      if (++u.p == P) { u.p = 0; next_item(u); }
    };
    dma_q(q);
    next_pass(q);
    dma_block(f, 0);
    advance(f);
    if (n_steps > 1) {
      dma_block(f, 1);
      advance(f);
    }
    for (int g = 0; g < n_steps; ++g) {
#ifdef CLIPMI_TUNING
      long long* sp = (stamps != nullptr && lane == 0 && blockIdx.x < 8 && g < 64) ? stamps + (((size_t)blockIdx.x * 64 + g) * (RNW + 1) + wave) * 8 : nullptr;
      if (sp) sp[0] = (long long)__builtin_amdgcn_s_memtime();
#endif
      // Block g (and every Q tile a pass start needs) has landed once at most the 32 youngest pieces are in flight: those are block g + 1's, issued
      // a step after block g's; the Q pieces of a step are issued BEFORE its block pieces, so they are never among the 32 youngest when needed.
      if (g + 1 < n_steps) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CLIPMI_TUNING
      if (sp) sp[1] = (long long)__builtin_amdgcn_s_memtime();
#endif
      if constexpr (!(CLIPMI_RING_ABLATE & 32)) __builtin_amdgcn_s_barrier();   // block g is in LDS for every wave; every wave is done with block g - 1, whose slot block g + 2 takes
#ifdef CLIPMI_TUNING
      if (sp) sp[2] = (long long)__builtin_amdgcn_s_memtime();
#endif
      // the NEXT pass's Q tiles go out one step AFTER the pass start: by this barrier every compute wave has taken its fragments of the current ones
      if (c.b == 1 && q.item < n_items) {
        dma_q(q);
        next_pass(q);
      }
      if (g + 2 < n_steps) {
        dma_block(f, (g + 2) % RNSLOT);
        advance(f);
      }
#ifdef CLIPMI_TUNING
      if (sp) sp[7] = (long long)__builtin_amdgcn_s_memtime();
#endif
      if (c.b == NB - 1 && c.p == P - 1 && plan.n_rounds > 0) {   // the compute waves' merge barriers of a split pass end (below): 1 + 2 per round - 1
        const int nbar = 2 * plan.n_rounds;
        for (int i = 0; i < nbar; ++i) __builtin_amdgcn_s_barrier();
      }
      advance(c);
    }
    return;
  }

  // ================= compute waves: no vector-memory LOAD in their instruction stream =================
  const unsigned my_info = plan.info[wave];             // (scalar load, once: see RingPlan)
  const int last_qt = ring_qt(my_info);
  int kro[4], vro[2];
  frag_offsets(lane, kro, vro);
  f16x8 qf[4], ones;
  {   // 1.0 in every half: defined by inline asm, never written again
    unsigned o0, o1, o2, o3;
    asm volatile("v_mov_b32 %0, 0x3c003c00\n\tv_mov_b32 %1, 0x3c003c00\n\tv_mov_b32 %2, 0x3c003c00\n\tv_mov_b32 %3, 0x3c003c00\n\ts_nop 3"
                 : "=v"(o0), "=v"(o1), "=v"(o2), "=v"(o3));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    ones = __builtin_bit_cast(f16x8, u32x4{o0, o1, o2, o3});
  }
  f32x16 oacc[2], lacc;
  float m_run = NEG_BIG;
  const uint32_t q_lds = lds_base + (uint32_t)(RNSLOT * RSLOT + wave * 4096);

  for (int g = 0; g < n_steps; ++g) {
    const int slot = g % RNSLOT;
#ifdef CLIPMI_TUNING
    long long* sp = (stamps != nullptr && lane == 0 && blockIdx.x < 8 && g < 64) ? stamps + (((size_t)blockIdx.x * 64 + g) * (RNW + 1) + wave) * 8 : nullptr;
    if (sp) sp[0] = (long long)__builtin_amdgcn_s_memtime();
#endif
    if constexpr (!(CLIPMI_RING_ABLATE & 32)) __builtin_amdgcn_s_barrier();   // the loader's vmcnt wait came first: block g (and at a pass start this wave's Q tile) is in LDS
#ifdef CLIPMI_TUNING
    if (sp) sp[2] = (long long)__builtin_amdgcn_s_memtime();
    if (sp) sp[1] = (long long)__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));   // HW_ID.SIMD_ID of this wave (compute waves have no stamp 1)
#endif
    const bool split = c.p == P - 1 && plan.n_rounds > 0;
    if (c.b == 0) {
      auto qat = [&](int off) { uint32_t r; asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(off), "s"(q_lds)); return r; };
      const uint32_t qa[4] = {qat(kro[0]), qat(kro[1]), qat(kro[2]), qat(kro[3])};
      read_k<0>(qf, qa);
      lds_wait4<0>(qf[0], qf[1], qf[2], qf[3]);
      m_run = NEG_BIG;
#pragma unroll
      for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; lacc[e] = 0.f; }
      CLIPMI_VALU_TO_MFMA_FENCE3(oacc[0], oacc[1], lacc);
    }
    // ---- this wave's share of block g
    const int qt = c.p == P - 1 ? last_qt : c.p * RNW + wave;
    {
      int first = split ? ring_first(my_info) : 0, count = split ? ring_count(my_info) : RTPB;
      const int live = (L - c.b * RKEYS + 31) >> 5;          // live key tiles of this block (>= 1)
      if (first + count > live) count = live - first;
      if (qt >= 0) {
#ifdef CLIPMI_TUNING
        long long* spa = sp;
#else
        long long* spa = nullptr;
#endif
        // two loops with ONE body each: the accumulators then stay in their registers (a loop whose body chooses between two instantiations, or
        // a conditional rescale written on one element of an MFMA tuple, made hipcc copy all 48 accumulator registers at every join)
        for (; count >= 2; count -= 2, first += 2) {           // softmax groups of two key tiles ...
          const uint32_t sb = lds_base + (uint32_t)(slot * RSLOT + first * 4096);
          const uint32_t ka[4] = {sb + (uint32_t)kro[0], sb + (uint32_t)kro[1], sb + (uint32_t)kro[2], sb + (uint32_t)kro[3]};
          const uint32_t va[2] = {sb + (uint32_t)vro[0], sb + (uint32_t)vro[1]};
          ring_attend<2>(ka, va, qf, ones, c.b * RKEYS + first * 32, L, hh, m_run, oacc, lacc, spa);
        }
        for (; count >= 1; --count, ++first) {                   // ... and a last one of a single tile
          const uint32_t sb = lds_base + (uint32_t)(slot * RSLOT + first * 4096);
          const uint32_t ka[4] = {sb + (uint32_t)kro[0], sb + (uint32_t)kro[1], sb + (uint32_t)kro[2], sb + (uint32_t)kro[3]};
          const uint32_t va[2] = {sb + (uint32_t)vro[0], sb + (uint32_t)vro[1]};
          ring_attend<1>(ka, va, qf, ones, c.b * RKEYS + first * 32, L, hh, m_run, oacc, lacc, spa);
        }
        // Whatever follows the block may read the accumulators with the vector pipe -- the pass end does, and hipcc itself copies the tuples where
        // the loop bodies meet --: never right behind the last MFMAs (CLIPMI_MFMA_TO_VALU_FENCE3, common.h).  32 wait states per block: < 1 %.
        CLIPMI_MFMA_TO_VALU_FENCE3(oacc[0], oacc[1], lacc);
      }
    }
#ifdef CLIPMI_TUNING
    asm volatile("" :: "v"(oacc[0][0]), "v"(oacc[1][15]), "v"(lacc[0]));
    if (sp) sp[5] = (long long)__builtin_amdgcn_s_memtime();
#endif
    // ---- pass end: merge the partials of a split tile, store the tile
    if (c.b == NB - 1) {
      int owner = wave;
      float l_run = lacc[0];   // every row of the ones-tile holds the same sum (read behind the fence that ends every group: ring_attend)
      if (split) {
        __builtin_amdgcn_s_barrier();   // every wave is done reading this block: its slot is scratch until the next step's barrier
        const uint32_t scratch = lds_base + (uint32_t)(slot * RSLOT);
        owner = ring_leader(my_info);
        for (int round = 0; round < plan.n_rounds; ++round) {
          if (owner != wave && ring_round(my_info) == round && qt >= 0) {   // partner: park (O, m, l)
            const uint32_t spot = scratch + (uint32_t)(ring_spot(my_info) * RSPOT);
#pragma unroll
            for (int r4 = 0; r4 < 8; ++r4) {
              const f32x4 v = f32x4{oacc[r4 >> 2][(r4 & 3) * 4 + 0], oacc[r4 >> 2][(r4 & 3) * 4 + 1], oacc[r4 >> 2][(r4 & 3) * 4 + 2], oacc[r4 >> 2][(r4 & 3) * 4 + 3]};
              asm volatile("ds_write_b128 %0, %1" :: "v"(spot + (uint32_t)(r4 * 1024 + lane * 16)), "v"(v) : "memory");
            }
            asm volatile("ds_write_b32 %0, %1" :: "v"(spot + (uint32_t)(8192 + lane * 4)), "v"(m_run) : "memory");
            asm volatile("ds_write_b32 %0, %1" :: "v"(spot + (uint32_t)(8192 + 256 + lane * 4)), "v"(l_run) : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (owner == wave && qt >= 0) {
            constexpr float C = 0.125f * LOG2E;
            for (int w2 = 0; w2 < RNW; ++w2) {
              const unsigned i2 = plan.info[w2];
              if (w2 == wave || ring_leader(i2) != wave || ring_round(i2) != round || ring_qt(i2) < 0) continue;   // wave-uniform
              const uint32_t spot = scratch + (uint32_t)(ring_spot(i2) * RSPOT);
              float m2, l2;
              f32x4 o2[8];
              asm volatile("ds_read_b32 %0, %1" : "=v"(m2) : "v"(spot + (uint32_t)(8192 + lane * 4)) : "memory");
              asm volatile("ds_read_b32 %0, %1" : "=v"(l2) : "v"(spot + (uint32_t)(8192 + 256 + lane * 4)) : "memory");
#pragma unroll
              for (int r4 = 0; r4 < 8; ++r4) asm volatile("ds_read_b128 %0, %1" : "=v"(o2[r4]) : "v"(spot + (uint32_t)(r4 * 1024 + lane * 16)) : "memory");
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(m2), "+v"(l2), "+v"(o2[0]), "+v"(o2[1]), "+v"(o2[2]), "+v"(o2[3]), "+v"(o2[4]), "+v"(o2[5]), "+v"(o2[6]), "+v"(o2[7]) :: "memory");
              const float m_new = fmaxf(m_run, m2);
              const float a1 = __builtin_amdgcn_exp2f((m_run - m_new) * C), a2 = __builtin_amdgcn_exp2f((m2 - m_new) * C);
#pragma unroll
              for (int r4 = 0; r4 < 8; ++r4)
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[r4 >> 2][(r4 & 3) * 4 + e] = oacc[r4 >> 2][(r4 & 3) * 4 + e] * a1 + o2[r4][e] * a2;
              l_run = l_run * a1 + l2 * a2;
              m_run = m_new;
            }
          }
          if (round + 1 < plan.n_rounds) __builtin_amdgcn_s_barrier();   // the spots are free again for the next round
        }
      }
      // Straight from the registers: four 16-byte stores per lane (store_out).  The tile is 1 / 19 of the item's traffic here, not a quarter as in
      // the 197-token kernel: full-line staging through LDS would need a free region -- i.e. one more workgroup barrier per pass.
      const bool stores = owner == wave && qt >= 0;          // wave-uniform; every lane of a storing wave takes part in store_out's lane swaps
      if (stores) {
        const int q0 = qt * 32;
        store_out(out + ((int64_t)c.n * L + (q0 + r32 < L ? q0 + r32 : L - 1)) * D + c.h * 64, oacc, l_run, hh, q0 + r32 < L);
      }
    }
#ifdef CLIPMI_TUNING
    if (sp) sp[6] = (long long)__builtin_amdgcn_s_memtime();
#endif
    advance(c);
  }
}

// Host side of the plan: passes of RNW query tiles.  A short last pass is split by key tile only when EVERY tile of it can be (R tiles x `ways`
// waves <= RNW, ways = 2 or 4): a pass lasts as long as its largest share, so splitting some of the tiles buys nothing and costs merge rounds.
static RingPlan make_ring_plan(int L) {
  RingPlan pl;
  memset(&pl, 0, sizeof(pl));
  const int nqt = (L + 31) / 32;
  pl.n_pass = (nqt + RNW - 1) / RNW;
  pl.n_blocks = (L + RKEYS - 1) / RKEYS;
  auto pack = [](int qt, int first, int count, int leader, int round, int spot) {
    return (unsigned)(qt + 1) | (unsigned)first << 8 | (unsigned)count << 10 | (unsigned)leader << 13 | (unsigned)round << 17 | (unsigned)spot << 19;
  };
  const int base = (pl.n_pass - 1) * RNW;
  const int R = nqt - base;                                // tiles of the last pass (1..RNW)
  for (int w = 0; w < RNW; ++w) pl.info[w] = pack(w < R ? base + w : -1, 0, RTPB, w, 0, 0);
  const int ways = 4 * R <= RNW ? 4 : 2 * R <= RNW ? 2 : 1;
  if (ways > 1) {
    const int cnt = RTPB / ways;
    int w = 0, partners = 0;
    for (int i = 0; i < R; ++i) {
      const int lead = w;
      for (int j = 0; j < ways; ++j, ++w) {
        pl.info[w] = pack(base + i, j * cnt, cnt, lead, j > 0 ? partners / 3 : 0, j > 0 ? partners % 3 : 0);
        if (j > 0) ++partners;
      }
    }
    for (; w < RNW; ++w) pl.info[w] = pack(-1, 0, RTPB, w, 0, 0);
    pl.n_rounds = (partners + 2) / 3;
  }
  return pl;
}

int launch_ring(const half_t* qkv, half_t* out, int N, int L, int H, hipStream_t s) {
  static DeviceOnce attr_once;
  ensure_dynamic_lds(attention_ring_kernel, RSMEM, attr_once);
  const RingPlan plan = make_ring_plan(L);
  const int n_cu = device_cus();
  const int n_items = N * H;
  const int grid = n_items < n_cu ? n_items : n_cu;
#ifdef CLIPMI_TUNING
  hipLaunchKernelGGL(attention_ring_kernel, dim3(grid), dim3(RTHREADS), RSMEM, s, qkv, out, L, H, n_items, plan, g_tuning_stamps.load(std::memory_order_relaxed));
#else
  hipLaunchKernelGGL(attention_ring_kernel, dim3(grid), dim3(RTHREADS), RSMEM, s, qkv, out, L, H, n_items, plan);   // RNW compute waves + the loader
#endif
  return check_launch("attention_ring_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Short sequences (L <= 32: the text tower after dead-row elimination -- CoOp prompts end at token ~22 of 77, zero-shot templates at ~10 --, round 5).
// One key tile and one query tile per (sequence, head) item: the persistent kernel above gives such an item a whole workgroup (four waves, one of them
// working), two workgroup barriers and a 96-row staging loop.  Here an item belongs to ONE WAVE: the wave brings its K | V tile (32 rows x 128 B each,
// rows >= L outside the buffer descriptor: zero) into its own 8 KiB of LDS by LDS-DMA, takes its Q fragments from global memory, waits for its own
// vmcnt and computes -- no workgroup barrier anywhere; twelve to sixteen such waves per CU hide each other's load latency.  Same arithmetic as the
// other kernels (attend_block: S^T = K Q^T, masks, in-register softmax, P as the B operand of O^T += V^T P^T, row sums on a ones-tile).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void attention_small_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int L, int H, int causal,
                                                                 int n_items) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 8192];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  char* const mine = smem + wave * 8192;                 // K tile | V tile of this wave's current item
  const int kswz = (r32 >> 1) & 7;
  const char* kread[4];
  const char* vread[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kread[ks] = mine + r32 * 128 + (((2 * ks + hh) ^ kswz) << 4);
  {
    const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int fq = (qq >> 1) & 1;
    const int lane_base = 4096 + hh * 512 + qq * 128 + ((((lane >> 4) & 1) * 2 + (pp >> 1)) << 4) + (pp & 1) * 8;
    vread[0] = mine + lane_base + fq * 64;
    vread[1] = mine + lane_base + (1 - fq) * 64;
  }
  const int lr = lane >> 3, cs = lane & 7;
  const int q = r32, qc = q < L ? q : L - 1;
  const int stride = gridDim.x * 4;
  for (int item = blockIdx.x * 4 + wave; item < n_items; item += stride) {
    const int n = item / H, h = item - n * H;
    const half_t* base = qkv + (int64_t)n * L * ld + h * 64;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, ((int64_t)L * ld - h * 64) * 2);   // rows >= L: outside, read as zero
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
      const int row = pc * 8 + lr;
      const int koff = (row * (int)ld + D) * 2;
      CLIPMI_BUFFER_LOAD_LDS16(rs, mine + pc * 1024, koff + ((cs ^ ((row >> 1) & 7)) << 4), 0);
      CLIPMI_BUFFER_LOAD_LDS16(rs, mine + 4096 + pc * 1024, koff + D * 2 + ((cs ^ (((row >> 1) & 1) << 2)) << 4), 0);
    }
    f16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(base + (int64_t)qc * ld + ks * 16 + hh * 8);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]) :: "memory");   // this wave's tiles and fragments are in
    f32x16 oacc[2], lacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; lacc[e] = 0.f; }
    float m_run = NEG_BIG;
    attend_block<1, 1, 0>(kread, vread, qf, 0, L, causal, 0, q, hh, m_run, oacc, lacc);
    CLIPMI_MFMA_TO_VALU_FENCE3(oacc[0], oacc[1], lacc);   // the vector pipe takes the accumulators over below: not right behind the item's last MFMAs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every fragment read of this item is back before the next item's DMA may land on the tiles
    store_out(out + ((int64_t)n * L + qc) * D + h * 64, oacc, lacc[0], hh, q < L);
  }
}

int launch_small(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s) {
  const int n_items = N * H;
  const int slots = device_cus() * 4;                     // four workgroups of four waves per CU
  const int need = (n_items + 3) / 4;
  hipLaunchKernelGGL(attention_small_kernel, dim3(need < slots ? need : slots), dim3(256), 0, s, qkv, out, L, H, causal, n_items);
  return check_launch("attention_small_kernel");
}

template <int NKT, int GROUP, int DENSE>
int launch_persist(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s) {
  constexpr int SMEM = 2 * 2 * NKT * 32 * 128;
  static DeviceOnce attr_once;
  auto fn = attention_persist_kernel<NKT, GROUP, DENSE>;
  ensure_dynamic_lds(fn, SMEM, attr_once);
  const int n_cu = device_cus();
  const int nqt = (L + 31) / 32;
  const int nw = nqt < 4 ? 4 : nqt;                  // <= 7 here
  // (three workgroups per CU fit the three-key-tile instantiations -- 48 KiB, 146-153 VGPRs -- and were measured: no faster,
  // profiles/r04_text_attention.txt)
  const int per_cu = SMEM <= 80 * 1024 ? 2 : 1;
  const int n_items = N * H;
  const int grid = n_items < n_cu * per_cu ? n_items : n_cu * per_cu;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(nw * 64), SMEM, s, qkv, out, L, H, causal, n_items);
  return check_launch("attention_persist_kernel");
}

}  // namespace

int launch_attention(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s) {
  if (N == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(qkv && out, CLIPMI_ERR_ARG, "attention: null pointer");
  CLIPMI_REQUIRE(N > 0 && L > 0 && H > 0, CLIPMI_ERR_SHAPE, "attention: bad shape N=%d L=%d H=%d", N, L, H);
  CLIPMI_REQUIRE((int64_t)N * H < (1ll << 31), CLIPMI_ERR_SHAPE, "attention: grid too large");
  CLIPMI_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 8 == 0, CLIPMI_ERR_ARG, "attention: unaligned pointer");
  if (L <= 32 && options().attn_small.load(std::memory_order_relaxed) != 0) return launch_small(qkv, out, N, L, H, causal, s);
  if (L <= 96) {
    if (causal && L > 64) return launch_persist<3, 3, 2>(qkv, out, N, L, H, causal, s);
    return launch_persist<3, 3, 0>(qkv, out, N, L, H, causal, s);
  }
  if (L <= 224) {
    if (!causal && L > 192) {
      // 193..200 tokens: every operand by DMA from a loader wave (attention_vision_kernel); wider rows do not fit the LDS.
      // Option attn_loader = 0 keeps the persistent kernel (same bits: the bit-identity reference of the tests).
      const int loader = options().attn_loader.load(std::memory_order_relaxed);
      if (L <= VROWS && loader != 0) return loader == 2 ? launch_vision<true>(qkv, out, N, L, H, s) : launch_vision<false>(qkv, out, N, L, H, s);
      return launch_persist<7, 4, 1>(qkv, out, N, L, H, causal, s);
    }
    return launch_persist<7, 4, 0>(qkv, out, N, L, H, causal, s);
  }
  // Longer than one key block (ViT-L/14: 257 tokens, ViT-L/14@336: 577).  Non-causal (every CLIP tower of that length): the ring kernel --
  // persistent workgroups, 128-key blocks through a three-slot LDS ring, pinned fragment reads, query tiles in passes of RNW = 11, a short last
  // pass split by key tile.  Option attn_ring = 0, a causal mask, or a shape outside its limits keep the round-1 streaming kernel
  // (two-slot ring of 128- / 224-key blocks; the reference of the ring kernel's parity test).
  if (!causal && options().attn_ring.load(std::memory_order_relaxed) != 0 && L <= RMAXPASS * RNW * 32 && (int64_t)L * 3 * H * 64 * 2 < (1ll << 31))
    return launch_ring(qkv, out, N, L, H, s);
  return L <= 320 ? launch_stream<4>(qkv, out, N, L, H, causal, s) : launch_stream<7>(qkv, out, N, L, H, causal, s);
}

}  // namespace clipmi
