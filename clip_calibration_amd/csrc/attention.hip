// Multi-head self attention after the packed in-projection (reference clip/model.py:181-183 ->
// nn.MultiheadAttention -> scaled_dot_product_attention; SURVEY a-5a), head_dim 64, fp16 in/out, fp32 softmax.
//
// One workgroup per (sequence, head); each wave owns one 32-query tile.  Sequences here are short (77 / 197 /
// 199 / 257 / 577 tokens), so K and V of a key block (NKT*32 keys) are staged ONCE per workgroup into LDS with
// global_load_lds (K: 128-B rows, XOR swizzle for ds_read_b128 row reads; V: 128-B rows, a second XOR so that
// ds_read_b64_tr_b16 transposed reads spread over the bank row).  Q fragments come straight from global memory.
//
//   S^T tile = K_tile(32 keys x 64) * Q^T           v_mfma_f32_32x32x16_f16, A = K rows, B = Q rows
//   softmax over keys                                 keys live in the 16 accumulator registers x NKT tiles of a lane
//                                                     (its query is the lane's column) -> in-register max/sum, one
//                                                     cross-half shuffle; online rescale across key blocks
//   O^T tile += V^T(32 d x 16 keys) * P^T             the S^T accumulators, packed to fp16, ARE the B operand
//                                                     (cdna_hip_programming.md §3 "accumulator tile as the next
//                                                     MFMA's operand"); A = V^T via transposed LDS reads
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

template <int NKT, int GROUP, bool TR>
__global__ __launch_bounds__(512, 2) void attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                           int L, int H, int causal, int nkb) {
  constexpr int KEYS = NKT * 32;
  constexpr int KS_BYTES = KEYS * 128;
  constexpr int VT_STRIDE = NKT * 64 + 8;  // bytes per d-row of the transposed image (non-TR path)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + KS_BYTES;

  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, hh = lane >> 5;
  const int D = H * 64;
  const int64_t ld = 3 * (int64_t)D;
  const int n = blockIdx.x / H, h = blockIdx.x - n * H;
  const half_t* base = qkv + (int64_t)n * L * ld + h * 64;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, ((int64_t)L * ld - h * 64) * 2);

  const int q0 = (blockIdx.y * (nthr >> 6) + wave) * 32;
  const bool active = q0 < L;  // wave-uniform
  const int q = q0 + r32;
  const int qc = q < L ? q : L - 1;

  f16x8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const f16x8*>(base + (int64_t)qc * ld + ks * 16 + hh * 8);

  f32x16 oacc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  // lane-constant LDS read bases; everything else is an immediate offset
  const int kswz = (r32 >> 1) & 7;
  const char* kread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kread[ks] = Ks + r32 * 128 + (((2 * ks + hh) ^ kswz) << 4);
  const char* vread[2];
  if constexpr (TR) {
    // ds_read_b64_tr_b16: per 16-lane group a block of 4 rows (keys k0..k0+3) x 16 columns (d0..d0+15); lane 4q+p
    // of the group supplies the address of row q, columns 4p..4p+3 and receives column (lane&15) of the 4 rows.
    // k0 = kt*32 + ss*16 + hh*4 (+8), d0 = dt*32 + ((lane>>4)&1)*16.  V swizzle: 16-B chunk ^= ((row>>1)&1)<<2,
    // and (row>>1)&1 == (q>>1)&1 because k0 % 4 == 0.
    const int i16 = lane & 15, qq = i16 >> 2, pp = i16 & 3;
    const int fq = (qq >> 1) & 1;
    const int lane_base = hh * 512 + qq * 128 + ((((lane >> 4) & 1) * 2 + (pp >> 1)) << 4) + (pp & 1) * 8;
    vread[0] = Vs + lane_base + fq * 64;          // dt = 0: chunk bit 2 = 0 ^ fq
    vread[1] = Vs + lane_base + (1 - fq) * 64;    // dt = 1: chunk bit 2 = 1 ^ fq
  } else {
    vread[0] = Vs + r32 * VT_STRIDE + hh * 8;
    vread[1] = Vs + (32 + r32) * VT_STRIDE + hh * 8;
  }

  for (int kb = 0; kb < nkb; ++kb) {
    const int kb0 = kb * KEYS;
    if (kb > 0) __syncthreads();  // everyone finished reading the previous block
    // ---- stage K (and V) : slot p = row*8 + c' holds data chunk c' ^ swizzle(row)
    const int nwaves = nthr >> 6;
    for (int it = wave; it < KEYS / 8; it += nwaves) {  // one wave-instruction = 8 rows x 128 B; `it` is scalar
      const int pw = it * 64;                           // wave-uniform slot base
      const int p = pw + lane;
      const int row = p >> 3, cs = p & 7;
      const int key = kb0 + row;                      // rows >= L are outside the descriptor: they read as zero
      const int koff = (key * (int)ld + D) * 2;
      CLIPMI_BUFFER_LOAD_LDS16(rs, Ks + pw * 16, koff + ((cs ^ ((row >> 1) & 7)) << 4), 0);
      if constexpr (TR) {
        CLIPMI_BUFFER_LOAD_LDS16(rs, Vs + pw * 16, koff + D * 2 + ((cs ^ (((row >> 1) & 1) << 2)) << 4), 0);
      } else {
        const int kc = key < L ? key : L - 1;
        const f16x8 v = *reinterpret_cast<const f16x8*>(base + (int64_t)kc * ld + 2 * D + cs * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<half_t*>(Vs + (cs * 8 + e) * VT_STRIDE + row * 2) = v[e];
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (active) {
      // key tiles are consumed in groups of GROUP with an online-softmax rescale between groups: GROUP*16 live
      // score registers instead of NKT*16
#pragma unroll
      for (int g0 = 0; g0 < NKT; g0 += GROUP) {
        f32x16 s[GROUP];
        // ---- S^T = K Q^T
#pragma unroll
        for (int t = 0; t < GROUP; ++t) {
          const int kt = g0 + t;
          if (kt < NKT) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[t][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const f16x8 kf = *reinterpret_cast<const f16x8*>(kread[ks] + kt * 4096);
              s[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[t], 0, 0, 0);
            }
          }
        }
        // ---- scale, mask, group max
        float mloc = NEG_BIG;
#pragma unroll
        for (int t = 0; t < GROUP; ++t) {
          const int kt = g0 + t;
          if (kt < NKT) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int key = kb0 + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
              float v = s[t][e] * (0.125f * LOG2E);  // log2-domain scores
              const bool dead = (key >= L) || (causal && key > q);
              v = dead ? NEG_BIG : v;
              s[t][e] = v;
              mloc = fmaxf(mloc, v);
            }
          }
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float lsum = 0.f;
#pragma unroll
        for (int t = 0; t < GROUP; ++t) {
          const int kt = g0 + t;
          if (kt < NKT) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const float p = __builtin_amdgcn_exp2f(s[t][e] - m_new);
              s[t][e] = p;
              lsum += p;
            }
          }
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * alpha + lsum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
        // ---- O^T += V^T P^T
#pragma unroll
        for (int t = 0; t < GROUP; ++t) {
          const int kt = g0 + t;
          if (kt < NKT) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
              f16x8 pf;
#pragma unroll
              for (int j = 0; j < 8; ++j) pf[j] = (half_t)s[t][8 * ss + j];
#pragma unroll
              for (int dt = 0; dt < 2; ++dt) {
                f16x4 lo, hi;
                if constexpr (TR) {
                  lo = tr_read(vread[dt] + kt * 4096 + ss * 2048);
                  hi = tr_read(vread[dt] + kt * 4096 + ss * 2048 + 1024);
                } else {
                  lo = *reinterpret_cast<const f16x4*>(vread[dt] + kt * 64 + ss * 32);
                  hi = *reinterpret_cast<const f16x4*>(vread[dt] + kt * 64 + ss * 32 + 16);
                }
                const f16x8 vf = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[dt], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  if (active && q < L) {
    const float inv = 1.0f / l_run;
    half_t* orow = out + ((int64_t)n * L + q) * D + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)(oacc[dt][rr * 4 + e] * inv);
        *reinterpret_cast<f16x4*>(orow + dt * 32 + rr * 8 + hh * 4) = o;
      }
  }
}

template <int NKT, int GROUP, bool TR>
int launch_t(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s) {
  constexpr int KEYS = NKT * 32;
  constexpr int SMEM = KEYS * 128 + (TR ? KEYS * 128 : 64 * (NKT * 64 + 8));
  static bool attr_set = false;
  auto fn = attention_kernel<NKT, GROUP, TR>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  const int nqt = (L + 31) / 32;
  const int nw = nqt < 4 ? 4 : (nqt > 8 ? 8 : nqt);
  const int qsplit = (nqt + nw - 1) / nw;
  const int nkb = (L + KEYS - 1) / KEYS;
  hipLaunchKernelGGL(fn, dim3(N * H, qsplit), dim3(nw * 64), SMEM, s, qkv, out, L, H, causal, nkb);
  return check_launch("attention_kernel");
}

}  // namespace

// CLIPMI_ATTN_NO_TR=1 selects the register-transposed V image instead of ds_read_b64_tr_b16 (A/B + bring-up aid).
static bool use_tr() {
  const char* e = getenv("CLIPMI_ATTN_NO_TR");
  return !(e && e[0] == '1');
}

int launch_attention(const half_t* qkv, half_t* out, int N, int L, int H, int causal, hipStream_t s) {
  CLIPMI_REQUIRE(qkv && out, CLIPMI_ERR_ARG, "attention: null pointer");
  if (N == 0) return CLIPMI_OK;
  CLIPMI_REQUIRE(N > 0 && L > 0 && H > 0, CLIPMI_ERR_SHAPE, "attention: bad shape N=%d L=%d H=%d", N, L, H);
  CLIPMI_REQUIRE((int64_t)N * H < (1ll << 31), CLIPMI_ERR_SHAPE, "attention: grid too large");
  CLIPMI_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)out % 8 == 0, CLIPMI_ERR_ARG, "attention: unaligned pointer");
  const bool tr = use_tr();
  if (L <= 96) return tr ? launch_t<3, 3, true>(qkv, out, N, L, H, causal, s) : launch_t<3, 3, false>(qkv, out, N, L, H, causal, s);
  return tr ? launch_t<7, 4, true>(qkv, out, N, L, H, causal, s) : launch_t<7, 4, false>(qkv, out, N, L, H, causal, s);
}

}  // namespace clipmi
