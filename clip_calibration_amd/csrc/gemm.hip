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
#include "common.h"

namespace clipmi {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * BK * 2;     // one operand tile, 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // activations + weights
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;  // 64 KiB -> 2 workgroups per CU

struct KArgs {
  const half_t* A; int64_t lda;
  const half_t* W; int64_t ldw;
  const float* bias;
  const float* residual;
  void* out; int64_t ldo;
  int M, N, K;
  const float* pos; int patches; int tokens;
  int tiles_n; int nwg;
};

__device__ __forceinline__ float quick_gelu(float t) { return t / (1.0f + __expf(-1.702f * t)); }

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(const KArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave & 1, wave_n = wave >> 1;

  // bijective XCD-aware remap: blockIdx % 8 labels the XCD; give each label a contiguous range of tiles
  const int bid = blockIdx.x;
  const int xcd = bid & 7, q = a.nwg >> 3, r = a.nwg & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tile_m = wg / a.tiles_n;
  const int tile_n = wg - tile_m * a.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- staging addresses: thread t, instruction i writes LDS 16-B slot p = i*256 + t of the tile;
  //      slot p holds row p>>3, data chunk (p&7) ^ ((row>>1)&7)  (XOR swizzle, applied on the source)
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  const half_t* xsrc[4];
  const half_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int mr = m0 + i * 32 + srow; mr = mr < a.M ? mr : a.M - 1;   // clamp: rows >= M are never stored
    int nr = n0 + i * 32 + srow; nr = nr < a.N ? nr : a.N - 1;
    xsrc[i] = a.A + (int64_t)mr * a.lda + schunk * 8;
    wsrc[i] = a.W + (int64_t)nr * a.ldw + schunk * 8;
  }
  const int lds_wave_off = wave * 1024;  // 64 lanes x 16 B

  auto stage = [&](int buf, int kt) {
    char* xs = smem + buf * STAGE_BYTES + lds_wave_off;
    char* ws = xs + TILE_BYTES;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(CLIPMI_GLOBAL_PTR(xsrc[i] + k0), CLIPMI_LDS_PTR(xs + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(CLIPMI_GLOBAL_PTR(wsrc[i] + k0), CLIPMI_LDS_PTR(ws + i * 4096), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside an operand tile): lane reads row (lane&15) of its 16-row tile,
  //      data chunk ks*4 + (lane>>4), stored at chunk ^ ((row>>1)&7)
  const int r16 = lane & 15, g4 = lane >> 4;
  const int swz = (r16 >> 1) & 7;
  int foff[2];
  foff[0] = r16 * 128 + (((0 + g4) ^ swz) << 4);
  foff[1] = r16 * 128 + (((4 + g4) ^ swz) << 4);
  const int xbase = wave_m * 64 * 128;
  const int wbase = TILE_BYTES + wave_n * 64 * 128;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = a.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt landed for every wave; everyone finished reading the other buffer
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 xf[4], wf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const f16x8*>(st + xbase + j * 2048 + foff[ks]);
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const f16x8*>(st + wbase + i * 2048 + foff[ks]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: acc[i][j][e] = C[m = m0 + wave_m*64 + j*16 + (lane&15)][n = n0 + wave_n*64 + i*16 + (lane>>4)*4 + e]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wave_m * 64 + j * 16 + r16;
    if (m >= a.M) continue;
    int64_t orow = m;
    const float* posrow = nullptr;
    if constexpr (EPI == EPI_PATCH_POS) {
      const int b = m / a.patches;
      const int t = m - b * a.patches + 1;
      orow = (int64_t)b * a.tokens + t;
      posrow = a.pos + (int64_t)t * a.N;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wave_n * 64 + i * 16 + g4 * 4;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if constexpr (EPI == CLIPMI_EPI_BIAS || EPI == CLIPMI_EPI_BIAS_QUICKGELU || EPI == CLIPMI_EPI_BIAS_RESIDUAL) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + n);
        v += b;
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_QUICKGELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
      }
      if constexpr (EPI == CLIPMI_EPI_BIAS_RESIDUAL) {
        const f32x4 rr = *reinterpret_cast<const f32x4*>(a.residual + orow * a.ldo + n);
        v += rr;
      }
      if constexpr (EPI == EPI_PATCH_POS) {
        const f32x4 pp = *reinterpret_cast<const f32x4*>(posrow + n);
        v += pp;
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

template <int EPI, bool OUT_F32>
int launch_one(const KArgs& k, hipStream_t s) {
  static bool attr_set = false;
  auto fn = gemm_f16_kernel<EPI, OUT_F32>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES) !=
        hipSuccess) {
      (void)hipGetLastError();
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(fn, dim3(k.nwg), dim3(256), SMEM_BYTES, s, k);
  return check_launch("gemm_f16_kernel");
}

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  CLIPMI_REQUIRE(a.A && a.W && a.out, CLIPMI_ERR_ARG, "gemm: null operand");
  CLIPMI_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, CLIPMI_ERR_SHAPE, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  CLIPMI_REQUIRE(a.K % BK == 0, CLIPMI_ERR_SHAPE, "gemm: K=%d must be a multiple of %d", a.K, BK);
  CLIPMI_REQUIRE(a.N % 4 == 0, CLIPMI_ERR_SHAPE, "gemm: N=%d must be a multiple of 4", a.N);
  CLIPMI_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldo % 4 == 0, CLIPMI_ERR_SHAPE,
                 "gemm: leading dimensions must keep 16-byte alignment (lda=%lld ldw=%lld ldo=%lld)", (long long)a.lda,
                 (long long)a.ldw, (long long)a.ldo);
  CLIPMI_REQUIRE(a.lda >= a.K && a.ldw >= a.K && a.ldo >= a.N, CLIPMI_ERR_SHAPE, "gemm: leading dimension too small");
  CLIPMI_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.out % 16 == 0),
                 CLIPMI_ERR_ARG, "gemm: operands must be 16-byte aligned");
  const bool f32 = a.out_dtype == CLIPMI_F32;
  CLIPMI_REQUIRE(f32 || a.out_dtype == CLIPMI_F16, CLIPMI_ERR_ARG, "gemm: bad out_dtype %d", a.out_dtype);

  KArgs k;
  k.A = a.A; k.lda = a.lda; k.W = a.W; k.ldw = a.ldw; k.bias = a.bias; k.residual = a.residual;
  k.out = a.out; k.ldo = a.ldo; k.M = a.M; k.N = a.N; k.K = a.K;
  k.pos = a.pos; k.patches = a.patches; k.tokens = a.tokens;
  const int tiles_m = (a.M + BM - 1) / BM;
  k.tiles_n = (a.N + BN - 1) / BN;
  const int64_t nwg = (int64_t)tiles_m * k.tiles_n;
  CLIPMI_REQUIRE(nwg < (1ll << 30), CLIPMI_ERR_SHAPE, "gemm: grid too large");
  k.nwg = (int)nwg;

  switch (a.epilogue) {
    case CLIPMI_EPI_NONE:
      return f32 ? launch_one<CLIPMI_EPI_NONE, true>(k, s) : launch_one<CLIPMI_EPI_NONE, false>(k, s);
    case CLIPMI_EPI_BIAS:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      return f32 ? launch_one<CLIPMI_EPI_BIAS, true>(k, s) : launch_one<CLIPMI_EPI_BIAS, false>(k, s);
    case CLIPMI_EPI_BIAS_QUICKGELU:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      return f32 ? launch_one<CLIPMI_EPI_BIAS_QUICKGELU, true>(k, s) : launch_one<CLIPMI_EPI_BIAS_QUICKGELU, false>(k, s);
    case CLIPMI_EPI_BIAS_RESIDUAL:
      CLIPMI_REQUIRE(a.bias && (uintptr_t)a.bias % 16 == 0, CLIPMI_ERR_ARG, "gemm: bias missing/unaligned");
      CLIPMI_REQUIRE(a.residual && (uintptr_t)a.residual % 16 == 0, CLIPMI_ERR_ARG, "gemm: residual missing/unaligned");
      CLIPMI_REQUIRE(f32, CLIPMI_ERR_ARG, "gemm: the residual stream is fp32");
      return launch_one<CLIPMI_EPI_BIAS_RESIDUAL, true>(k, s);
    case EPI_PATCH_POS:
      CLIPMI_REQUIRE(a.pos && a.patches > 0 && a.tokens > a.patches && f32, CLIPMI_ERR_ARG, "gemm: bad patch epilogue");
      return launch_one<EPI_PATCH_POS, true>(k, s);
    default:
      set_error("gemm: unknown epilogue %d", a.epilogue);
      return CLIPMI_ERR_ARG;
  }
}

}  // namespace clipmi
