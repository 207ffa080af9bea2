// Persistent row-range kernel of the fp16-stream residual GEMMs (out-proj, c_proj; variant 16 of launch_gemm, gemm.hip).
#include "gemm_common.h"

// build-time A/B of the cache policy of the ACTIVATION operand's LDS-DMA (out-proj reads the attention rows exactly once: profiles/r06_attn_outproj_pair.txt)
#ifndef CLIPMI_RSTREAM_A_AUX
#define CLIPMI_RSTREAM_A_AUX 0
#endif

namespace clipmi {
namespace gemm {
namespace {

// ---------------------------------------------------------------------------------------------------------------
// Persistent ROW-RANGE kernel for the fp16-stream residual GEMMs (EPI_RESIDUAL_FOLD16: out-proj, c_proj; round 3):
//     x16[m, n] = fp16(x16[m, n] + bias[n] + A[m, :] . W[n, :])  in place,  + the LayerNorm-fold row partials of the rounded rows.
//
// Why.  gemm_pp_kernel runs these shapes as 320 x 256 tiles, one per workgroup, 1.85 rounds of 256 CUs; phase stamps
// (profiles/r02_gemm_stamps_final.txt) put an out-proj tile at 1.3 us prologue + 20.4 us main loop + 11.3 us epilogue with the
// matrix pipe idle: the residual rows come in and the sums go out as one burst per round (84 MB at once, 7.4 TB/s: bandwidth
// bound), every CU reaches that burst at the same moment, and the next workgroup cannot start before the stores have drained.
// A first form of this kernel (profiles/r03_rstream_slots_ab.txt) held the output slices in registers and streamed them, and the
// residual, through the next tile's K loop -- which needs tiles of <= 224 rows, i.e. three per CU, and a K-step costs ~3300
// cycles whatever the tile height (the load part of a phase, not the matrix pipe, sets it): 36 K-steps against 24 lost.
//
// What.  One workgroup per CU (eight waves, two per SIMD, the four-phase ping-pong loop of gemm_pp_kernel).  The M x N problem is cut
// into (row range, 256-column tile) UNITS, one per workgroup: the rows in `groups` near-equal ranges of 32-row pairs, every range
// taken by tiles_n workgroups with adjacent ids (same XCD: the activation rows reach its L2 once).  A workgroup walks its range in
// TILES of 32 nb rows, nb <= 10, equal to within one pair (M = 50 432, N = 768 on 256 CUs: 85 ranges of 18-19 pairs = tiles of 10 + 9
// or 9 + 9 pairs: every CU carries 18-19 blocks through 2 x nk K-steps, where the 320-row grid gives 218 CUs 20 and 38 CUs 10).
//   * The residual is PRELOADED INTO THE ACCUMULATORS: a tile starts from acc = bias + residual (fp32) and the MFMAs add the products
//     on top, so the tile end needs no operand -- round, row partials, store.  The residual of tile i + 1 is loaded (two 16-byte loads
//     per 16 x 64 slice, the layout the stores use) while tile i is being converted, into the registers its accumulators free.
//   * Stores are the YOUNGEST operations in the queue: every load and every LDS-DMA piece of the next tile is issued first, all
//     stores of the finished tile last, and the wait that opens the next tile is vmcnt(#stores) -- the stores drain under its K loop.
//   * The next tile's first stage is DMA'd before the conversion starts; the row partials of the four column waves meet in LDS and
//     are reduced behind the tile's closing barrier.
// Arithmetic: (bias + residual) + sum_k products in K order, one rounding to fp16 -- the same sum as the tile kernels' (sum_k
// products + bias) + residual in a different fp32 order: results agree with gemm_pp_kernel to the last fp16 bit on all but a few
// elements in ten thousand (a rounding boundary), never more than one ulp; run-to-run bit-identical
// (tests/test_gpu_ops.py::test_gemm_residual_f16_vs_reference, ::test_gemm_residual_stream_race_screen).
//
// K loop (as gemm_pp_kernel): a K-step is four phases per wave (k-half ks = p >> 1, row half jh = p & 1): a LOAD part -- the
// phase's fragments by pinned LDS reads (the half's activation blocks, plus the 4 weight blocks when jh == 0) and, in phases 0..2, a
// third of this wave's nine LDS-DMA pieces of the next stage -- and a COMPUTE part of 4 x (blocks of the half) MFMAs on registers
// only, a workgroup barrier after each part; waves 4-7 run one part behind waves 0-3.  A wave's nb blocks are split h0 = ceil(nb / 2)
// | h1 = nb - h0 over the two row halves; block b < 5 of the accumulator array is block b of half 0, block b >= 5 block b - 5 of
// half 1 (uniform branches skip the blocks a shorter tile does not have).
// Needs K >= 2 K-steps, N % 8 == 0.
// ---------------------------------------------------------------------------------------------------------------
struct RStream {
  static constexpr int TM = 10, H = 5, TN = 4, NT = 512;
  static constexpr int XB = 32 * TM * 128, WB = 256 * 128, STAGE = XB + WB;   // 320 activation rows + 256 weight rows of 64 k
  static constexpr int BIAS_OFF = 2 * STAGE, RED_OFF = BIAS_OFF + 256 * 4;
  static constexpr int SMEM = RED_OFF + 4 * 32 * TM * 8;                      // + [4 column waves][320 rows] (sum, sumsq)
  static_assert(SMEM <= 160 * 1024, "row-range kernel LDS");
};

__global__ __launch_bounds__(512, 2) void gemm_rstream_kernel(const KArgs a, const int groups, const int pairs) {
  using R = RStream;
  constexpr int TM = R::TM, H = R::H, TN = R::TN;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave & 1, wave_n = wave >> 1;
  const int grp = wave >> 2;   // uniform: waves w and w + 4 share a SIMD
  const int r16 = lane & 15, g4 = lane >> 4;
  const int nk = a.K / BK;

  // ---- this workgroup's unit: logical id u (contiguous ranges of ids per XCD label, as tile_coords) -> (row range g, column tile tn)
  int u;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = a.nwg >> 3, r = a.nwg & 7;
    u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int g = u / a.tiles_n, tn = u - g * a.tiles_n;
  const int n0 = tn * 256;
  const int p_lo = (int)((int64_t)g * pairs / groups), p_hi = (int)((int64_t)(g + 1) * pairs / groups);
  const int len = p_hi - p_lo;                              // >= 1 (launcher)
  const int n_tiles = (len + TM - 1) / TM, nb_base = len / n_tiles, nb_rem = len - nb_base * n_tiles;
  // tile k of the range: nb = nb_base + (k < nb_rem) pairs, the taller tiles first
  auto tile_nb = [&](int k) { return nb_base + (k < nb_rem ? 1 : 0); };
  auto tile_m0 = [&](int k) { return (p_lo + k * nb_base + (k < nb_rem ? k : nb_rem)) * 32; };

  // ---- staging: wave w, piece p covers rows 64 p + 8 w .. + 7 of an operand's stage image (1 KiB), XOR swizzle on the source
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
  const int xoff0 = (srow * (int)a.lda + schunk * 8) * 2, woff0 = (srow * (int)a.ldw + schunk * 8) * 2;
  const int xstep = 64 * (int)a.lda * 2, wstep = 64 * (int)a.ldw * 2;
  auto row_off = [](int base, int add) {
    int r;
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(base), "s"(add));
    return r;
  };
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(a.W + (int64_t)n0 * a.ldw, ((int64_t)(a.N - n0) * a.ldw) * 2);   // rows >= N read as zero
  // piece P = 0..4: activations (only the rows this tile has), 5..8: weights
  auto piece = [&](auto p_tag, const __amdgpu_buffer_rsrc_t& xrs, int nbx, int buf, int kt) {
    constexpr int P = decltype(p_tag)::value;
    if constexpr (P < 5) {
      if (64 * P + wave * 8 < 32 * nbx)   // uniform
        buffer_load_lds16_aux<CLIPMI_RSTREAM_A_AUX>(xrs, smem + buf * R::STAGE + P * 8192 + wave * 1024, row_off(xoff0, P * xstep), kt * BK * 2);
    } else {
      CLIPMI_BUFFER_LOAD_LDS16(wrs, smem + buf * R::STAGE + R::XB + (P - 5) * 8192 + wave * 1024, row_off(woff0, (P - 5) * wstep), kt * BK * 2);
    }
  };
  auto stage_all = [&](const __amdgpu_buffer_rsrc_t& xrs, int nbx, int buf) {   // a tile's first stage: every wave its nine pieces
    piece(std::integral_constant<int, 0>{}, xrs, nbx, buf, 0); piece(std::integral_constant<int, 1>{}, xrs, nbx, buf, 0);
    piece(std::integral_constant<int, 2>{}, xrs, nbx, buf, 0); piece(std::integral_constant<int, 3>{}, xrs, nbx, buf, 0);
    piece(std::integral_constant<int, 4>{}, xrs, nbx, buf, 0); piece(std::integral_constant<int, 5>{}, xrs, nbx, buf, 0);
    piece(std::integral_constant<int, 6>{}, xrs, nbx, buf, 0); piece(std::integral_constant<int, 7>{}, xrs, nbx, buf, 0);
    piece(std::integral_constant<int, 8>{}, xrs, nbx, buf, 0);
  };

  // ---- fragment read offsets
  const int swz = (r16 >> 1) & 7;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t f0 = (uint32_t)(r16 * 128 + (((0 + g4) ^ swz) << 4)), f1 = (uint32_t)(r16 * 128 + (((4 + g4) ^ swz) << 4));
  const uint32_t wb = (uint32_t)(R::XB + wave_n * 64 * 128);

  // ---- 16 x 64 slices of this wave's part: store / load layout (gemm_stream_kernel: v_permlane16_swap on block pairs -> 8 consecutive
  // columns per lane, two 16-byte accesses per slice).  voffset = lane part + scalar part; out of range = dropped / zero.
  half_t* x16 = a.x16;
  const int lcol = (g4 & 1) * 16 + (g4 >> 1) * 8;
  const int st_lane = (r16 * (int)a.ldo + lcol) * 2;
  const bool col_ok[2] = {n0 + wave_n * 64 + lcol < a.N, n0 + wave_n * 64 + lcol + 32 < a.N};
  // accumulator block b of a tile of nbx pairs: half 0 holds h0 = ceil(nbx / 2) blocks (b = 0 .. h0 - 1), half 1 the other nbx - h0
  // (b = 5 ..); position = its index among the wave's nbx live blocks, i.e. rows 16 pos .. of the wave's part, which starts at row
  // wave_m * 16 nbx of the tile
  // (a tile has nbx >= 8 pairs, so only blocks 4 and 9 can be absent; nbx = 0 stands for "no tile": nothing is live)
  auto blk_live = [&](int nbx, int b) { return nbx > 0 && (b < H ? b < (nbx + 1) / 2 : b - H < nbx / 2); };
  auto blk_pos = [&](int nbx, int b) { return b < H ? b : (nbx + 1) / 2 + b - H; };
  auto slice_voff = [&](int nbx, int b, int pp) {
    const int in_range = row_off(st_lane, ((wave_m * nbx + blk_pos(nbx, b)) * 16 * (int)a.ldo + wave_n * 64) * 2 + pp * 64);
    return (blk_live(nbx, b) && col_ok[pp]) ? in_range : (int)0xFFFFFFF0;
  };
  auto tile_rsrc = [&](int m0x) { return make_rsrc(x16 + (int64_t)m0x * a.ldo + n0, ((int64_t)(a.M - m0x) * a.ldo - n0) * 2); };
  auto pack_slice = [&](const f16x4 (&v)[TN], u32x4 (&o)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const u32x2 lo = __builtin_bit_cast(u32x2, v[2 * p]), hi = __builtin_bit_cast(u32x2, v[2 * p + 1]);
      const auto r0 = __builtin_amdgcn_permlane16_swap(lo[0], hi[0], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(lo[1], hi[1], false, false);
      o[p] = u32x4{r0[0], r1[0], r0[1], r1[1]};
    }
  };
  auto unpack_slice = [&](const u32x4 (&o)[2], f16x4 (&v)[TN]) {   // the swap is its own inverse
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const auto r0 = __builtin_amdgcn_permlane16_swap(o[p][0], o[p][2], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(o[p][1], o[p][3], false, false);
      v[2 * p] = __builtin_bit_cast(f16x4, u32x2{r0[0], r1[0]});
      v[2 * p + 1] = __builtin_bit_cast(f16x4, u32x2{r0[1], r1[1]});
    }
  };

  f32x4 acc[TN][TM];
  // bias of this lane's columns: pinned LDS reads (the compiler would put a vmcnt(0) in front of an ordinary one)
  auto load_bias = [&](f32x4 (&bb)[TN]) {
    const uint32_t ba = lds_base + (uint32_t)(R::BIAS_OFF + (wave_n * 64 + g4 * 4) * 4);
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:192\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(bb[0]), "=&v"(bb[1]), "=&v"(bb[2]), "=&v"(bb[3]) : "v"(ba));
  };
  auto start_tile = [&]() {   // acc = bias: the MFMAs add the products on top, the residual arrives during the K loop
    f32x4 bb[TN];
    load_bias(bb);
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int i = 0; i < TN; ++i) acc[i][b] = bb[i];
    // the accumulators are MFMA sources (SrcC) of the K-step that follows, and these are VALU copies: have every copy made HERE (the
    // empty statements name the registers), then the wait states of CLIPMI_VALU_TO_MFMA_FENCE (common.h) -- hipcc sank some of the
    // copies to three instructions in front of the first MFMA that reads them (tools/mfma_hazard_scan.py)
    static_assert(TM == 10, "the statements below name TM accumulators each");
#pragma unroll
    for (int i = 0; i < TN; ++i)
      asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]), "+v"(acc[i][4]), "+v"(acc[i][5]), "+v"(acc[i][6]),
                   "+v"(acc[i][7]), "+v"(acc[i][8]), "+v"(acc[i][9]));
    asm volatile("s_nop 3");
  };

  // ---- first tile: bias of this unit's 256 columns (once), stage 0
  int kt_tile = 0;
  int m0 = tile_m0(0), nb = tile_nb(0);
  __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.A + (int64_t)m0 * a.lda, ((int64_t)(a.M - m0) * a.lda) * 2);
  __amdgpu_buffer_rsrc_t ors = tile_rsrc(m0);   // this tile's rows of the stream: residual in, sums out
  int first_buf = 0;
  if (wave == 0) {
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(a.bias + n0, (int64_t)(a.N - n0) * 4);
    CLIPMI_BUFFER_LOAD_LDS16(brs, smem + R::BIAS_OFF, lane * 16, 0);
  }
  stage_all(xrs, nb, 0);
#ifdef CLIPMI_TUNING
  const bool stamp = a.stamps != nullptr && tid == 0;
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage 0, the bias
  __builtin_amdgcn_s_barrier();
  start_tile();

  // The residual TRICKLES IN during the tile's own K loop: K-step c (c = 0 .. 4) requests slices 2c, 2c + 1 of this wave's part
  // (four 16-byte loads into `rin`, issued behind the K-step's pieces and left in flight by its counted wait), K-step c + 1 adds them
  // to the accumulators in its third load part (the compiler puts its own counted vmcnt in front of the first use: VMEM returns in
  // order) and re-uses the registers for the next request.
  u32x4 rin[2][2];
  constexpr int NCHUNK = TM / 2;
  auto kstep = [&](auto more_tag, auto req_tag, auto add_tag, int kt) {
    constexpr bool MORE = decltype(more_tag)::value;      // a next K-step exists: its stage is DMA'd during this one
    constexpr int REQ = decltype(req_tag)::value;         // >= 0: request residual chunk REQ in this K-step
    constexpr int ADD = decltype(add_tag)::value;         // >= 0: add residual chunk ADD (requested one K-step ago)
    constexpr int NREQ = REQ >= 0 ? 4 : 0;                // VMEM operations issued behind the pieces
    const int buf = (first_buf + kt) & 1;
    const int h0 = (nb + 1) >> 1, h1 = nb >> 1;
    const uint32_t sb = lds_base + (uint32_t)(buf * R::STAGE);
    const uint32_t xlo = sb + (uint32_t)(wave_m * nb * 16 * 128), xhi = xlo + (uint32_t)(h0 * 2048);
    const uint32_t wa0 = sb + wb + f0, wa1 = sb + wb + f1;
    f16x8 wf[4], xf[H];
    auto phase = [&](auto p_tag) {
      constexpr int P = decltype(p_tag)::value;
      constexpr int KS = P >> 1, JH = P & 1;
      const int hc = JH ? h1 : h0;   // live blocks of this half (uniform): 4 or 5
      // ---- load part
      {
        const uint32_t xa = (JH ? xhi : xlo) + (KS ? f1 : f0);
        // all five reads, whatever the half holds: a fragment register that one path leaves undefined would stay live across the whole
        // tile loop (16 registers the tile end needs); the surplus read costs one LDS access per phase in a 9-pair tile
        ds_read128<0 * 2048>(xf[0], xa); ds_read128<1 * 2048>(xf[1], xa); ds_read128<2 * 2048>(xf[2], xa);
        ds_read128<3 * 2048>(xf[3], xa); ds_read128<4 * 2048>(xf[4], xa);
        if constexpr (JH == 0) {
          const uint32_t wa = KS ? wa1 : wa0;
          ds_read128<0>(wf[0], wa); ds_read128<2048>(wf[1], wa); ds_read128<4096>(wf[2], wa); ds_read128<6144>(wf[3], wa);
        }
      }
      if constexpr (MORE && P < 3) {
        piece(std::integral_constant<int, P * 3 + 0>{}, xrs, nb, buf ^ 1, kt + 1);
        piece(std::integral_constant<int, P * 3 + 1>{}, xrs, nb, buf ^ 1, kt + 1);
        piece(std::integral_constant<int, P * 3 + 2>{}, xrs, nb, buf ^ 1, kt + 1);
      }
      if constexpr (P == 2 && ADD >= 0) {
        // the chunk requested a K-step ago: un-swap, add in fp32 (the last MFMA on these blocks was issued at least two parts ago, the
        // next one follows the barrier below)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int B = 2 * ADD + q;   // a constant after unrolling (ADD is one)
          if (blk_live(nb, B)) {   // uniform
            const u32x4 hv[2] = {rin[0][q], rin[1][q]};
            f16x4 res[TN];
            unpack_slice(hv, res);
#pragma unroll
            for (int i = 0; i < TN; ++i) add_f16x4(acc[i][B], res[i]);
          }
        }
      }
      if constexpr (P == 3 && REQ >= 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          rin[0][q] = __builtin_amdgcn_raw_buffer_load_b128(ors, slice_voff(nb, 2 * REQ + q, 0), 0, 0);
          rin[1][q] = __builtin_amdgcn_raw_buffer_load_b128(ors, slice_voff(nb, 2 * REQ + q, 1), 0, 0);
        }
      }
      if constexpr (JH == 0) lgkm_wait4<0>(wf[0], wf[1], wf[2], wf[3]);
      lgkm_wait_x<0, H>(xf);
      if constexpr (P == 3) {
        if (grp == 1) wait_vmcnt<NREQ>();   // this wave's pieces of the next stage have landed (the request behind them may be in flight)
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- compute part: registers only, accumulators tied to the destination (see gemm_stream_kernel).  Tiles have nb >= 8
      // (launcher): the first four blocks of either half always exist and run back to back; one uniform branch guards the fifth (a
      // branch in front of every block cost ~8 % of the loop: 2.0 us against 1.7 us per K-step of a 320-row tile)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < H; ++j) {
        if (j < H - 1 || hc == H) {
#pragma unroll
          for (int i = 0; i < TN; ++i)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][JH * H + j]) : "v"(wf[i]), "v"(xf[j]));
        }
      }
      __builtin_amdgcn_s_setprio(0);
      if constexpr (P == 3) {
        if (grp == 0) wait_vmcnt<NREQ>();
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

  while (true) {
#ifdef CLIPMI_TUNING
    long long* sp = a.stamps + ((size_t)u * 4 + (kt_tile < 3 ? kt_tile : 3)) * 8;
    if (stamp) { sp[0] = (long long)__builtin_amdgcn_s_memrealtime(); sp[5] = (long long)blockIdx.x; }
#endif
    constexpr std::false_type no{};
    constexpr std::true_type yes{};
    using N1 = std::integral_constant<int, -1>;
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // VALU-written accumulators -> the first asm MFMA that reads them
    if (grp == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 start one part later
    static_assert(NCHUNK == 5, "five residual chunks ride on K-steps 0..4 and are added in K-steps 1..5");
    kstep(yes, std::integral_constant<int, 0>{}, N1{}, 0);
#ifdef CLIPMI_TUNING
    if (stamp) { sp[1] = (long long)__builtin_amdgcn_s_memrealtime(); sp[6] = (long long)__builtin_amdgcn_s_memtime(); }
#endif
    kstep(yes, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, 1);
    kstep(yes, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, 2);
    kstep(yes, std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{}, 3);
    kstep(yes, std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{}, 4);
    kstep(yes, N1{}, std::integral_constant<int, 4>{}, 5);
    for (int kt = 6; kt < nk - 1; ++kt) kstep(yes, N1{}, N1{}, kt);   // nk >= 7 (launcher)
    kstep(no, N1{}, N1{}, nk - 1);
    if (grp == 0) __builtin_amdgcn_s_barrier();   // ... and waves 0-3 wait out the last compute part of waves 4-7
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the asm MFMAs' results are read by compiler-scheduled VALU code from here on
#ifdef CLIPMI_TUNING
    if (stamp) { sp[2] = (long long)__builtin_amdgcn_s_memrealtime(); sp[7] = (long long)__builtin_amdgcn_s_memtime(); }
#endif

    // ---- tile end: nothing is in flight (the last K-step waited for everything), the accumulators hold bias + residual + products.
    // The next tile's first stage is requested FIRST, the 2 TM stores of this tile follow it: the counted wait below covers exactly
    // the former, the stores drain under the next tile's K loop.
    const int last_buf = (first_buf + nk - 1) & 1;
    const int cm0 = m0, cnb = nb;
    const bool has_next = kt_tile + 1 < n_tiles;
    int m0n = m0, nbn = nb;
    if (has_next) {
      m0n = tile_m0(kt_tile + 1);
      nbn = tile_nb(kt_tile + 1);
      xrs = make_rsrc(a.A + (int64_t)m0n * a.lda, ((int64_t)(a.M - m0n) * a.lda) * 2);
      first_buf = last_buf ^ 1;   // the buffer that was NOT read last is free
      stage_all(xrs, nbn, first_buf);
    }
    const uint32_t red_lane = lds_base + (uint32_t)(R::RED_OFF + (wave_n * 32 * TM + wave_m * 16 * cnb + r16) * 8);
#pragma unroll
    for (int b = 0; b < TM; ++b) {
      if (blk_live(cnb, b)) {   // uniform; both arms issue two stores (the operation count behind the pieces stays static)
        f16x4 cv[TN];
        float rsum = 0.f, rsq = 0.f;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const f32x4 v = acc[i][b];
          cv[i] = f16x4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          fold_row_sums16(cv[i], rsum, rsq);   // the partials are those of the ROUNDED row
        }
        rsum = row4_sum(rsum);   // the 4 lanes of a row
        rsq = row4_sum(rsq);
        if (g4 == 0) {
          const float2 pr = make_float2(rsum, rsq);
          asm volatile("ds_write_b64 %0, %1" ::"v"(red_lane + (uint32_t)(blk_pos(cnb, b) * 128)), "v"(pr) : "memory");   // row 16 pos + r16 of this wave's part
        }
        u32x4 o[2];
        pack_slice(cv, o);
        __builtin_amdgcn_raw_buffer_store_b128(o[0], ors, slice_voff(cnb, b, 0), 0, CLIPMI_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(o[1], ors, slice_voff(cnb, b, 1), 0, CLIPMI_STORE_AUX);
      } else {
        const u32x4 z = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, ors, (int)0xFFFFFFF0, 0, 0);   // out of range: dropped
        __builtin_amdgcn_raw_buffer_store_b128(z, ors, (int)0xFFFFFFF0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);   // one slice at a time: the accumulators die as they are converted
    }
#ifdef CLIPMI_TUNING
    if (stamp) sp[3] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // the next tile's stage 0 has landed (the 2 TM stores behind it may still be on their way); this wave's row partials are in LDS
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * TM) : "memory");
    __builtin_amdgcn_s_barrier();
#ifdef CLIPMI_TUNING
    if (stamp) sp[4] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
    // ---- row partials of the finished tile: thread t owns row t, adds the four column waves in order (as the tile kernels do)
    if (tid < 32 * cnb) {
      const int m = cm0 + tid;
      if (m < a.M) {
        const float2* red = reinterpret_cast<const float2*>(smem + R::RED_OFF);
        float sx = 0.f, sq = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float2 pr = red[w * 32 * TM + tid];
          sx += pr.x;
          sq += pr.y;
        }
        *reinterpret_cast<float2*>(a.stats_out + 2 * ((int64_t)tn * a.M + m)) = make_float2(sx, sq);
      }
    }
    if (!has_next) break;
    ++kt_tile;
    m0 = m0n;
    nb = nbn;
    ors = tile_rsrc(m0);
    start_tile();
  }
}

// the row-range kernel takes a shape when every range splits into tiles of 8 .. 10 pairs of rows (its K loop runs the first four
// blocks of a row half unconditionally): ranges of 8-10, 16-20, 24-30 or >= 32 pairs
}  // namespace

bool rstream_fits(const KArgs& k) {
  const int n_cu = device_cus() & ~7;
  const int tiles_n = (k.N + 255) / 256;
  if (n_cu < 8 || tiles_n > n_cu || tiles_n > LN_MAX_PARTS) return false;
  const int groups = n_cu / tiles_n;
  const int64_t pairs = ((int64_t)k.M + 31) / 32;
  if (!(k.K >= 7 * BK && (k.N & 7) == 0 && (k.ldo & 7) == 0 && stream_offsets_ok(k))) return false;
  for (int64_t len = pairs / groups; len <= (pairs + groups - 1) / groups; ++len) {   // the two range lengths that occur
    if (len < 8) return false;
    const int64_t n_tiles = (len + RStream::TM - 1) / RStream::TM;
    if (len / n_tiles < 8) return false;
  }
  return true;
}

int launch_rstream(KArgs k, hipStream_t s) {
  static DeviceOnce attr_once;
  ensure_dynamic_lds(gemm_rstream_kernel, RStream::SMEM, attr_once);
  const int n_cu = device_cus() & ~7;
  k.tiles_n = (k.N + 255) / 256;
  const int groups = n_cu / k.tiles_n;
  const int pairs = (int)(((int64_t)k.M + 31) / 32);
  k.nwg = groups * k.tiles_n;
  k.band = k.tiles_n;
  hipLaunchKernelGGL(gemm_rstream_kernel, dim3(k.nwg), dim3(RStream::NT), RStream::SMEM, s, k, groups, pairs);
  return check_launch("gemm_rstream_kernel");
}

}  // namespace gemm
}  // namespace clipmi
