// 256x256x64 bf16 GEMM tile, software-pipelined: ONE barrier per K step, placed where it cannot starve the
// matrix pipe.
//
// Eight waves (2 along M x 4 along N, wave tile 128 x 64).  A K step is four phases of 16 MFMAs
// (phase = (m half, 32-deep k slice)); the fragments of phase p+1 are read from LDS while phase p's MFMAs
// run (two register sets), so a wave never waits on LDS at a phase boundary.  Per K step kt:
//
//   phase 0: read frags(kt, ph1)      | DMA W pieces of tile kt+2 | MFMA ph0
//   phase 1: read frags(kt, ph2)      |                           | MFMA ph1
//   phase 2: read frags(kt, ph3)      |                           | MFMA ph2
//            wait: LDS reads of tile kt done, my DMA pieces of tile kt+1 landed;  BARRIER
//   phase 3: read frags(kt+1, ph0)    | DMA A pieces of tile kt+2 | MFMA ph3   (operands already in registers)
//
// Workgroups are persistent (one per CU): after an output tile's last K step the first two K tiles of the
// workgroup's next output tile are put in flight before the epilogue, so the epilogue's stores hide their
// latency.
//
// The barrier publishes tile kt+1 (every wave waited for its own DMA pieces) and retires tile kt's buffer
// (every wave waited for its own reads of it), so that buffer can take tile kt+2's DMA right away; those
// loads then have a whole K step (~2000 cycles) to land before they are waited for.  Phase 3's MFMAs need
// nothing that the barrier guards, so the matrix pipe has work the moment the barrier opens.
#pragma once
#include "gemm_tile.h"

// BN: tile width in output features (a multiple of 16 * (8 / WR)); WR: waves along M (2 or 4).
// <256, 2> is the general shape (wave tile 128 x 64).  <224, 4> and <288, 4> (wave tile 64 x 112 / 64 x 144)
// exist because the tile count should be a whole number of 256-CU rounds: with 12288 rows, N = 3584 is
// 2.6 rounds of 256-wide tiles but exactly 3 rounds of 224-wide ones, N = 4608 is 3.4 rounds of 256-wide
// tiles but exactly 3 of 288-wide ones.
// AMODE = A_CONV3X3 (even split only): the A rows are output pixels of an NHWC image and K runs over (tap, channel),
// as in gemm_bf16_nt_kernel; a piece's source pixel is carried across K steps and re-derived when the tap changes.
template <int EPI, int BN, int WR, bool LEAD = false, int AMODE = A_DENSE>
__global__ __launch_bounds__(512) void gemm256_pipelined_kernel(GemmParams p) {
  static_assert(AMODE == A_DENSE || !LEAD, "the gathered A operand uses the even DMA split");
  constexpr int BM = 256, NW = 8, WC = NW / WR, TM = BM / WR, TN = BN / WC, FM = TM / 16, FN = TN / 16, FH = FM / 2;
  static_assert(TN % 16 == 0 && FM % 2 == 0 && BN % 8 == 0, "wave tile must be whole 16x16 fragments");
  static_assert(EPI != EPI_SWIGLU || FN % 2 == 0, "SwiGLU pairs gate and up fragments");
  constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int W_PIECES = BN / 8, W_ITERS = (W_PIECES + NW - 1) / NW;  // 1-KiB DMA pieces of the W tile
  constexpr bool W_EVEN = W_PIECES % NW == 0;
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor builtins exist only in the device pass
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;

  // The workgroup is persistent: it walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x = number of
  // CUs, a multiple of the 8 XCDs, so a workgroup's tiles stay on its XCD's share of the tile order).
  const int ntiles = p.tiles_m * p.tiles_n;
  const int piece_row = lane >> 3;
  const int src_chunk = (lane & 7) ^ piece_row;
  int m0, n0;
  // ---- even split (LEAD false): every wave stages an eighth of each tile.  A piece's address is a scalar base (the
  // operand + the K step's offset, both uniform) + one 32-bit lane offset that is fixed for the tile, so the pieces
  // of a tile cost one register each and the K step's advance is scalar arithmetic.
  unsigned a_src[4], w_src[W_ITERS];
  int a_oy[4], a_ox[4], conv_c0 = 0, conv_tap = -1;     // A_CONV3X3: output pixel of each piece's row; K-step state
  const bf16* conv_src[4];
  auto set_tile_even = [&](int t) {
    int tile_m, tile_n;
    gemm_tile_of_block(p, t, tile_m, tile_n);
    m0 = tile_m * BM;
    n0 = tile_n * BN;
#ifdef BAGEL_DIAG_SAME_TILE   // every workgroup loads tile (0, 0): what the kernel would do with no traffic beyond L2
    const int lm0 = 0, ln0 = 0;
#else
    const int lm0 = m0, ln0 = n0;
#endif
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = (it * NW + wave) * 8 + piece_row;
      if constexpr (AMODE == A_DENSE) {
        a_src[it] = (unsigned)(((long)min(lm0 + row, p.M - 1) * p.lda + src_chunk * 8) * 2);
      } else {
        const int grow = min(lm0 + row, p.M - 1);
        a_oy[it] = grow / p.Wout;
        a_ox[it] = grow - a_oy[it] * p.Wout;
      }
    }
    conv_tap = -1;
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it) {
      const int row = (it * NW + wave) * 8 + piece_row;
      w_src[it] = (unsigned)(((long)min(ln0 + row, p.N - 1) * p.ldw + src_chunk * 8) * 2);
    }
  };
  auto conv_set_tap = [&](int tap) {
    conv_tap = tap;
    const int tap_dy = tap / 3 - 1, tap_dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int y = a_oy[it] + tap_dy, x = a_ox[it] + tap_dx;
      bool ok = y >= 0 && y < p.Hout && x >= 0 && x < p.Wout;
      if (p.upsample == 1) {
        y >>= 1;
        x >>= 1;
      } else if (p.upsample == 2) {
        y = 2 * a_oy[it] + tap_dy + 1;
        x = 2 * a_ox[it] + tap_dx + 1;
        ok = y < p.Hin && x < p.Win;
      }
      conv_src[it] = ok ? p.A + ((long)y * p.Win + x) * p.Cin + src_chunk * 8 : nullptr;
    }
  };
  auto stage_a_even = [&](int kt) {
    char* base = smem + (kt & 1) * STAGE_BYTES;
    if constexpr (AMODE == A_DENSE) {
      const char* src = (const char*)p.A + kt * (BK * 2);
#pragma unroll
      for (int it = 0; it < 4; ++it)
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + a_src[it]), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
    } else {
      // called with kt = 0, 1, 2, ... for every tile: the tap and the pieces' pixels change every Cin / 64 calls
      if (conv_tap < 0) {
        conv_c0 = 0;
        conv_set_tap(0);
      } else {
        conv_c0 += BK;
        if (conv_c0 >= p.Cin) {
          conv_c0 = 0;
          conv_set_tap(conv_tap + 1);
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const bf16* src = conv_src[it] != nullptr ? conv_src[it] + conv_c0 : p.zeros + src_chunk * 8;
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
      }
    }
  };
  auto stage_w_even = [&](int kt) {
    char* base = smem + (kt & 1) * STAGE_BYTES + A_BYTES;
    const char* src = (const char*)p.W + kt * (BK * 2);
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it)
      if (W_EVEN || it + 1 < W_ITERS || it * NW + wave < W_PIECES)  // wave-uniform
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + w_src[it]), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
  };

  // ---- leading-wave split (LEAD true)
  // DMA pieces are dealt unevenly.  Waves w and w + 4 share a SIMD, and the older one (w < 4) wins the MFMA
  // arbitration, runs ahead and then idles at the K step's barrier while the younger one finishes alone:
  // every cycle the younger wave spends issuing DMA is then a cycle with an empty matrix pipe.  So the
  // leading waves issue (LEAD ? 8 : 4)/8 of the pieces.
  // Pieces are fetched through buffer descriptors based at the tile's first row (address = base + a per-lane
  // 32-bit offset fixed for the tile + a scalar K offset), which keeps a piece at one register.
  constexpr int A_PIECES = BM / 8;
  constexpr int LA = A_PIECES * (LEAD ? 8 : 4) / 8 / 4, TA = (A_PIECES - 4 * LA) / 4;                // per leading / trailing wave
  constexpr int LW = (W_PIECES * (LEAD ? 8 : 4) / 8 + 3) / 4, TW = (W_PIECES - 4 * LW + 3) / 4 > 0 ? (W_PIECES - 4 * LW + 3) / 4 : 0;
  constexpr int NA = LA > TA ? LA : TA, NWI = LW > TW ? LW : TW;
  const bool lead = wave < 4;
  auto a_piece = [&](int it) { return lead ? (it < LA ? it * 4 + wave : -1) : (it < TA ? LA * 4 + it * 4 + wave - 4 : -1); };
  auto w_piece = [&](int it) {
    const int pc = lead ? (it < LW ? it * 4 + wave : -1) : (it < TW ? LW * 4 + it * 4 + wave - 4 : -1);
    return pc < W_PIECES ? pc : -1;
  };
  __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
  int a_voff[NA], w_voff[NWI];
  auto set_tile_lead = [&](int t) {
    int tile_m, tile_n;
    gemm_tile_of_block(p, t, tile_m, tile_n);
    m0 = tile_m * BM;
    n0 = tile_n * BN;
    a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (long)m0 * p.lda), 0, 0x7fffffff, 0x00020000);
    w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int row = max(a_piece(it), 0) * 8 + piece_row;
      a_voff[it] = (int)(((long)min(row, p.M - 1 - m0) * p.lda + src_chunk * 8) * 2);
    }
#pragma unroll
    for (int it = 0; it < NWI; ++it) {
      const int row = max(w_piece(it), 0) * 8 + piece_row;
      w_voff[it] = (int)(((long)min(row, p.N - 1 - n0) * p.ldw + src_chunk * 8) * 2);
    }
  };
  auto stage_a_lead = [&](int kt) {
    char* base = smem + (kt & 1) * STAGE_BYTES;
    if (lead) {  // wave-uniform
#pragma unroll
      for (int it = 0; it < LA; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, LDS_PTR(base + (it * 4 + wave) * 1024), 16, a_voff[it],
                                                 kt * (BK * 2), 0, 0);
    } else {
#pragma unroll
      for (int it = 0; it < TA; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, LDS_PTR(base + (LA * 4 + it * 4 + wave - 4) * 1024), 16,
                                                 a_voff[it], kt * (BK * 2), 0, 0);
    }
  };
  auto stage_w_lead = [&](int kt) {
    char* base = smem + (kt & 1) * STAGE_BYTES + A_BYTES;
    if (lead) {
#pragma unroll
      for (int it = 0; it < LW; ++it)
        if (it * 4 + 3 < W_PIECES || it * 4 + wave < W_PIECES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(base + (it * 4 + wave) * 1024), 16, w_voff[it],
                                                   kt * (BK * 2), 0, 0);
    } else {
#pragma unroll
      for (int it = 0; it < TW; ++it)
        if (LW * 4 + it * 4 + 3 < W_PIECES || LW * 4 + it * 4 + wave - 4 < W_PIECES)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(base + (LW * 4 + it * 4 + wave - 4) * 1024), 16,
                                                   w_voff[it], kt * (BK * 2), 0, 0);
    }
  };

  auto set_tile = [&](int t) { if constexpr (LEAD) set_tile_lead(t); else set_tile_even(t); };
  auto stage_a = [&](int kt) { if constexpr (LEAD) stage_a_lead(kt); else stage_a_even(kt); };
  auto stage_w = [&](int kt) { if constexpr (LEAD) stage_w_lead(kt); else stage_w_even(kt); };

  const int frow = lane & 15, fq = lane >> 4;
  int a_off[2], w_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int chunk = (kk * 4 + fq) ^ (frow & 7);
    a_off[kk] = (wr * TM + frow) * ROW_BYTES + chunk * 16;
    w_off[kk] = A_BYTES + (wc * TN + frow) * ROW_BYTES + chunk * 16;
  }

  f32x4 acc[FN][FM];

  bf16x8 af[2][FH], wf[2][FN];
  auto read_a = [&](const char* base, int set, int mh, int kk) {
#pragma unroll
    for (int j = 0; j < FH; ++j) af[set][j] = *(const bf16x8*)(base + a_off[kk] + (mh * FH + j) * 16 * ROW_BYTES);
  };
  auto read_w = [&](const char* base, int kk) {
#pragma unroll
    for (int i = 0; i < FN; ++i) wf[kk][i] = *(const bf16x8*)(base + w_off[kk] + i * 16 * ROW_BYTES);
  };
  auto mma = [&](int set, int mh, int kk) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FH; ++j)
        acc[i][mh * FH + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], af[set][j], acc[i][mh * FH + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nk = p.K / BK;
  auto stage_head = [&]() {  // K tiles 0 and 1 of the tile set_tile() selected
    stage_a(0);
    stage_w(0);
    if (nk > 1) {
      stage_a(1);
      stage_w(1);
    }
  };
  int t = blockIdx.x;
  // true when the previous tile's epilogue left exactly EPI_STORES store instructions in flight behind this tile's
  // first DMA pieces: the wait below then covers the DMA alone and the stores drain under the first K step
  bool stores_behind = false;
  constexpr int EPI_STORES = gemm_epilogue_wide_stores<EPI, FN, FM>();
  static_assert(EPI_STORES <= 63, "s_waitcnt vmcnt holds 6 bits");
#ifdef BAGEL_DIAG_CLOCK   // shader cycles and 100 MHz reference ticks around the workgroup's whole life
  const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  set_tile(t);
  stage_head();
  while (true) {
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K tiles 0 and 1 were issued before the previous tile's epilogue (or just above): publish tile 0.
    // vmcnt counts loads, LDS-DMA and stores together in issue order, so "all but the EPI_STORES youngest" is
    // exactly "every DMA piece has landed" when the epilogue issued that many stores after them.
    if (stores_behind) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(EPI_STORES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_a(smem, 0, 0, 0);
    read_w(smem, 0);

  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + (kt & 1) * STAGE_BYTES;
    const char* next = smem + ((kt + 1) & 1) * STAGE_BYTES;
    // phase 0: (mh 0, kk 0) from set 0; prefetch (mh 0, kk 1) into set 1
    read_a(base, 1, 0, 1);
    read_w(base, 1);
    if (kt >= 1 && kt + 1 < nk) stage_w(kt + 1);  // its A pieces went out in the previous step's phase 3
    mma(0, 0, 0);
    // phase 1: (mh 0, kk 1) from set 1; prefetch (mh 1, kk 0) into set 0
    read_a(base, 0, 1, 0);
    mma(1, 0, 1);
    // phase 2: (mh 1, kk 0) from set 0; prefetch (mh 1, kk 1) into set 1
    read_a(base, 1, 1, 1);
    mma(0, 1, 0);
    // all my reads of this tile are issued: wait for them and for my DMA pieces of the next tile, then sync
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // phase 3: (mh 1, kk 1) from set 1; prefetch the next tile's phase 0; refill this tile's buffer
    if (kt + 1 < nk) {
      read_a(next, 0, 0, 0);
      read_w(next, 0);
      if (kt + 2 < nk) stage_a(kt + 2);
    }
    mma(1, 1, 1);
  }

    // Both LDS stages are free once the last barrier has been passed (every wave waited for its own reads
    // before it): start the next tile's first two K tiles now, so they land while this tile's results are
    // being written out.
    const int out_m = m0 + wr * TM, out_n = n0 + wc * TN;
    const int tn = t + gridDim.x;
    if (tn < ntiles) {
      set_tile(tn);
      stage_head();
    }
#ifdef BAGEL_DIAG_NO_EPILOGUE   // keeps the accumulators live, writes (almost) nothing
    {
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (keep == 12345.678f) p.C[(long)out_m * p.ldc + out_n] = (bf16)keep;
    }
#else
    stores_behind = gemm_epilogue<EPI, FN, FM>(p, acc, out_m, out_n, lane);
#endif
    if (tn >= ntiles) break;
    t = tn;
  }
#ifdef BAGEL_DIAG_CLOCK
  if (p.diag != nullptr && tid == 0) {
    p.diag[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - diag_c0;
    p.diag[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
  }
#endif
#endif
}
