// 256x256x64 bf16 GEMM tile with the two wave groups of the workgroup running half a phase apart.
//
// Eight waves (2 along M x 4 along N, wave tile 128 x 64) share one CU; waves 0-3 and 4-7 sit pairwise on
// the four SIMDs.  In the plain one-barrier-per-K-step loop all eight waves run in lockstep: everybody
// reads fragments from LDS (matrix pipe idle), then everybody issues MFMAs (LDS idle) - measured 57 %
// MFMA busy.  Here a K step is cut into four phases of 16 MFMAs (one 64 x 64 quadrant x 32-deep k), each
// preceded by the LDS reads of exactly the fragments that phase needs, and wave group 1 runs one barrier
// behind group 0, so on every SIMD one wave's fragment reads / DMA issue overlap the other wave's MFMAs:
//
//      slot:      0      1      2      3      4      5      6      7      8 ...
//      group 0:  L0     C0     L1     C1     L2     C2     L3     C3     L0'
//      group 1:   -     L0     C0     L1     C1     L2     C2     L3     C3
//
// L = ds_read the phase's fragments (+ issue half of the next tile's LDS-DMA in L0/L1), C = 16 MFMAs.
// Every slot ends with a raw s_barrier executed by all eight waves.  Hazards:
//  * tile t+1 is DMA'd into the other LDS buffer from slot 8t on; that buffer's last readers (group 1,
//    L3 of tile t-1, slot 8t-1) waited lgkmcnt(0) before their barrier;
//  * tile t+1 is first read in slot 8t+8; in slot 8t+7 group 0 (C3) and group 1 (L3) both wait vmcnt(0)
//    for their own DMA pieces before the barrier, so every piece has landed and is visible.
#pragma once
#include "gemm_tile.h"

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_pingpong_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, NW = 8, TM = 128, TN = 64, FM = 8, FN = 4;
  constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, STAGE_BYTES = A_BYTES + W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int tile_m, tile_n;
  gemm_tile_of_block(p, blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;  // wave row (= group) and wave column

  // ---- LDS-DMA source addresses (pre-swizzled), 4 A pieces and 4 W pieces of 8 rows per wave and tile
  const int piece_row = lane >> 3;
  const int src_chunk = (lane & 7) ^ piece_row;
  const bf16* a_src[4];
  const bf16* w_src[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = (it * NW + wave) * 8 + piece_row;
    a_src[it] = p.A + (long)min(m0 + row, p.M - 1) * p.lda + src_chunk * 8;
    w_src[it] = p.W + (long)min(n0 + row, p.N - 1) * p.ldw + src_chunk * 8;
  }
  auto stage_a = [&](int kt, int buf) {
    char* base = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[it] + kt * BK), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
  };
  auto stage_w = [&](int kt, int buf) {
    char* base = smem + buf * STAGE_BYTES + A_BYTES;
#pragma unroll
    for (int it = 0; it < 4; ++it)
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(w_src[it] + kt * BK), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets
  const int frow = lane & 15, fq = lane >> 4;
  int a_off[2], w_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int chunk = (kk * 4 + fq) ^ (frow & 7);
    a_off[kk] = (grp * TM + frow) * ROW_BYTES + chunk * 16;
    w_off[kk] = A_BYTES + (wc * TN + frow) * ROW_BYTES + chunk * 16;
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage_a(0, 0);
  stage_w(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one slot behind

  bf16x8 wf[2][FN];
  for (int kt = 0; kt < nk; ++kt) {
    const char* base = smem + (kt & 1) * STAGE_BYTES;
    const bool more = kt + 1 < nk;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int mh = ph >> 1, kk = ph & 1;
      // ---------------- L: fragments of this phase (+ next tile's DMA in the first two phases)
      if (ph == 0 && more) stage_a(kt + 1, (kt + 1) & 1);
      if (ph == 1 && more) stage_w(kt + 1, (kt + 1) & 1);
      bf16x8 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *(const bf16x8*)(base + a_off[kk] + (mh * 64 + j * 16) * ROW_BYTES);
      if (mh == 0) {
#pragma unroll
        for (int i = 0; i < FN; ++i) wf[kk][i] = *(const bf16x8*)(base + w_off[kk] + i * 16 * ROW_BYTES);
      }
      if (ph == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---------------- C: one 64 x 64 quadrant, 32-deep k
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][mh * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][i], af[j], acc[i][mh * 4 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (ph == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();

  gemm_epilogue<EPI, FN, FM>(p, acc, m0 + grp * TM, n0 + wc * TN, lane);
}
