// The 128x128-class MFMA GEMM tile shared by the dense projection GEMM (gemm.hip) and the implicit-GEMM
// 3x3 convolution of the VAE decoder (vae.hip).  See gemm.hip for the structure.
#pragma once
#include "common.h"

enum { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2, EPI_BIAS_RESIDUAL = 3, EPI_F32_SCALED = 4, EPI_BIAS_GELU = 5 };
enum { A_DENSE = 0, A_CONV3X3 = 1 };

struct GemmParams {
  const bf16* A; long lda;
  const bf16* W; long ldw;
  bf16* C; long ldc;
  const bf16* bias;
  const bf16* R; long ldr;
  int M, N, K;
  int tiles_m, tiles_n;
  float alpha;                 // EPI_F32_SCALED: C (fp32) = alpha * acc
  // ksplit > 1 (gemm_bf16_nt_kernel with EPI_F32_SCALED only): workgroup (x, y) multiplies K slice y of tile x and
  // writes its fp32 partial tile to slab y of C (slab = fp32 elements per slab, ldc = row stride inside a slab)
  int ksplit;
  long slab;
  // A_CONV3X3: A is an NHWC image [Hin][Win][Cin]; row m of the GEMM is output pixel (m / Wout, m % Wout),
  // K runs over (ky, kx, c); taps outside the image read `zeros`.  upsample: 0 = stride 1, padding 1;
  // 1 = the same on the 2x nearest-upsampled input; 2 = stride 2 with one zero row/column on the bottom/right.
  int Hout, Wout, Hin, Win, Cin, upsample;
  const bf16* zeros;
  unsigned long long* diag;    // diagnostic builds only (BAGEL_DIAG_*): per-workgroup stamps; never read otherwise
};

constexpr int BK = 64;            // bf16 elements per K step (128-byte LDS rows)
constexpr int ROW_BYTES = BK * 2;

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so each XCD gets a contiguous
// range of tiles, walked in bands of 8 M-tiles so a band's A rows and the W column panels stay in that
// XCD's L2.  Bijective for any tile count.
__device__ __forceinline__ void gemm_tile_of_block(const GemmParams& p, int block, int& tile_m, int& tile_n) {
  const int nwg = p.tiles_m * p.tiles_n;
  int wg = block;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
#ifndef GEMM_GROUP_M
#define GEMM_GROUP_M 8   // M-tiles per band; a diagnostic build may change it (the walk's fabric traffic is measured per value)
#endif
  constexpr int GROUP_M = GEMM_GROUP_M;
  const int band = wg / (GROUP_M * p.tiles_n);
  const int band_first = band * GROUP_M;
  const int band_rows = min(p.tiles_m - band_first, GROUP_M);
  const int in_band = wg - band * GROUP_M * p.tiles_n;
  tile_m = band_first + in_band % band_rows;
  tile_n = in_band / band_rows;
}

// Epilogue shared by the tile kernels: this lane holds, for fragment (i, j), features
// n_base + i*16 + fq*4 .. +3 of token row m_base + j*16 + frow  (fq = lane >> 4, frow = lane & 15).
//
// Two things keep it short (it runs with the matrix pipe idle, on every CU at the same moment):
//  * Loads first, stores last.  The residual usually IS the output buffer (x += f(x)), so the compiler keeps every
//    residual load behind every earlier store; fragment by fragment that is one exposed memory round trip per
//    fragment.  All residual / bias values of a group of rows are requested before the group's first store.
//  * 16-byte accesses.  A fragment gives a lane 4 features (8 bytes) and its 16-lane rows fq = 0..3 cover 32
//    contiguous bytes of a token row.  One v_permlane16_swap per dword between the packed values of two
//    neighbouring fragments X, Y leaves rows 0 / 2 with features 0-7 / 8-15 of X and rows 1 / 3 with those of Y,
//    so each lane moves 16 contiguous bytes and a wave instruction covers 64-byte runs: half the memory
//    instructions and requests for the same bytes.  The swap is its own inverse, so residual loads use it too.
__device__ __forceinline__ void swap16(bf16x4& x, bf16x4& y) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  union U { bf16x4 v; unsigned int d[2]; } a, b;
  a.v = x;
  b.v = y;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const u32x2 r = __builtin_amdgcn_permlane16_swap(a.d[k], b.d[k], false, false);
    a.d[k] = r[0];
    b.d[k] = r[1];
  }
  x = a.v;
  y = b.v;
}

// Store instructions the wide path issues per wave when no row or column of the wave's tile is cut off.
template <int EPI, int FN, int FM>
constexpr int gemm_epilogue_wide_stores() {
  constexpr int NV = EPI == EPI_SWIGLU ? FN / 2 : FN;
  return FM * (NV / 2 + NV % 2);
}

// Returns true (uniform over the wave) when exactly gemm_epilogue_wide_stores() store instructions were issued, all
// of them after every load of this function: a caller with older asynchronous loads in flight may then wait with
// s_waitcnt vmcnt(that count) for its loads alone and let the stores drain under later work.
template <int EPI, int FN, int FM>
__device__ __forceinline__ bool gemm_epilogue(const GemmParams& p, const f32x4 (&acc)[FN][FM], int m_base, int n_base,
                                              int lane) {
  const int frow = lane & 15, fq = lane >> 4;
  constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_RESIDUAL || EPI == EPI_BIAS_GELU;
  constexpr bool HAS_RES = EPI == EPI_RESIDUAL || EPI == EPI_BIAS_RESIDUAL;
  constexpr bool SWIGLU = EPI == EPI_SWIGLU;
  constexpr int NV = SWIGLU ? FN / 2 : FN;      // packed 4-feature output values per row group
  // Wide path: every column of this wave's tile is inside N and 16-byte accesses are aligned (uniform over the wave)
  const int o_base = SWIGLU ? n_base / 2 : n_base;
  bool wide = EPI != EPI_F32_SCALED && n_base + FN * 16 <= p.N && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 && o_base % 8 == 0;
  if constexpr (HAS_RES) wide = wide && p.ldr % 8 == 0 && ((uintptr_t)p.R & 15) == 0;
  // column of this lane's 16-byte access for the value pair (v, v + 1) / of its 8-byte access for value v
  // (value v covers the 16 output features from o_base + 16 v: with SwiGLU a gate/up fragment pair makes one value)
  auto wide_col = [&](int v) { return o_base + (v + (fq & 1)) * 16 + (fq >> 1) * 8; };
  auto narrow_col = [&](int v) { return o_base + v * 16 + fq * 4; };

  bf16x4 b4[FN];
  if constexpr (HAS_BIAS) {
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int n = n_base + i * 16 + fq * 4;
      b4[i] = (p.bias != nullptr && n < p.N) ? *(const bf16x4*)(p.bias + n) : bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }
  }
  constexpr int JG = HAS_RES ? (FM % 4 == 0 ? 4 : FM % 2 == 0 ? 2 : 1) : 1;   // row groups whose residuals are in flight together
#pragma unroll
  for (int j0 = 0; j0 < FM; j0 += JG) {
    bf16x4 r4[JG][FN];
    if constexpr (HAS_RES) {
#pragma unroll
      for (int jj = 0; jj < JG; ++jj) {
        const int m = min(m_base + (j0 + jj) * 16 + frow, p.M - 1);
        const bf16* rrow = p.R + (long)m * p.ldr;
        if (wide) {
#pragma unroll
          for (int i = 0; i + 1 < FN; i += 2) {
            const bf16x8 w8 = *(const bf16x8*)(rrow + wide_col(i));
            r4[jj][i] = bf16x4{w8[0], w8[1], w8[2], w8[3]};
            r4[jj][i + 1] = bf16x4{w8[4], w8[5], w8[6], w8[7]};
          }
          if constexpr (FN % 2 == 1) r4[jj][FN - 1] = *(const bf16x4*)(rrow + narrow_col(FN - 1));
        } else {
#pragma unroll
          for (int i = 0; i < FN; ++i) r4[jj][i] = *(const bf16x4*)(rrow + min(narrow_col(i), p.N - 4));
        }
      }
      if (wide) {
#pragma unroll
        for (int jj = 0; jj < JG; ++jj)
#pragma unroll
          for (int i = 0; i + 1 < FN; i += 2) swap16(r4[jj][i], r4[jj][i + 1]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < JG; ++jj) {
      const int j = j0 + jj;
      const int m = m_base + j * 16 + frow;
      if constexpr (EPI == EPI_F32_SCALED) {
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          const int n = n_base + i * 16 + fq * 4;
          if (n >= p.N) continue;
          f32x4 o32;
#pragma unroll
          for (int r = 0; r < 4; ++r) o32[r] = acc[i][j][r] * p.alpha;
          *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = o32;
        }
      } else {
        // the packed output values of this 16-row group, then their stores
        bf16x4 o[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (SWIGLU) {
              const float g = round_bf16(acc[2 * v][j][r]);
              const float u = round_bf16(acc[2 * v + 1][j][r]);
              o[v][r] = (bf16)(round_bf16(silu_f32(g)) * u);
            } else if constexpr (EPI == EPI_RESIDUAL) {
              o[v][r] = (bf16)(round_bf16(acc[v][j][r]) + (float)r4[jj][v][r]);
            } else if constexpr (EPI == EPI_BIAS_RESIDUAL) {
              o[v][r] = (bf16)(round_bf16(acc[v][j][r] + (float)b4[v][r]) + (float)r4[jj][v][r]);
            } else if constexpr (EPI == EPI_BIAS_GELU) {
              o[v][r] = (bf16)gelu_tanh_f32(round_bf16(acc[v][j][r] + (float)b4[v][r]));
            } else {
              o[v][r] = (bf16)(acc[v][j][r] + (float)b4[v][r]);
            }
          }
        }
        bf16* crow = p.C + (long)min(m, p.M - 1) * p.ldc;
        if (wide) {   // the swaps need every lane, so rows past M take part and only skip their stores
#pragma unroll
          for (int v = 0; v + 1 < NV; v += 2) {
            swap16(o[v], o[v + 1]);
            if (m < p.M)
              *(bf16x8*)(crow + wide_col(v)) = bf16x8{o[v][0], o[v][1], o[v][2], o[v][3], o[v + 1][0], o[v + 1][1], o[v + 1][2], o[v + 1][3]};
          }
          if constexpr (NV % 2 == 1) {
            if (m < p.M) *(bf16x4*)(crow + narrow_col(NV - 1)) = o[NV - 1];
          }
        } else if (m < p.M) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const int n_in = n_base + (SWIGLU ? 2 * v : v) * 16 + fq * 4;   // weight row that produced the value
            if (n_in < p.N) *(bf16x4*)(crow + narrow_col(v)) = o[v];
          }
        }
      }
    }
  }
  return wide && m_base + FM * 16 <= p.M;
}

// NS: LDS stages.  2: tile kt+1 is requested when tile kt is published and has one K step to land - enough when two
// workgroups share a CU and cover each other.  3: tile kt+2 is requested at that point, for launches that leave a CU
// one workgroup (few tiles, split-K), where a K step otherwise lasts one memory round trip.
template <int BM, int BN, int WM, int WN, int EPI, int AMODE, int NS = 2>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel(GemmParams p) {
  static_assert(NS == 2 || NS == 3, "two or three LDS stages");
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
  constexpr int FM = TM / 16, FN = TN / 16;  // 16x16 fragments per wave
  constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int A_ITERS = BM / 8 / NW, W_ITERS = BN / 8 / NW;  // 1-KiB DMA pieces per wave per stage
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  int tile_m, tile_n;
  gemm_tile_of_block(p, blockIdx.x, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- staging addresses: lane -> (row within 8-row piece, swizzled source chunk)
  const int piece_row = lane >> 3;
  const int src_chunk = (lane & 7) ^ piece_row;
  const bf16* a_src[A_ITERS];
  const bf16* w_src[W_ITERS];
  int a_oy[A_ITERS], a_ox[A_ITERS];
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int row = (it * NW + wave) * 8 + piece_row;
    const int grow = min(m0 + row, p.M - 1);
    if constexpr (AMODE == A_DENSE) {
      a_src[it] = p.A + (long)grow * p.lda + src_chunk * 8;
    } else {
      a_oy[it] = grow / p.Wout;
      a_ox[it] = grow - a_oy[it] * p.Wout;
      a_src[it] = p.A + src_chunk * 8;
    }
  }
#pragma unroll
  for (int it = 0; it < W_ITERS; ++it) {
    const int row = (it * NW + wave) * 8 + piece_row;
    const int grow = min(n0 + row, p.N - 1);
    w_src[it] = p.W + (long)grow * p.ldw + src_chunk * 8;
  }

  // A_CONV3X3: K runs over (tap, channel).  The tap and the per-piece source (bounds test, pixel address) change only
  // every Cin / 64 K steps, and stage() is called with consecutive kt, so they are carried from call to call and
  // re-derived only when the channel offset wraps (for Cin = 512 once in eight K steps).
  int conv_c0 = 0, conv_tap = -1;
  const bf16* conv_src[A_ITERS];
  auto conv_set_tap = [&](int tap) {
    conv_tap = tap;
    const int tap_dy = tap / 3 - 1, tap_dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
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
      // out-of-image taps read the zero page at every channel offset: its pointer is biased so that + c0 stays on it
      conv_src[it] = ok ? a_src[it] + ((long)y * p.Win + x) * p.Cin : nullptr;
    }
  };

  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE_BYTES;
    const int koff = kt * BK;
    if constexpr (AMODE == A_CONV3X3) {
      if (conv_tap < 0) {            // first call: wherever the K range starts
        const int tap = koff / p.Cin;
        conv_c0 = koff - tap * p.Cin;
        conv_set_tap(tap);
      } else {
        conv_c0 += BK;
        if (conv_c0 >= p.Cin) {
          conv_c0 = 0;
          conv_set_tap(conv_tap + 1);
        }
      }
    }
#ifdef BAGEL_DIAG_CONV_A_ONCE   // timing only (wrong results): the A tile is fetched for the first tap alone
    if (AMODE == A_CONV3X3 && conv_tap != 0) goto a_done;
#endif
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) {
      const bf16* src;
      if constexpr (AMODE == A_DENSE) src = a_src[it] + koff;
      else src = conv_src[it] != nullptr ? conv_src[it] + conv_c0 : p.zeros + src_chunk * 8;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
    }
#ifdef BAGEL_DIAG_CONV_A_ONCE
  a_done:;
#endif
#pragma unroll
    for (int it = 0; it < W_ITERS; ++it)
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(w_src[it] + koff),
                                       LDS_PTR(base + A_BYTES + (it * NW + wave) * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets (bytes within a tile image)
  const int frow = lane & 15, fq = lane >> 4;
  int a_off[2], w_off[2];  // per 32-deep k step; rows advance by 16*ROW_BYTES per fragment (row & 7 unchanged)
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int chunk = (kk * 4 + fq) ^ (frow & 7);
    a_off[kk] = (wm * TM + frow) * ROW_BYTES + chunk * 16;
    w_off[kk] = A_BYTES + (wn * TN + frow) * ROW_BYTES + chunk * 16;
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int nk = p.K / BK, k_first = 0;
  if constexpr (EPI == EPI_F32_SCALED && AMODE == A_DENSE) {
    if (p.ksplit > 1) {   // the launcher makes every slice non-empty
      const int per = (nk + p.ksplit - 1) / p.ksplit;
      k_first = blockIdx.y * per;
      nk = min(per, nk - k_first);
      p.C = (bf16*)((float*)p.C + (long)blockIdx.y * p.slab);
    }
  }
  stage(k_first, 0);
  if (NS == 3 && nk > 1) stage(k_first + 1, 1);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt's LDS-DMA must have landed before the barrier publishes it (explicit: the compiler's barrier
    // fence does not reliably wait for LDS-DMA); the barrier also retires tile kt-1's buffer
    if constexpr (NS == 3) {
      // tile kt+1 may still be in flight: wait for all but its pieces.  The bare barrier instruction: __syncthreads()
      // carries a fence, which drains the counter and with it the request that should stay in flight.
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(A_ITERS + W_ITERS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (kt + NS - 1 < nk) stage(k_first + kt + NS - 1, NS == 3 ? (buf == 0 ? 2 : buf - 1) : (kt + 1) & 1);
    const char* base = smem + buf * STAGE_BYTES;
    buf = buf + 1 == NS ? 0 : buf + 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[FM], wf[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) af[j] = *(const bf16x8*)(base + a_off[kk] + j * 16 * ROW_BYTES);
#pragma unroll
      for (int i = 0; i < FN; ++i) wf[i] = *(const bf16x8*)(base + w_off[kk] + i * 16 * ROW_BYTES);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
  }

  gemm_epilogue<EPI, FN, FM>(p, acc, m0 + wm * TM, n0 + wn * TN, lane);
}

