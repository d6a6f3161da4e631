// bf16 MFMA GEMM for the Mixture-of-Transformers projections (gfx950).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )        A, W, C bf16; fp32 accumulate
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out,in]), which is the
// layout the 16x16x32 bf16 MFMA wants for both fragments: lane l holds 8 consecutive k of row (l & 15).
// The weight tile is the MFMA "A" operand and the activation tile the "B" operand, so every lane ends
// up with 4 consecutive output features of one token row (8-byte stores, 8-byte bias/residual loads).
//
// Tiles are staged global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4), two LDS buffers, one
// barrier per 64-deep K step.  LDS rows are 128 B (64 bf16); the 16-byte chunk index is XORed with
// (row & 7) so that the ds_read_b128 fragment reads of 16 different rows at one k offset hit 16
// different bank groups.  The DMA destination is lane-linear, so the XOR is applied to the per-lane
// SOURCE address and again on the read.
//
// Epilogues (the rounding points follow the reference's bf16 autocast graph):
//   EPI_BIAS      C = bf16(acc + bias)                        nn.Linear, qwen2_navit.py:540-547
//   EPI_RESIDUAL  C = bf16(float(bf16(acc)) + R)              residual add, qwen2_navit.py:818,836
//   EPI_SWIGLU    C[:, f] = bf16(bf16(silu(bf16(g))) * bf16(u))  Qwen2MLP, modeling_qwen2.py:200-201
//                 with W rows interleaved [16 gate rows | 16 up rows] per 16 features.
#include "common.h"

enum { EPI_BIAS = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2 };

struct GemmParams {
  const bf16* A; long lda;
  const bf16* W; long ldw;
  bf16* C; long ldc;
  const bf16* bias;
  const bf16* R; long ldr;
  int M, N, K;
  int tiles_m, tiles_n;
};

constexpr int BK = 64;            // bf16 elements per K step (128-byte LDS rows)
constexpr int ROW_BYTES = BK * 2;

template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bf16_nt_kernel(GemmParams p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tile
  constexpr int FM = TM / 16, FN = TN / 16;  // 16x16 fragments per wave
  constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES;
  constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
  constexpr int A_ITERS = BM / 8 / NW, W_ITERS = BN / 8 / NW;  // 1-KiB DMA pieces per wave per stage
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows must split over the waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous range of tiles, and walk tiles in bands of 8 M-tiles so a band's A rows and the
  // W column panels stay in that XCD's L2.
  const int nwg = p.tiles_m * p.tiles_n;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  constexpr int GROUP_M = 8;
  const int band = wg / (GROUP_M * p.tiles_n);
  const int band_first = band * GROUP_M;
  const int band_rows = min(p.tiles_m - band_first, GROUP_M);
  const int in_band = wg - band * GROUP_M * p.tiles_n;
  const int tile_m = band_first + in_band % band_rows;
  const int tile_n = in_band / band_rows;
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
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int row = (it * NW + wave) * 8 + piece_row;
    const int grow = min(m0 + row, p.M - 1);
    a_src[it] = p.A + (long)grow * p.lda + src_chunk * 8;
  }
#pragma unroll
  for (int it = 0; it < W_ITERS; ++it) {
    const int row = (it * NW + wave) * 8 + piece_row;
    const int grow = min(n0 + row, p.N - 1);
    w_src[it] = p.W + (long)grow * p.ldw + src_chunk * 8;
  }

  auto stage = [&](int kt, int buf) {
    char* base = smem + buf * STAGE_BYTES;
    const int koff = kt * BK;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it)
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(a_src[it] + koff), LDS_PTR(base + (it * NW + wave) * 1024), 16, 0, 0);
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

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // waits this wave's DMA (vmcnt 0) then the barrier: tile kt is in LDS, tile kt-1 is consumed
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const char* base = smem + (kt & 1) * STAGE_BYTES;
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

  // ---- epilogue: lane holds features n..n+3 (n = ... + fq*4) of token row m (= ... + frow)
#pragma unroll
  for (int j = 0; j < FM; ++j) {
    const int m = m0 + wm * TM + j * 16 + frow;
    if (m >= p.M) continue;
    if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
      for (int i = 0; i < FN; i += 2) {
        const int n = n0 + wn * TN + i * 16 + fq * 4;  // row of the interleaved weight
        if (n >= p.N) continue;
        const int f = (n0 + wn * TN + i * 16) / 2 + fq * 4;  // output feature
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = round_bf16(acc[i][j][r]);
          const float u = round_bf16(acc[i + 1][j][r]);
          o[r] = (bf16)(round_bf16(silu_f32(g)) * u);
        }
        *(bf16x4*)(p.C + (long)m * p.ldc + f) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int n = n0 + wn * TN + i * 16 + fq * 4;
        if (n >= p.N) continue;
        bf16x4 o;
        if constexpr (EPI == EPI_RESIDUAL) {
          const bf16x4 r4 = *(const bf16x4*)(p.R + (long)m * p.ldr + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16)(round_bf16(acc[i][j][r]) + (float)r4[r]);
        } else {
          if (p.bias != nullptr) {
            const bf16x4 b4 = *(const bf16x4*)(p.bias + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)(acc[i][j][r] + (float)b4[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)acc[i][j][r];
          }
        }
        *(bf16x4*)(p.C + (long)m * p.ldc + n) = o;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(GemmParams p, int epilogue, hipStream_t stream) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n), block(WM * WN * 64);
  const size_t lds = 2 * (BM + BN) * ROW_BYTES;
#define LAUNCH(EPI)                                                                              \
  do {                                                                                           \
    auto kern = gemm_bf16_nt_kernel<BM, BN, WM, WN, EPI>;                                        \
    static bool configured = false;                                                              \
    if (!configured) {                                                                           \
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      configured = true;                                                                         \
    }                                                                                            \
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);                                       \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: LAUNCH(EPI_BIAS); break;
    case EPI_RESIDUAL: LAUNCH(EPI_RESIDUAL); break;
    case EPI_SWIGLU: LAUNCH(EPI_SWIGLU); break;
    default: bagel_set_error("bagel_gemm_bf16: unknown epilogue %d", epilogue); return 1;
  }
#undef LAUNCH
  BAGEL_LAUNCH_CHECK();
  return 0;
}

extern "C" int bagel_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                               int K, const void* bias, const void* residual, long ldr, int epilogue,
                               void* stream) {
  BAGEL_CHECK(A && W && C, "bagel_gemm_bf16: null operand");
  BAGEL_CHECK(M > 0 && N > 0 && K > 0, "bagel_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
  BAGEL_CHECK(K % BK == 0, "bagel_gemm_bf16: K=%d must be a multiple of %d", K, BK);
  BAGEL_CHECK(N % 4 == 0, "bagel_gemm_bf16: N=%d must be a multiple of 4", N);
  BAGEL_CHECK(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "bagel_gemm_bf16: bad leading dims");
  BAGEL_CHECK(ldc % 4 == 0, "bagel_gemm_bf16: ldc must be a multiple of 4");
  BAGEL_CHECK(((uintptr_t)A | (uintptr_t)W) % 16 == 0 && (uintptr_t)C % 8 == 0, "bagel_gemm_bf16: misaligned operand");
  if (epilogue == EPI_RESIDUAL)
    BAGEL_CHECK(residual && ldr % 4 == 0 && (uintptr_t)residual % 8 == 0, "bagel_gemm_bf16: residual epilogue needs R");
  if (epilogue == EPI_SWIGLU) BAGEL_CHECK(N % 32 == 0, "bagel_gemm_bf16: swiglu needs N %% 32 == 0");
  if (bias) BAGEL_CHECK((uintptr_t)bias % 8 == 0, "bagel_gemm_bf16: misaligned bias");
  GemmParams p;
  p.A = (const bf16*)A; p.lda = lda;
  p.W = (const bf16*)W; p.ldw = ldw;
  p.C = (bf16*)C; p.ldc = ldc;
  p.bias = (const bf16*)bias;
  p.R = (const bf16*)residual; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K;
  return launch_gemm<128, 128, 2, 2>(p, epilogue, (hipStream_t)stream);
}
