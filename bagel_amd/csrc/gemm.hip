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
//   EPI_BIAS_RESIDUAL  C = bf16(bf16(acc + bias) + R)         x + conv(x) in the VAE decoder
//   EPI_F32_SCALED     C (fp32) = alpha * acc                 attention scores of the VAE mid block
//   EPI_BIAS_GELU      C = bf16(gelu_tanh(bf16(acc + bias)))    SiglipMLP.fc1 / MLPconnector.fc1 + activation
#include "gemm_tile.h"
#include "gemm_pipelined.h"
#include <algorithm>

#ifdef BAGEL_DIAG_CLOCK
static unsigned long long* g_gemm_diag = nullptr;   // diagnostic builds: [2 x workgroups] stamps of the next launches
extern "C" void bagel_set_gemm_diag(void* buf) { g_gemm_diag = (unsigned long long*)buf; }
#endif
static int g_gemm_variant = -1;  // -1: choose by problem size
static int g_gemm_persistent = 1;  // 0: one workgroup per output tile

static int g_gemm_stages = 0;   // development knob: 0 automatic, 2 / 3 force the LDS stages of the 128 x 128 kernel
extern "C" void bagel_set_gemm_stages(int n) { g_gemm_stages = n == 2 || n == 3 ? n : 0; }

// CUs of the current device, rounded down to a multiple of the 8 XCDs (256 on MI355X)
static int compute_units() {
  const int cu = bagel_cu_count();
  return cu >= 8 ? cu / 8 * 8 : 256;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(GemmParams p, int epilogue, hipStream_t stream) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const dim3 grid(p.tiles_m * p.tiles_n), block(WM * WN * 64);
  // at most one workgroup per CU: three LDS stages, so that a tile has two K steps to land
  const bool deep = g_gemm_stages > 0 ? g_gemm_stages == 3 : p.tiles_m * p.tiles_n <= compute_units();
  const size_t lds = (deep ? 3 : 2) * (BM + BN) * ROW_BYTES;
#define LAUNCH(EPI)                                                                              \
  do {                                                                                           \
    auto k2 = gemm_bf16_nt_kernel<BM, BN, WM, WN, EPI, A_DENSE, 2>;                              \
    auto k3 = gemm_bf16_nt_kernel<BM, BN, WM, WN, EPI, A_DENSE, 3>;                              \
    static BagelLdsAllowed allowed2, allowed3;                                                   \
    bagel_allow_lds(allowed2, (const void*)k2, 2 * (BM + BN) * ROW_BYTES);                       \
    bagel_allow_lds(allowed3, (const void*)k3, 3 * (BM + BN) * ROW_BYTES);                       \
    hipLaunchKernelGGL(deep ? k3 : k2, grid, block, lds, stream, p);                             \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: LAUNCH(EPI_BIAS); break;
    case EPI_RESIDUAL: LAUNCH(EPI_RESIDUAL); break;
    case EPI_SWIGLU: LAUNCH(EPI_SWIGLU); break;
    case EPI_BIAS_RESIDUAL: LAUNCH(EPI_BIAS_RESIDUAL); break;
    case EPI_F32_SCALED: LAUNCH(EPI_F32_SCALED); break;
    case EPI_BIAS_GELU: LAUNCH(EPI_BIAS_GELU); break;
    default: bagel_set_error("bagel_gemm_bf16: unknown epilogue %d", epilogue); return 1;
  }
#undef LAUNCH
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Split-K for row counts between the weight-streaming kernels and a chip-filling tile count (17 ... 1024 rows of a
// projection with few column tiles: a text prompt's prefill).  One 128 x 128 tile per 128 output features is 28-36
// workgroups walking the whole reduction one after the other; K slices put SPLIT times as many on the chip, their fp32
// partial tiles go through a workspace, and a second launch adds them in slice order and applies the epilogue.
// The split depends on (N, K) only, so a row's result does not depend on how many rows share the launch.
// ---------------------------------------------------------------------------------------------
constexpr int SPLITK_MAX_ROWS = 1024;

// Row counts fall into two classes, (16, 512] and (512, 1024]; inside a class the split is the same for every M.
// Above 512 rows only long reductions are split (slices stay at least 32 K steps deep: the down projection), because
// the partial tiles then cost as much traffic as the weights; above 1024 rows the split measured slower.
static int gemm_ksplit(int M, int N, int K, int epilogue) {
  if (M <= 16 || M > SPLITK_MAX_ROWS) return 1;
  if (epilogue != EPI_BIAS && epilogue != EPI_RESIDUAL && epilogue != EPI_BIAS_RESIDUAL && epilogue != EPI_BIAS_GELU) return 1;
  const int tiles_n = (N + 127) / 128, nk = K / BK;
  const int row_class = M <= 512 ? 1 : 2;
  int ks = std::min(256 / tiles_n / row_class, row_class == 1 ? nk / 4 : nk / 32);
  if (ks < 2) return 1;
  const int per = (nk + ks - 1) / ks;
  return (nk + per - 1) / per;   // every slice non-empty
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int ksplit, long slab, int M, int N,
                                                                 const bf16* __restrict__ bias, const bf16* __restrict__ R, long ldr,
                                                                 bf16* __restrict__ C, long ldc, int epilogue) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;   // one thread: four features of one row
  const int nq = N / 4;
  if (q >= (long)M * nq) return;
  const int m = (int)(q / nq), n = (int)(q - (long)m * nq) * 4;
  const float* src = part + (long)m * N + n;
  f32x4 a = *(const f32x4*)src;
  for (int s = 1; s < ksplit; ++s) a += *(const f32x4*)(src + s * slab);
  bf16x4 b4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, r4 = b4, o;
  if (bias != nullptr && epilogue != EPI_RESIDUAL) b4 = *(const bf16x4*)(bias + n);
  if (epilogue == EPI_RESIDUAL || epilogue == EPI_BIAS_RESIDUAL) r4 = *(const bf16x4*)(R + (long)m * ldr + n);
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // the arithmetic of gemm_epilogue()
    if (epilogue == EPI_RESIDUAL) o[i] = (bf16)(round_bf16(a[i]) + (float)r4[i]);
    else if (epilogue == EPI_BIAS_RESIDUAL) o[i] = (bf16)(round_bf16(a[i] + (float)b4[i]) + (float)r4[i]);
    else if (epilogue == EPI_BIAS_GELU) o[i] = (bf16)gelu_tanh_f32(round_bf16(a[i] + (float)b4[i]));
    else o[i] = (bf16)(a[i] + (float)b4[i]);
  }
  *(bf16x4*)(C + (long)m * ldc + n) = o;
}

bool bagel_launch_reduce_residual_rmsnorm(const float* part, int slices, int M, int N, const bf16* R, long ldr, bf16* C, long ldc,
                                          const bf16* norm_w, bf16* Xn, long ldn, float eps, hipStream_t st);   // skinny.hip

// norm_w != nullptr (residual epilogue only): the reduce launch also applies the RMSNorm that follows, Xn = norm(C)
static int launch_gemm_splitk(GemmParams p, int epilogue, int ksplit, float* workspace, hipStream_t stream,
                              const bf16* norm_w = nullptr, bf16* Xn = nullptr, long ldn = 0, float eps = 0.f) {
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 127) / 128;
  GemmParams q = p;
  q.C = (bf16*)workspace;
  q.ldc = p.N;
  q.alpha = 1.0f;
  q.ksplit = ksplit;
  q.slab = (long)p.M * p.N;
  q.bias = nullptr;
  q.R = nullptr;
  const dim3 grid(p.tiles_m * p.tiles_n, ksplit), block(256);
  // at most one workgroup per CU: three LDS stages (same rule as launch_gemm)
  const bool deep = g_gemm_stages > 0 ? g_gemm_stages == 3 : (long)p.tiles_m * p.tiles_n * ksplit <= compute_units();
  const size_t lds = (deep ? 3 : 2) * (128 + 128) * ROW_BYTES;
  auto k2 = gemm_bf16_nt_kernel<128, 128, 2, 2, EPI_F32_SCALED, A_DENSE, 2>;
  auto k3 = gemm_bf16_nt_kernel<128, 128, 2, 2, EPI_F32_SCALED, A_DENSE, 3>;
  static BagelLdsAllowed allowed2, allowed3;
  bagel_allow_lds(allowed2, (const void*)k2, 2 * (128 + 128) * ROW_BYTES);
  bagel_allow_lds(allowed3, (const void*)k3, 3 * (128 + 128) * ROW_BYTES);
  hipLaunchKernelGGL(deep ? k3 : k2, grid, block, lds, stream, q);
  BAGEL_LAUNCH_CHECK();
  if (norm_w != nullptr) {
    const bool ok = bagel_launch_reduce_residual_rmsnorm(workspace, ksplit, p.M, p.N, p.R, p.ldr, p.C, p.ldc, norm_w, Xn, ldn, eps, stream);
    BAGEL_CHECK(ok, "bagel_gemm_ws_residual_rmsnorm_bf16: shape outside the fused reduce's limits");
    BAGEL_LAUNCH_CHECK();
    return 0;
  }
  const long quads = (long)p.M * (p.N / 4);
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, workspace, ksplit, q.slab,
                     p.M, p.N, p.bias, p.R, p.ldr, p.C, p.ldc, epilogue);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

static int g_gemm_lead = -1;  // development knob: -1 automatic, 0 / 1 force the DMA split (bagel_set_gemm_lead)
template <int BN, int WR>
static int launch_pipelined(GemmParams p, int epilogue, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + BN - 1) / BN;
  const dim3 grid(g_gemm_persistent ? std::min(p.tiles_m * p.tiles_n, compute_units()) : p.tiles_m * p.tiles_n), block(512);
  const size_t lds = 2 * (256 + BN) * ROW_BYTES;
  constexpr bool PAIRS = (BN / (8 / WR) / 16) % 2 == 0;  // SwiGLU needs an even number of fragments per wave
  // Long reductions (the down projection, K = 18944) run 5 % faster with all DMA pieces on the four leading
  // waves; on K = 3584 it is neutral to slightly slower, so those keep the even split.
  const bool lead = g_gemm_lead >= 0 ? g_gemm_lead != 0 : p.K >= 8192;
#define LAUNCH(EPI)                                                                              \
  do {                                                                                           \
    auto k0 = gemm256_pipelined_kernel<EPI, BN, WR, false>;                                    \
    auto k1 = gemm256_pipelined_kernel<EPI, BN, WR, true>;                                     \
    static BagelLdsAllowed allowed0, allowed1;                                                   \
    bagel_allow_lds(allowed0, (const void*)k0, (int)lds);                                        \
    bagel_allow_lds(allowed1, (const void*)k1, (int)lds);                                        \
    hipLaunchKernelGGL(lead ? k1 : k0, grid, block, lds, stream, p);                             \
  } while (0)
  switch (epilogue) {
    case EPI_BIAS: LAUNCH(EPI_BIAS); break;
    case EPI_RESIDUAL: LAUNCH(EPI_RESIDUAL); break;
    case EPI_SWIGLU:
      if constexpr (PAIRS) {
        LAUNCH(EPI_SWIGLU);
        break;
      } else {
        bagel_set_error("bagel_gemm_bf16: this tile shape has no SwiGLU epilogue");
        return 1;
      }
    case EPI_BIAS_RESIDUAL: LAUNCH(EPI_BIAS_RESIDUAL); break;
    case EPI_F32_SCALED: LAUNCH(EPI_F32_SCALED); break;
    case EPI_BIAS_GELU: LAUNCH(EPI_BIAS_GELU); break;
    default: bagel_set_error("bagel_gemm_bf16: unknown epilogue %d", epilogue); return 1;
  }
#undef LAUNCH
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// Tile shape for a problem: the cheapest of the candidates, where a shape costs (rounds of workgroups over
// the 256 CUs) x (time of one round, in units of 256-wide pipelined tile columns).  The 128x128 kernel
// runs two workgroups per CU at about 0.77 of the pipelined kernel's rate.
static int choose_gemm_variant(int M, int N, int epilogue) {
  const long tm = (M + 255) / 256;
  auto rounds = [](long tiles, long per_round) { return (tiles + per_round - 1) / per_round; };
  double best = (double)rounds((long)((M + 127) / 128) * ((N + 127) / 128), 512) * 166.0;
  int variant = 0;
  const int bn[3] = {256, 224, 288}, id[3] = {3, 4, 5};
  for (int c = 0; c < 3; ++c) {
    if (c > 0 && epilogue == EPI_SWIGLU) continue;
    // the 288-wide tile spills a few registers: measured about 6 % slower per output element
    const double cost = (double)rounds(tm * ((N + bn[c] - 1) / bn[c]), 256) * bn[c] * (bn[c] == 288 ? 1.06 : 1.0);
    if (cost < best) {
      best = cost;
      variant = id[c];
    }
  }
  // SwiGLU (gate+up, the widest N): the 4 x 2 wave layout measured 2-4 % faster than 2 x 4 on the same tile
  if (variant == 3 && epilogue == EPI_SWIGLU) variant = 6;
  return variant;
}

static int gemm_entry(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                      int K, const void* bias, const void* residual, long ldr, int epilogue,
                      float alpha, void* workspace, long workspace_bytes, void* stream) {
  BAGEL_CHECK(A && W && C, "bagel_gemm_bf16: null operand");
  BAGEL_CHECK(M > 0 && N > 0 && K > 0, "bagel_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
  BAGEL_CHECK(K % BK == 0, "bagel_gemm_bf16: K=%d must be a multiple of %d", K, BK);
  BAGEL_CHECK(N % 4 == 0, "bagel_gemm_bf16: N=%d must be a multiple of 4", N);
  BAGEL_CHECK(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "bagel_gemm_bf16: bad leading dims");
  BAGEL_CHECK(lda < (1L << 21) && ldw < (1L << 21), "bagel_gemm_bf16: leading dims beyond the 31-bit tile offsets");
  BAGEL_CHECK(ldc % 4 == 0, "bagel_gemm_bf16: ldc must be a multiple of 4");
  BAGEL_CHECK(((uintptr_t)A | (uintptr_t)W) % 16 == 0 && (uintptr_t)C % 8 == 0, "bagel_gemm_bf16: misaligned operand");
  if (epilogue == EPI_RESIDUAL || epilogue == EPI_BIAS_RESIDUAL)
    BAGEL_CHECK(residual && ldr % 4 == 0 && (uintptr_t)residual % 8 == 0, "bagel_gemm_bf16: residual epilogue needs R");
  if (epilogue == EPI_BIAS_RESIDUAL || epilogue == EPI_BIAS_GELU) BAGEL_CHECK(bias, "bagel_gemm_bf16: this epilogue needs bias");
  if (epilogue == EPI_F32_SCALED) BAGEL_CHECK((uintptr_t)C % 16 == 0, "bagel_gemm_bf16: fp32 output must be 16-byte aligned");
  if (epilogue == EPI_SWIGLU) BAGEL_CHECK(N % 32 == 0, "bagel_gemm_bf16: swiglu needs N %% 32 == 0");
  if (bias) BAGEL_CHECK((uintptr_t)bias % 8 == 0, "bagel_gemm_bf16: misaligned bias");
  GemmParams p;
  p.A = (const bf16*)A; p.lda = lda;
  p.W = (const bf16*)W; p.ldw = ldw;
  p.C = (bf16*)C; p.ldc = ldc;
  p.bias = (const bf16*)bias;
  p.R = (const bf16*)residual; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha;
  p.diag = nullptr;
  p.ksplit = 1;
  p.slab = 0;
#ifdef BAGEL_DIAG_CLOCK
  p.diag = g_gemm_diag;
#endif
  if (workspace != nullptr && g_gemm_variant < 0) {
    const int ksplit = gemm_ksplit(M, N, K, epilogue);
    if (ksplit > 1 && (uintptr_t)workspace % 16 == 0 && (long)ksplit * M * N * (long)sizeof(float) <= workspace_bytes &&
        N % 4 == 0 && ldc % 4 == 0 && (uintptr_t)C % 8 == 0 && (bias == nullptr || (uintptr_t)bias % 8 == 0))
      return launch_gemm_splitk(p, epilogue, ksplit, (float*)workspace, (hipStream_t)stream);
  }
  // 256-row software-pipelined tiles (8 waves, one workgroup per CU) or 128x128 tiles (4 waves, two
  // workgroups per CU).  Measured on the 7B gate+up shape (random operands): 128x128 1.10, software-pipelined
  // 1.38 PFLOP/s (a 256x256 one-barrier loop reached 1.25 and a ping-pong schedule 1.29; neither is kept).
  const int variant = g_gemm_variant >= 0 ? g_gemm_variant : choose_gemm_variant(M, N, epilogue);
  switch (variant) {
    case 3: return launch_pipelined<256, 2>(p, epilogue, (hipStream_t)stream);
    case 4: return launch_pipelined<224, 4>(p, epilogue, (hipStream_t)stream);
    case 5: return launch_pipelined<288, 4>(p, epilogue, (hipStream_t)stream);
    case 6: return launch_pipelined<256, 4>(p, epilogue, (hipStream_t)stream);
    case 0: return launch_gemm<128, 128, 2, 2>(p, epilogue, (hipStream_t)stream);
    case 8: return launch_gemm<64, 128, 2, 2>(p, epilogue, (hipStream_t)stream);   // never chosen automatically (tools/midm_bench.py)
    default: bagel_set_error("bagel_gemm_bf16: unknown tile variant %d", variant); return 1;
  }
}

extern "C" int bagel_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                               int K, const void* bias, const void* residual, long ldr, int epilogue,
                               float alpha, void* stream) {
  return gemm_entry(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, alpha, nullptr, 0, stream);
}

// The same product with a workspace: 17 ... 1024 rows of a projection with few column tiles take the split-K form when
// the workspace holds bagel_gemm_workspace_bytes(M, N, K, epilogue); anything else runs as bagel_gemm_bf16 does.
extern "C" int bagel_gemm_ws_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                                  int K, const void* bias, const void* residual, long ldr, int epilogue,
                                  float alpha, void* workspace, long workspace_bytes, void* stream) {
  return gemm_entry(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, alpha, workspace, workspace_bytes, stream);
}

extern "C" int bagel_rmsnorm_bf16(const void* x, long ldx, void* out, long ldo, int rows, int H, const void* w_first,
                                  const void* w_second, int split_row, float eps, void* stream);

// C = bf16(bf16(A . W^T) + R) and Xn = RMSNorm(C) * norm_weight.  Where bagel_gemm_ws_bf16 would split the reduction
// (and N <= 8192), the launch that adds the partial tiles applies the residual and the norm as well; otherwise this is
// bagel_gemm_ws_bf16 followed by bagel_rmsnorm_bf16.
extern "C" int bagel_gemm_ws_residual_rmsnorm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M,
                                                   int N, int K, const void* residual, long ldr, const void* norm_weight,
                                                   void* Xn, long ldn, float eps, void* workspace, long workspace_bytes,
                                                   void* stream) {
  BAGEL_CHECK(A && W && C && residual && norm_weight && Xn, "bagel_gemm_ws_residual_rmsnorm_bf16: null operand");
  BAGEL_CHECK(M > 0 && N > 0 && K > 0 && K % BK == 0 && N % 4 == 0, "bagel_gemm_ws_residual_rmsnorm_bf16: bad shape M=%d N=%d K=%d", M, N, K);
  const int ksplit = g_gemm_variant < 0 ? gemm_ksplit(M, N, K, EPI_RESIDUAL) : 1;
  const bool fused = ksplit > 1 && ksplit <= 16 && N <= 8192 && workspace != nullptr && (uintptr_t)workspace % 16 == 0 &&
                     (long)ksplit * M * N * (long)sizeof(float) <= workspace_bytes && lda >= K && ldw >= K && lda % 8 == 0 &&
                     ldw % 8 == 0 && ((uintptr_t)A | (uintptr_t)W) % 16 == 0;
  if (!fused) {
    const int rc = gemm_entry(A, lda, W, ldw, C, ldc, M, N, K, nullptr, residual, ldr, EPI_RESIDUAL, 1.0f, workspace, workspace_bytes, stream);
    if (rc != 0) return rc;
    return bagel_rmsnorm_bf16(C, ldc, Xn, ldn, M, N, norm_weight, norm_weight, M, eps, stream);
  }
  GemmParams p;
  p.A = (const bf16*)A; p.lda = lda;
  p.W = (const bf16*)W; p.ldw = ldw;
  p.C = (bf16*)C; p.ldc = ldc;
  p.bias = nullptr;
  p.R = (const bf16*)residual; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.alpha = 1.0f;
  p.diag = nullptr;
  p.ksplit = 1;
  p.slab = 0;
  return launch_gemm_splitk(p, EPI_RESIDUAL, ksplit, (float*)workspace, (hipStream_t)stream, (const bf16*)norm_weight, (bf16*)Xn, ldn, eps);
}

extern "C" long bagel_gemm_workspace_bytes(int M, int N, int K, int epilogue) {
  if (M <= 0 || N <= 0 || K <= 0 || K % BK != 0) return 0;
  const int ksplit = gemm_ksplit(M, N, K, epilogue);
  return ksplit > 1 ? (long)ksplit * M * N * (long)sizeof(float) : 0;
}

// Development knob: forces the tile variant bagel_gemm_bf16 launches (tools/gemm_bench.py); -1 = automatic.
extern "C" void bagel_set_gemm_variant(int v) { g_gemm_variant = v; }

// Development knob: 0 launches the pipelined tiles with one workgroup per output tile instead of one
// persistent workgroup per CU (A/B timing in tools/step_ab.py).
extern "C" void bagel_set_gemm_persistent(int on) { g_gemm_persistent = on; }

// Development knob: forces the DMA split of the pipelined tiles (1 = leading waves stage everything).
extern "C" void bagel_set_gemm_lead(int v) { g_gemm_lead = v; }
