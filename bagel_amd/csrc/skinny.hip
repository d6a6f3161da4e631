// Weight-streaming GEMM for a handful of token rows (gfx950).
//
//   C[M,N] = epilogue( X[M,K] . W[N,K]^T ),   M <= 64
//
// This is the understanding expert's share of a generation forward: two text tokens
// (<|vision_start|>, <|vision_end|>) per sample go through the und weights
// (qwen2_navit.py:537-547, 604, 826-832) while thousands of latent tokens go through the gen weights.
// The work is HBM-bound: every weight byte is read exactly once.  Each wave owns 16 output features and
// one K slice, loads its weight rows straight into VGPRs (64 contiguous bytes per lane per 128-deep step,
// two full cache lines per weight row), and feeds the 16x16x32 MFMA with the weights as the A operand and
// up to four 16-row groups of X as B operands, so the weights are still read once when M > 16.
// The k order inside an MFMA is permuted identically for both operands (a dot product does not care).
// K slices leave fp32 partial sums in a workspace; a second kernel adds them in slice order and applies
// bias / SiLU / residual / SwiGLU, so results are bitwise reproducible.
#include "common.h"

enum { SK_BIAS = 0, SK_RESIDUAL = 1, SK_SWIGLU = 2, SK_BIAS_SILU = 3 };

template <int G>
__global__ __launch_bounds__(256) void skinny_partial_kernel(const bf16* __restrict__ X, long ldx,
                                                             const bf16* __restrict__ W, long ldw,
                                                             float* __restrict__ part, int M, int N, int K,
                                                             int splitk) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ngroups = N >> 4;
  if (gw >= ngroups * splitk) return;
  const int ng = gw % ngroups, s = gw / ngroups;
  const int ks = K / splitk;
  const int r = lane & 15, h = lane >> 4;

  const bf16* wp = W + (long)(ng * 16 + r) * ldw + s * ks + h * 32;
  const bf16* xp[G];
  bool xok[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int m = g * 16 + r;
    xok[g] = m < M;
    xp[g] = X + (long)min(m, M - 1) * ldx + s * ks + h * 32;
  }
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
  for (int k = 0; k < ks; k += 128) {
    bf16x8 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = __builtin_nontemporal_load((const bf16x8*)(wp + k) + j);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bf16x8 xf[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xf[j] = *((const bf16x8*)(xp[g] + k) + j);
        if (!xok[g]) xf[j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[j], acc[g], 0, 0, 0);
    }
  }
  // lane holds features n..n+3 (n = ng*16 + h*4) of row m = g*16 + r
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int m = g * 16 + r;
    if (m < M) *(f32x4*)(part + ((long)s * M + m) * N + ng * 16 + h * 4) = acc[g];
  }
}

// Sum the K-slice partials in slice order and apply the epilogue.  One thread per output element.
__global__ void skinny_reduce_kernel(const float* __restrict__ part, int splitk, int M, int N,
                                     const bf16* __restrict__ bias, const bf16* __restrict__ R, long ldr,
                                     bf16* __restrict__ C, long ldc, int epilogue) {
  const int ncols = epilogue == SK_SWIGLU ? N / 2 : N;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * ncols) return;
  const int m = idx / ncols, c = idx % ncols;
  auto sum = [&](int n) {
    float a = 0.f;
    for (int s = 0; s < splitk; ++s) a += part[((long)s * M + m) * N + n];
    return a;
  };
  float out;
  if (epilogue == SK_SWIGLU) {
    const int ng = (c >> 4) * 32 + (c & 15);
    const float g = round_bf16(sum(ng)), u = round_bf16(sum(ng + 16));
    out = round_bf16(silu_f32(g)) * u;
  } else {
    float a = sum(c);
    if (bias) a += (float)bias[c];
    if (epilogue == SK_RESIDUAL) a = round_bf16(a) + (float)R[(long)m * ldr + c];
    if (epilogue == SK_BIAS_SILU) a = silu_f32(round_bf16(a));
    out = a;
  }
  C[(long)m * ldc + c] = (bf16)out;
}

extern "C" long bagel_skinny_gemm_workspace_bytes(int M, int N) { return (long)16 * M * N * sizeof(float); }

extern "C" int bagel_skinny_gemm_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M,
                                      int N, int K, const void* bias, const void* residual, long ldr,
                                      int epilogue, void* workspace, long workspace_bytes, void* stream) {
  BAGEL_CHECK(X && W && C && workspace, "bagel_skinny_gemm_bf16: null operand");
  BAGEL_CHECK(M > 0 && M <= 64, "bagel_skinny_gemm_bf16: M=%d outside 1..64", M);
  BAGEL_CHECK(N > 0 && N % 16 == 0, "bagel_skinny_gemm_bf16: N=%d must be a multiple of 16", N);
  BAGEL_CHECK(K > 0 && K % 128 == 0, "bagel_skinny_gemm_bf16: K=%d must be a multiple of 128", K);
  BAGEL_CHECK(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, "bagel_skinny_gemm_bf16: bad leading dims");
  BAGEL_CHECK(((uintptr_t)X | (uintptr_t)W | (uintptr_t)workspace) % 16 == 0, "bagel_skinny_gemm_bf16: misaligned operand");
  BAGEL_CHECK(epilogue >= 0 && epilogue <= 3, "bagel_skinny_gemm_bf16: unknown epilogue %d", epilogue);
  if (epilogue == SK_RESIDUAL) BAGEL_CHECK(residual, "bagel_skinny_gemm_bf16: residual epilogue needs R");
  if (epilogue == SK_SWIGLU) BAGEL_CHECK(N % 32 == 0, "bagel_skinny_gemm_bf16: swiglu needs N %% 32 == 0");
  // K slices: a divisor of K/128, at most 16, keeping the grid near 4096 waves
  const int ksteps = K / 128, ngroups = N / 16;
  int splitk = 1;
  for (int d = 1; d <= 16; ++d)
    if (ksteps % d == 0 && (long)ngroups * d <= 4096) splitk = d;
  BAGEL_CHECK((long)splitk * M * N * (long)sizeof(float) <= workspace_bytes,
              "bagel_skinny_gemm_bf16: workspace too small (%ld bytes given)", workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int waves = ngroups * splitk;
  const dim3 grid((waves + 3) / 4), block(256);
  float* part = (float*)workspace;
  const int G = (M + 15) / 16;
  switch (G) {
    case 1: hipLaunchKernelGGL(skinny_partial_kernel<1>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk); break;
    case 2: hipLaunchKernelGGL(skinny_partial_kernel<2>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk); break;
    case 3: hipLaunchKernelGGL(skinny_partial_kernel<3>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk); break;
    default: hipLaunchKernelGGL(skinny_partial_kernel<4>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk); break;
  }
  BAGEL_LAUNCH_CHECK();
  const int ncols = epilogue == SK_SWIGLU ? N / 2 : N;
  const long total = (long)M * ncols;
  hipLaunchKernelGGL(skinny_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, splitk, M, N,
                     (const bf16*)bias, (const bf16*)residual, ldr, (bf16*)C, ldc, epilogue);
  BAGEL_LAUNCH_CHECK();
  return 0;
}
