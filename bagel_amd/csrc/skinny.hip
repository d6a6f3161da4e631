// Weight-streaming GEMM for a handful of token rows (gfx950).
//
//   C[M,N] = epilogue( X[M,K] . W[N,K]^T ),   M <= 64
//
// This is the understanding expert's share of a generation forward: two text tokens
// (<|vision_start|>, <|vision_end|>) per sample go through the und weights
// (qwen2_navit.py:537-547, 604, 826-832) while thousands of latent tokens go through the gen weights.
// The work is HBM-bound: every weight byte is read exactly once.  Each wave owns 16 output features and
// one K slice, loads its weight rows straight into VGPRs (two register stages, so a 4 KB step is always in
// flight behind the one being multiplied), and feeds the 16x16x32 MFMA with the weights as the A operand
// and up to four 16-row groups of X as B operands, so the weights are still read once when M > 16.
// K slices leave fp32 partial sums in a workspace; a second kernel adds them in slice order and applies
// bias / SiLU / residual / SwiGLU, so results are bitwise reproducible.
#include "common.h"
#include <algorithm>

enum { SK_BIAS = 0, SK_RESIDUAL = 1, SK_SWIGLU = 2, SK_BIAS_SILU = 3 };

template <int G>
__global__ __launch_bounds__(256) void skinny_partial_kernel(const bf16* __restrict__ X, long ldx,
                                                             const bf16* __restrict__ W, long ldw,
                                                             float* __restrict__ part, int M, int N, int K,
                                                             int splitk, int steps_per_slice) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int ngroups = N >> 4;
  if (gw >= ngroups * splitk) return;
  const int ng = gw % ngroups, s = gw / ngroups;
  const int k0 = s * steps_per_slice * 128;
  const int nsteps = min(steps_per_slice, K / 128 - s * steps_per_slice);   // the last slice may be shorter
  const int r = lane & 15, h = lane >> 4;

  // MFMA j of a 128-deep step covers k = 32 j .. 32 j + 31 of the step; lane (r, h) supplies k = 32 j + 8 h .. + 7,
  // so one load instruction reads 64 contiguous bytes of each of the wave's 16 weight rows.
  const bf16* wp = W + (long)(ng * 16 + r) * ldw + k0 + h * 8;
  const bf16* xp[G];
  bool xok[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int m = g * 16 + r;
    xok[g] = m < M;
    xp[g] = X + (long)min(m, M - 1) * ldx + k0 + h * 8;
  }
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};

  // two register stages: the loads of step i + 1 are in flight while step i is multiplied
  bf16x8 wf[2][4], xf[2][G][4];
  auto load = [&](int st, int step) {
#pragma unroll
    // plain loads: the non-temporal policy measured 9-12 % slower on this stream (4.9 vs 5.5 TB/s on lm_head)
    for (int j = 0; j < 4; ++j) wf[st][j] = *(const bf16x8*)(wp + step * 128 + j * 32);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[st][g][j] = *(const bf16x8*)(xp[g] + step * 128 + j * 32);
  };
  auto compute = [&](int st) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[st][j], xf[st][g][j], acc[g], 0, 0, 0);
  };
  load(0, 0);
  for (int step = 0; step < nsteps; step += 2) {
    if (step + 1 < nsteps) load(1, step + 1);
    compute(0);
    if (step + 2 < nsteps) load(0, step + 2);
    if (step + 1 < nsteps) compute(1);
  }
  // lane holds features n..n+3 (n = ng*16 + h*4) of row m = g*16 + r; rows past M carried clamped copies
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int m = g * 16 + r;
    if (xok[g]) *(f32x4*)(part + ((long)s * M + m) * N + ng * 16 + h * 4) = acc[g];
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

// Reduce + residual + RMSNorm in one launch for the decode loop, where rows are few and every launch is
// latency: C = bf16(bf16(sum of partials) + R) as SK_RESIDUAL does, then Xn = w * bf16(C * rsqrt(mean(C^2) + eps))
// (Qwen2RMSNorm on the stored bf16 row, modeling_qwen2.py:54-59).  One workgroup per row.
constexpr int RN_THREADS = 1024, RN_MAXC = 8;  // columns per thread: N <= 8192
__global__ __launch_bounds__(RN_THREADS) void skinny_reduce_rmsnorm_kernel(const float* __restrict__ part, int splitk,
                                                                           int M, int N, const bf16* __restrict__ R,
                                                                           long ldr, bf16* __restrict__ C, long ldc,
                                                                           const bf16* __restrict__ w,
                                                                           bf16* __restrict__ Xn, long ldn, float eps) {
  const int m = blockIdx.x;
  float v[RN_MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < RN_MAXC; ++i) {
    const int c = threadIdx.x + RN_THREADS * i;
    if (c < N) {
      // all slices' loads are issued before the first add (the row is summed by one workgroup: latency-bound);
      // the sum runs in slice order like skinny_reduce_kernel's
      float p[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) p[s] = s < splitk ? part[((long)s * M + m) * N + c] : 0.f;
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < 16; ++s) a += p[s];
      const float o = round_bf16(round_bf16(a) + (float)R[(long)m * ldr + c]);
      C[(long)m * ldc + c] = (bf16)o;
      v[i] = o;
      ss += o * o;
    }
  }
  __shared__ float red[RN_THREADS / 64];
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < RN_THREADS / 64; ++i) tot += red[i];
  const float rs = rsqrtf(tot / (float)N + eps);
#pragma unroll
  for (int i = 0; i < RN_MAXC; ++i) {
    const int c = threadIdx.x + RN_THREADS * i;
    if (c < N) Xn[(long)m * ldn + c] = (bf16)((float)w[c] * round_bf16(v[i] * rs));
  }
}

extern "C" long bagel_skinny_gemm_workspace_bytes(int M, int N) { return (long)16 * M * N * sizeof(float); }

static int skinny_launch(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                         const void* bias, const void* residual, long ldr, int epilogue, void* workspace,
                         long workspace_bytes, const void* norm_weight, void* Xn, long ldn, float eps, void* stream) {
  BAGEL_CHECK(X && W && C && workspace, "bagel_skinny_gemm_bf16: null operand");
  BAGEL_CHECK(M > 0 && M <= 64, "bagel_skinny_gemm_bf16: M=%d outside 1..64", M);
  BAGEL_CHECK(N > 0 && N % 16 == 0, "bagel_skinny_gemm_bf16: N=%d must be a multiple of 16", N);
  BAGEL_CHECK(K > 0 && K % 128 == 0, "bagel_skinny_gemm_bf16: K=%d must be a multiple of 128", K);
  BAGEL_CHECK(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K, "bagel_skinny_gemm_bf16: bad leading dims");
  BAGEL_CHECK(((uintptr_t)X | (uintptr_t)W | (uintptr_t)workspace) % 16 == 0, "bagel_skinny_gemm_bf16: misaligned operand");
  BAGEL_CHECK(epilogue >= 0 && epilogue <= 3, "bagel_skinny_gemm_bf16: unknown epilogue %d", epilogue);
  if (epilogue == SK_RESIDUAL) BAGEL_CHECK(residual, "bagel_skinny_gemm_bf16: residual epilogue needs R");
  if (epilogue == SK_SWIGLU) BAGEL_CHECK(N % 32 == 0, "bagel_skinny_gemm_bf16: swiglu needs N %% 32 == 0");
  if (norm_weight) BAGEL_CHECK(Xn && epilogue == SK_RESIDUAL && N <= RN_THREADS * RN_MAXC, "bagel_skinny_gemm_residual_rmsnorm_bf16: bad arguments");
  // K slices (at most 16, at least two 128-deep steps each) so that the grid has about 24 waves per CU: the
  // kernel is latency-bound per wave, and HBM only saturates with tens of KB in flight per CU
  const int ksteps = K / 128, ngroups = N / 16;
  int splitk = (int)std::min<long>(16, std::max<long>(1, (6144 + ngroups - 1) / ngroups));
  splitk = std::min(splitk, std::max(1, ksteps / 2));
  const int steps_per_slice = (ksteps + splitk - 1) / splitk;
  splitk = (ksteps + steps_per_slice - 1) / steps_per_slice;
  BAGEL_CHECK((long)splitk * M * N * (long)sizeof(float) <= workspace_bytes,
              "bagel_skinny_gemm_bf16: workspace too small (%ld bytes given)", workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  const int waves = ngroups * splitk;
  const dim3 grid((waves + 3) / 4), block(256);
  float* part = (float*)workspace;
  const int G = (M + 15) / 16;
  switch (G) {
    case 1: hipLaunchKernelGGL(skinny_partial_kernel<1>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk, steps_per_slice); break;
    case 2: hipLaunchKernelGGL(skinny_partial_kernel<2>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk, steps_per_slice); break;
    case 3: hipLaunchKernelGGL(skinny_partial_kernel<3>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk, steps_per_slice); break;
    default: hipLaunchKernelGGL(skinny_partial_kernel<4>, grid, block, 0, st, (const bf16*)X, ldx, (const bf16*)W, ldw, part, M, N, K, splitk, steps_per_slice); break;
  }
  BAGEL_LAUNCH_CHECK();
  if (norm_weight) {
    hipLaunchKernelGGL(skinny_reduce_rmsnorm_kernel, dim3(M), dim3(RN_THREADS), 0, st, part, splitk, M, N, (const bf16*)residual,
                       ldr, (bf16*)C, ldc, (const bf16*)norm_weight, (bf16*)Xn, ldn, eps);
  } else {
    const int ncols = epilogue == SK_SWIGLU ? N / 2 : N;
    const long total = (long)M * ncols;
    hipLaunchKernelGGL(skinny_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part, splitk, M, N,
                       (const bf16*)bias, (const bf16*)residual, ldr, (bf16*)C, ldc, epilogue);
  }
  BAGEL_LAUNCH_CHECK();
  return 0;
}

extern "C" int bagel_skinny_gemm_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M,
                                      int N, int K, const void* bias, const void* residual, long ldr,
                                      int epilogue, void* workspace, long workspace_bytes, void* stream) {
  return skinny_launch(X, ldx, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, workspace, workspace_bytes, nullptr,
                       nullptr, 0, 0.f, stream);
}

extern "C" int bagel_skinny_gemm_residual_rmsnorm_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc,
                                                       int M, int N, int K, const void* residual, long ldr,
                                                       const void* norm_weight, void* Xn, long ldn, float eps,
                                                       void* workspace, long workspace_bytes, void* stream) {
  BAGEL_CHECK(norm_weight && Xn && ldn >= N, "bagel_skinny_gemm_residual_rmsnorm_bf16: null norm operand");
  return skinny_launch(X, ldx, W, ldw, C, ldc, M, N, K, nullptr, residual, ldr, SK_RESIDUAL, workspace, workspace_bytes,
                       norm_weight, Xn, ldn, eps, stream);
}
