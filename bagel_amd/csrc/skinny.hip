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
#include <type_traits>

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
    // plain loads: the non-temporal policy measured 9-12 % slower on this stream (4.9 vs 5.5 TB/s on lm_head)
#pragma unroll
#ifdef BAGEL_SKINNY_NT
    for (int j = 0; j < 4; ++j) wf[st][j] = __builtin_nontemporal_load((const bf16x8*)(wp + step * 128 + j * 32));
#else
    for (int j = 0; j < 4; ++j) wf[st][j] = *(const bf16x8*)(wp + step * 128 + j * 32);
#endif
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

// The same reduce + residual + RMSNorm launch for partial tiles that another kernel left in [slices][M][N] order (the
// split-K form of the tile GEMM, gemm.hip).  Returns false when the shape is outside the kernel's limits.
bool bagel_launch_reduce_residual_rmsnorm(const float* part, int slices, int M, int N, const bf16* R, long ldr, bf16* C, long ldc,
                                          const bf16* norm_w, bf16* Xn, long ldn, float eps, hipStream_t st) {
  if (slices > 16 || N > RN_THREADS * RN_MAXC) return false;
  hipLaunchKernelGGL(skinny_reduce_rmsnorm_kernel, dim3(M), dim3(RN_THREADS), 0, st, part, slices, M, N, R, ldr, C, ldc, norm_w, Xn,
                     ldn, eps);
  return true;
}

// ---------------------------------------------------------------------------------------------
// At most 4 token rows (a decode step of up to four samples): a GEMV-shaped kernel with no split-K and no partial
// sums.  A wave owns NF weight rows over the WHOLE reduction; its 64 lanes split K in 16-byte pieces, so every load
// instruction reads 1 KiB that is contiguous in memory (the weight row itself), ST - 1 steps in flight behind the one
// being multiplied.  The few activation rows are staged once per workgroup in LDS (its four waves share the copy),
// optionally RMS-normalised on the way (NORM: Qwen2RMSNorm, modeling_qwen2.py:54-59: x_n = w * bf16(x * rs),
// rs = rsqrt(mean(x^2) + eps)), while the first weight stages are already in flight; registers then hold weights
// only, and a wave's own instruction chain stays short - a projection of a decode step has two or three waves per
// SIMD, so whatever a wave computes besides the products is exposed.
// v_dot2c_f32_bf16 accumulates in fp32 per lane; the lanes' sums are combined by a butterfly that halves the number of
// live values at every exchange, after which lane (64 / V) v holds output value v.  Bias / SiLU / residual / SwiGLU
// are applied there, so one launch replaces the partial-sum and reduce launches of the split-K form (and the launch
// of a separate normalisation kernel) and writes no fp32 intermediates.
// Values are numbered v = m * NF + f (row m, weight row f of the wave).  With SwiGLU the wave's NF weight rows
// are NF/2 (gate, up) pairs: rows f and f + 16 of the [16 gate | 16 up] interleave.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// One butterfly level over lane ^ WD with NL live values in val[0 .. NL): NL > 1 halves them (a lane keeps the values
// whose index bit equals its own lane bit and receives the partner's copies), NL == 1 is a plain sum.  Either way
// val[i] = own + partner's on the same operands.
template <int WD, int NL, int V>
__device__ __forceinline__ void bfly_level(float (&val)[V], int lane) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
  constexpr int PAIRS = NL > 1 ? NL / 2 : 1;
  constexpr int OFF = NL > 1 ? NL / 2 : 0;
  const bool upper = (lane & WD) != 0;
#pragma unroll
  for (int i = 0; i < PAIRS; ++i) {
    const float a = val[i], b = val[i + OFF];
    if constexpr (WD == 32) {
      const u32x2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
      val[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (WD == 16) {
      const u32x2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
      val[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
      const float keep = upper ? b : a, give = upper ? a : b;
      val[i] = keep + xor_lane<WD>(give);
    }
  }
}

template <int NF, int MR, int ST, bool NORM>
__global__ __launch_bounds__(1024) void gemv_rows_kernel(const bf16* __restrict__ X, long ldx, const bf16* __restrict__ W,
                                                         long ldw, bf16* __restrict__ C, long ldc, int M, int N, int K,
                                                         const bf16* __restrict__ bias, const bf16* __restrict__ R,
                                                         long ldr, int epilogue, const bf16* __restrict__ norm_w,
                                                         float eps) {
  constexpr int V = NF * MR;              // values per wave, a power of two <= 32
  static_assert((V & (V - 1)) == 0 && V <= 32, "NF * MR must be a power of two, at most 32");
  const int tid = threadIdx.x, lane = tid & 63;
  const int nthreads = blockDim.x, nwaves = nthreads >> 6;   // the launcher picks the workgroup size (4 to 16 waves)
  const int wave = blockIdx.x * nwaves + (tid >> 6);
  const bool swiglu = epilogue == SK_SWIGLU;
  // weight rows of this wave
  int wrow[NF];
  if (swiglu) {
    // pair p of the wave covers output feature o = wave * (NF / 2) + p: gate row (o / 16) * 32 + o % 16, up row + 16
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int o = wave * (NF / 2) + (f >> 1);
      wrow[f] = (o >> 4) * 32 + (o & 15) + (f & 1) * 16;
    }
  } else {
#pragma unroll
    for (int f = 0; f < NF; ++f) wrow[f] = wave * NF + f;
  }
  // a wave whose rows are all out of range (N is a multiple of NF / of 32 with SwiGLU) still helps to stage the rows
  const bool live = wrow[0] < N;
  const bf16* wp[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) wp[f] = W + (long)min(wrow[f], N - 1) * ldw + lane * 8;

  float acc[MR][NF];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[m][f] = 0.f;

  const int nsteps = K / 512;             // K % 512 == 0 is checked by the launcher
  extern __shared__ __attribute__((aligned(16))) bf16 xs[];   // the (normalised) activation rows [MR][K]
  __shared__ float rs_s[MR];
  // ST register stages: the loads of the next ST - 1 steps are in flight behind the step being multiplied
  bf16x8 wf[ST][NF];
  auto load = [&](int st, int step) {
    // non-temporal: every weight byte is read once by one wave; measured 5-15 % faster than the default policy on
    // this kernel (decode step 3.15 against 3.30-3.46 ms per token, the vocabulary projection 6.7 against 5.8 TB/s)
#pragma unroll
#ifndef BAGEL_GEMV_DEFAULT_POLICY
    for (int f = 0; f < NF; ++f) wf[st][f] = __builtin_nontemporal_load((const bf16x8*)(wp[f] + step * 512));
#else
    for (int f = 0; f < NF; ++f) wf[st][f] = *(const bf16x8*)(wp[f] + step * 512);
#endif
  };
  auto compute = [&](int st, int cstep) {   // cstep: the step's index along K
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const bf16x8 x = *(const bf16x8*)(xs + (long)m * K + cstep * 512 + lane * 8);
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[m][f] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{wf[st][f][2 * e], wf[st][f][2 * e + 1]},
                                                      bf16x2_t{x[2 * e], x[2 * e + 1]}, acc[m][f], false);
    }
  };
  // the workgroup's threads stage the rows in LDS
  auto stage = [&]() {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const bf16* xr = X + (long)min(m, M - 1) * ldx;
      for (int e0 = tid * 8; e0 < K; e0 += nthreads * 8) {
        bf16x8 v = *(const bf16x8*)(xr + e0);
        if constexpr (NORM) {
          const bf16x8 nw = *(const bf16x8*)(norm_w + e0);
          const float rsm = rs_s[m];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)nw[j] * round_bf16((float)v[j] * rsm));
        }
        *(bf16x8*)(xs + (long)m * K + e0) = v;
      }
    }
  };
  if (live) {   // the first weight stages are in flight while the rows are staged
#pragma unroll
    for (int j = 0; j < ST - 1; ++j)
      if (j < nsteps) load(j, j);
  }
  if constexpr (NORM) {
    __shared__ float red[MR][16];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const bf16* xr = X + (long)min(m, M - 1) * ldx;
      float ss = 0.f;
      for (int e0 = tid * 8; e0 < K; e0 += nthreads * 8) {
        const bf16x8 v = *(const bf16x8*)(xr + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
      }
      ss = wave_sum(ss);
      if (lane == 0) red[m][tid >> 6] = ss;
    }
    __syncthreads();
    if (tid < MR) {
      float tot = 0.f;
      for (int w = 0; w < nwaves; ++w) tot += red[tid][w];
      rs_s[tid] = rsqrtf(tot / (float)K + eps);
    }
    __syncthreads();
  }
  stage();
  __syncthreads();
  if (!live) return;
  for (int step = 0; step < nsteps; step += ST) {
#pragma unroll
    for (int j = 0; j < ST; ++j) {
      if (step + j + ST - 1 < nsteps) load((j + ST - 1) % ST, step + j + ST - 1);
      if (step + j < nsteps) compute(j, step + j);
    }
  }

  // Butterfly over lane ^ 32, 16, 8, 4, 2, 1, always in that order: while more than one value is live an exchange
  // halves their number (a lane keeps the values whose index bit equals its own lane bit and receives the partner's
  // copies of them), afterwards a plain sum.  Either way a level computes own + partner on the same operands, so a
  // row's result does not depend on how many rows (MR) the launch carries.  The levels over 32 and 16 lanes are the
  // gfx950 lane swaps (the sum of a swap's two results is own + partner on every lane).
  float val[V];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int f = 0; f < NF; ++f) val[m * NF + f] = acc[m][f];
  bfly_level<32, V, V>(val, lane);
  bfly_level<16, (V / 2 > 1 ? V / 2 : 1), V>(val, lane);
  bfly_level<8, (V / 4 > 1 ? V / 4 : 1), V>(val, lane);
  bfly_level<4, (V / 8 > 1 ? V / 8 : 1), V>(val, lane);
  bfly_level<2, (V / 16 > 1 ? V / 16 : 1), V>(val, lane);
  bfly_level<1, (V / 32 > 1 ? V / 32 : 1), V>(val, lane);
  const float total = val[0];
  // every lane of a group of 64 / V lanes now holds the total of value v = lane / (64 / V)
  constexpr int GROUP = 64 / V;
  const int v = lane / GROUP, m = v / NF, f = v % NF;
  float partner = 0.f;
  if (swiglu) partner = __shfl_xor(total, GROUP, 64);      // gate lanes (f even) receive the up value of their pair
  if ((lane % GROUP) != 0 || m >= M) return;
  if (swiglu) {
    if (f & 1) return;
    const int o = wave * (NF / 2) + (f >> 1);
    const float g = round_bf16(total), u = round_bf16(partner);
    C[(long)m * ldc + o] = (bf16)(round_bf16(silu_f32(g)) * u);
    return;
  }
  const int n = wrow[0] + f;
  if (n >= N) return;
  float a = total;
  if (bias) a += (float)bias[n];
  if (epilogue == SK_RESIDUAL) a = round_bf16(a) + (float)R[(long)m * ldr + n];
  if (epilogue == SK_BIAS_SILU) a = silu_f32(round_bf16(a));
  C[(long)m * ldc + n] = (bf16)a;
}

constexpr long GEMV_LDS = 96 * 1024;   // staged activation rows per workgroup

static int gemv_rows_template(int M) { return M <= 1 ? 1 : M <= 2 ? 2 : 4; }

static bool gemv_eligible(int M, int N, int K, int epilogue) {
  if (K % 512 != 0 || N % 4 != 0 || (epilogue == SK_SWIGLU && N % 32 != 0)) return false;
  if ((long)gemv_rows_template(M) * K * 2 > GEMV_LDS) return false;   // the staged rows must fit LDS
  // measured on MI355X against the split-K form (tools/skinny_bench.py): ahead at one or two rows for every shape,
  // at three or four rows only for the narrow projections; beyond that the dot-product instruction is the bound
  // (eight rows: 2.3 TB/s) and the MFMA form wins
  return M <= 2 || (M <= 4 && epilogue != SK_SWIGLU && N <= 16384);
}

// ---------------------------------------------------------------------------------------------
// One-launch MFMA form for any row count up to 64: the split-K idea with the K slices inside ONE workgroup.
// A workgroup owns RB blocks of 16 weight rows (with SwiGLU: gate blocks and their up blocks alternate; the token-row
// fragments are loaded once per step for all RB blocks); its waves take
// the 128-deep K steps round-robin (slice q: steps q, q + ns, ...) with the split-K kernel's operand layout and two
// register stages, leave their fp32 accumulators in LDS, and waves 0 .. G-1 add them in slice order and apply the
// epilogue.  No partial sums in memory, no second launch.  The K split depends only on (N, K), and a row's products
// never meet another row's, so a row's result does not depend on how many rows share the launch.
// ---------------------------------------------------------------------------------------------
template <int G, int RB, int MAXT>
__global__ __launch_bounds__(MAXT) void stream_mfma_kernel(const bf16* __restrict__ X, long ldx, const bf16* __restrict__ W,
                                                           long ldw, bf16* __restrict__ C, long ldc, int M, int N, int K,
                                                           const bf16* __restrict__ bias, const bf16* __restrict__ R,
                                                           long ldr, int epilogue, int ns) {
  extern __shared__ __attribute__((aligned(16))) f32x4 red_s[];   // [ns][RB][G][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int row0 = blockIdx.x * (16 * RB);
  const int nsteps = K / 128;
  const bf16* wp[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) wp[b] = W + (long)(row0 + b * 16 + r) * ldw + h * 8;
  const bf16* xp[G];
#pragma unroll
  for (int g = 0; g < G; ++g) xp[g] = X + (long)min(g * 16 + r, M - 1) * ldx + h * 8;
  f32x4 acc[RB][G];
  bf16x8 wf[2][RB][4], xf[2][G][4];
  auto load = [&](int st, int step) {
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[st][b][j] = *(const bf16x8*)(wp[b] + step * 128 + j * 32);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[st][g][j] = *(const bf16x8*)(xp[g] + step * 128 + j * 32);
  };
  auto compute = [&](int st) {
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[b][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[st][b][j], xf[st][g][j], acc[b][g], 0, 0, 0);
  };
  // K slice q is steps q, q + ns, ...; a wave takes slices wave, wave + nw, ... (one each unless the register budget
  // of many token rows limits the workgroup to fewer waves than slices)
  for (int q = wave; q < ns; q += nw) {
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int g = 0; g < G; ++g) acc[b][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < nsteps) load(0, q);
    for (int s = q; s < nsteps; s += 2 * ns) {
      if (s + ns < nsteps) load(1, s + ns);
      compute(0);
      if (s + 2 * ns < nsteps) load(0, s + 2 * ns);
      if (s + ns < nsteps) compute(1);
    }
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int g = 0; g < G; ++g) red_s[((q * RB + b) * G + g) * 64 + lane] = acc[b][g];
  }
  __syncthreads();
  if (wave >= G) return;
  // wave g finishes row group g: lane (r, h) holds features row0 + 4h .. 4h + 3 of token row m = 16 g + r
  const int m = wave * 16 + r;
  f32x4 v[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    v[b] = red_s[(b * G + wave) * 64 + lane];
    for (int q = 1; q < ns; ++q) v[b] += red_s[((q * RB + b) * G + wave) * 64 + lane];
  }
  if (m >= M) return;
  bf16x4 out;
  if (epilogue == SK_SWIGLU) {   // even blocks hold the gate rows, odd blocks the up rows of the same 16 features
    if constexpr (RB >= 2) {
#pragma unroll
      for (int pr = 0; pr < RB / 2; ++pr) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          out[i] = (bf16)(round_bf16(silu_f32(round_bf16(v[2 * pr][i]))) * round_bf16(v[2 * pr + 1][i]));
        *(bf16x4*)(C + (long)m * ldc + (blockIdx.x * (RB / 2) + pr) * 16 + h * 4) = out;
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const int n = row0 + b * 16 + h * 4;
    bf16x4 rv = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (epilogue == SK_RESIDUAL) rv = *(const bf16x4*)(R + (long)m * ldr + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = v[b][i];
      if (bias) a += (float)bias[n + i];
      if (epilogue == SK_RESIDUAL) a = round_bf16(a) + (float)rv[i];
      if (epilogue == SK_BIAS_SILU) a = silu_f32(round_bf16(a));
      out[i] = (bf16)a;
    }
    *(bf16x4*)(C + (long)m * ldc + n) = out;
  }
}

// The same kernel with the weights brought in by LDS-DMA: the register form reads a weight row in 64-byte pieces (the
// MFMA operand layout ties four lanes to a row), two instructions per 128-byte line, which costs twice the requests
// per byte and rules out the non-temporal policy.  Here a wave's DMA instruction moves 1 KiB = four rows x 256
// contiguous bytes (whole lines, non-temporal) into the wave's own ring of S slots in LDS - no other wave touches it,
// so a counted s_waitcnt is the only synchronisation - and the MFMA fragments are read from there (16-byte chunk
// index XOR row, applied to the DMA source address and to the read, spreads the 16 rows over all banks).
// K slices, step order and the reduction are those of the register form: the two give the same bits.
template <int G, int RB, int S>
__global__ __launch_bounds__(512) void stream_dma_kernel(const bf16* __restrict__ X, long ldx, const bf16* __restrict__ W,
                                                         long ldw, bf16* __restrict__ C, long ldc, int M, int N, int K,
                                                         const bf16* __restrict__ bias, const bf16* __restrict__ R,
                                                         long ldr, int epilogue, int ns) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor builtins exist only in the device pass
  extern __shared__ __attribute__((aligned(16))) char dma_smem[];
  constexpr int SLOT = RB * 4096;          // one 128-deep step of the workgroup's RB row blocks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int r = lane & 15, h = lane >> 4;
  const int row0 = blockIdx.x * (16 * RB);
  const int nsteps = K / 128;
  char* ring = dma_smem + wave * (S * SLOT);
  f32x4* red_s = (f32x4*)(dma_smem + nw * (S * SLOT));   // [ns][RB][G][64]
  // DMA: piece i of a block is rows 4 i .. 4 i + 3; lane l lands at byte 16 l of the piece = row 4 i + (l >> 4),
  // position l & 15, and fetches the chunk that belongs there
  // (buffer addressing: the compiler's wait-count pass treats the global form of the DMA load as a flat access that may
  // touch LDS and then drains every counter at the first register use of an ordinary load)
  const bf16* w_base = W + (long)row0 * ldw;
  int dvoff[RB][4];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = 4 * i + (lane >> 4);
      dvoff[b][i] = (int)(((long)(b * 16 + rr) * ldw + (((lane & 15) ^ rr) * 8)) * 2);
    }
  int rd_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rd_off[j] = r * 256 + (((4 * j + h) ^ r) * 16);
  const bf16* xp[G];
#pragma unroll
  for (int g = 0; g < G; ++g) xp[g] = X + (long)min(g * 16 + r, M - 1) * ldx + h * 8;

  auto issue = [&](int slot, int step) {
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(ring + slot * SLOT + b * 4096 + i * 1024), 16, dvoff[b][i],
                                                 step * 256, 0, 2 /* non-temporal */);
  };
  // X fragments travel with the slots, S - 1 steps ahead, in a register ring: requested right behind the slot's DMA,
  // they complete in the same order, and one count covers both
  bf16x8 xf[S][G][4];
  auto xload = [&](const int B, int step) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[B][g][j] = *(const bf16x8*)(xp[g] + step * 128 + j * 32);
  };
  // the wave's steps, slice after slice: slice q is steps q, q + ns, ...; the wave takes slices wave, wave + nw, ...
  auto advance = [&](int& q, int& st) {
    st += ns;
    if (st >= nsteps) {
      q += nw;
      st = q;
    }
  };
  f32x4 acc[RB][G];
#pragma unroll
  for (int b = 0; b < RB; ++b)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[b][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  int pq = wave, ps = wave;   // producer: next step to request
  int cq = wave, cs = wave;   // consumer: step being multiplied
#pragma unroll
  for (int k = 0; k < S - 1; ++k)
    if (pq < ns) {
      issue(k, ps);
      xload(k, ps);
      advance(pq, ps);
    }
  // one step out of slot B (B and FULL are literals at every call and the lambda is always inlined, so both fold).
  // FULL: the steady state, where the step requests the slot S - 1 ahead and the wait count is exact.
  auto step = [&](const int B, const bool FULL) __attribute__((always_inline)) {
    constexpr int AHEAD = (S - 1) * (4 * RB + 4 * G);
    static_assert(AHEAD < 64, "vmcnt is a 6-bit count");
    if (FULL) {
      issue((B + S - 1) % S, ps);
      xload((B + S - 1) % S, ps);
      advance(pq, ps);
      // everything older than the S - 1 steps in flight has landed: this step's slot and X fragments
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(AHEAD) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const char* slot = ring + B * SLOT;
    bf16x8 wf[RB][4];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[b][j] = *(const bf16x8*)(slot + b * 4096 + rd_off[j]);
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[b][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[b][j], xf[B][g][j], acc[b][g], 0, 0, 0);
    int nq = cq, nst = cs;
    advance(nq, nst);
    if (nq != cq) {   // the slice is complete
#pragma unroll
      for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          red_s[((cq * RB + b) * G + g) * 64 + lane] = acc[b][g];
          acc[b][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    cq = nq;
    cs = nst;
  };
  // steps of this wave: slice q has nsteps / ns steps, one more when q < nsteps % ns
  int total = 0;
  {
    const int base = nsteps / ns, rem = nsteps - base * ns;
    for (int q = wave; q < ns; q += nw) total += base + (q < rem ? 1 : 0);
  }
  const int full = max(total - (S - 1), 0);   // steps that still have a slot to request
  int i = 0;
  for (; i + S <= full; i += S) {
    step(0, true);
    step(1, true);
    if (S > 2) step(2 % S, true);
    if (S > 3) step(3 % S, true);
  }
  // fewer than S steps that request, then the at most S - 1 steps already requested
  int slot = 0;
  for (; i < total; ++i) {
    const bool f = i < full;
    if (slot == 0) {
      if (f) step(0, true);
      else step(0, false);
    } else if (slot == 1) {
      if (f) step(1, true);
      else step(1, false);
    } else if (S > 2 && slot == 2) {
      if (f) step(2 % S, true);
      else step(2 % S, false);
    } else if (S > 3) {
      if (f) step(3 % S, true);
      else step(3 % S, false);
    }
    slot = slot + 1 == S ? 0 : slot + 1;
  }
  __syncthreads();
  if (wave >= G) return;
  const int m = wave * 16 + r;
  f32x4 v[RB];
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    v[b] = red_s[(b * G + wave) * 64 + lane];
    for (int q = 1; q < ns; ++q) v[b] += red_s[((q * RB + b) * G + wave) * 64 + lane];
  }
  if (m >= M) return;
  bf16x4 out;
  if (epilogue == SK_SWIGLU) {
    if constexpr (RB >= 2) {
#pragma unroll
      for (int pr = 0; pr < RB / 2; ++pr) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          out[i] = (bf16)(round_bf16(silu_f32(round_bf16(v[2 * pr][i]))) * round_bf16(v[2 * pr + 1][i]));
        *(bf16x4*)(C + (long)m * ldc + (blockIdx.x * (RB / 2) + pr) * 16 + h * 4) = out;
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < RB; ++b) {
    const int n = row0 + b * 16 + h * 4;
    bf16x4 rv = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    if (epilogue == SK_RESIDUAL) rv = *(const bf16x4*)(R + (long)m * ldr + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = v[b][i];
      if (bias) a += (float)bias[n + i];
      if (epilogue == SK_RESIDUAL) a = round_bf16(a) + (float)rv[i];
      if (epilogue == SK_BIAS_SILU) a = silu_f32(round_bf16(a));
      out[i] = (bf16)a;
    }
    *(bf16x4*)(C + (long)m * ldc + n) = out;
  }
#endif
}

static int g_stream_dma = 1;   // development knob: 0 keeps the register form at every row count (bagel_set_stream_dma)
extern "C" void bagel_set_stream_dma(int on) { g_stream_dma = on; }

template <int G, int RB, int S>
static void stream_dma_launch_one(const bf16* X, long ldx, const bf16* W, long ldw, bf16* C, long ldc, int M, int N, int K,
                                  const bf16* bias, const bf16* R, long ldr, int epilogue, int ns, hipStream_t st) {
  const int blocks = N / (16 * RB);
  const int nw = std::max(G, std::min(ns, 8));
  const size_t lds = (size_t)nw * S * RB * 4096 + (size_t)ns * RB * G * 1024;
  static BagelLdsAllowed allowed;
  bagel_allow_lds(allowed, (const void*)stream_dma_kernel<G, RB, S>, 160 * 1024);
  hipLaunchKernelGGL((stream_dma_kernel<G, RB, S>), dim3(blocks), dim3(64 * nw), lds, st, X, ldx, W, ldw, C, ldc, M, N, K, bias,
                     R, ldr, epilogue, ns);
}

// Waves per workgroup: enough waves on the chip (about 24 per CU when the shape allows) and K steps that divide evenly.
static int stream_waves_per_block(int blocks, int ksteps, int max_nw) {
  const long want = 24L * bagel_cu_count();
  int ideal = (int)std::min<long>(max_nw, std::max<long>(2, (want + blocks - 1) / blocks));
  ideal = std::min(ideal, ksteps);
  int best = ideal;
  double best_score = -1.0;
  for (int nw = std::max(1, ideal / 2); nw <= std::min({max_nw, ksteps, 2 * ideal}); ++nw) {
    const int per = (ksteps + nw - 1) / nw;
    const double balance = (double)ksteps / ((double)per * nw);              // 1.0: every wave has the same steps
    const double close = 1.0 - 0.1 * std::abs(nw - ideal) / (double)ideal;   // mild preference for the ideal count
    if (balance * close > best_score) {
      best_score = balance * close;
      best = nw;
    }
  }
  return best;
}

static int g_stream_mfma = 1;   // development knob: 0 keeps the split-K form (bagel_set_stream_mfma)
extern "C" void bagel_set_stream_mfma(int on) { g_stream_mfma = on; }

template <int G, int RB, int MAXT>
static void stream_launch_one(const bf16* X, long ldx, const bf16* W, long ldw, bf16* C, long ldc, int M, int N, int K,
                              const bf16* bias, const bf16* R, long ldr, int epilogue, int ns, hipStream_t st) {
  const int blocks = N / (16 * RB);
  // one wave per K slice where the workgroup may have that many; at least G waves: wave g finishes row group g
  const int nw = std::max(G, std::min(ns, MAXT / 64));
  hipLaunchKernelGGL((stream_mfma_kernel<G, RB, MAXT>), dim3(blocks), dim3(64 * nw), (size_t)ns * RB * G * 1024, st, X, ldx, W,
                     ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns);
}

// Row blocks per workgroup: one (two with SwiGLU), except for very wide projections (the vocabulary) at more than 8
// token rows, where four blocks share every load of the token-row fragments - as many bytes per step as a weight
// block once there are 16 rows (222 against 268 us at 16 rows; the same on gate+up, 592 workgroups, loses 14 %).
// More than 16 rows need 256 registers: 512-thread workgroups.
static void stream_launch(const bf16* X, long ldx, const bf16* W, long ldw, bf16* C, long ldc, int M, int N, int K,
                          const bf16* bias, const bf16* R, long ldr, int epilogue, hipStream_t st) {
  const int G = (M + 15) / 16;
  const bool sw = epilogue == SK_SWIGLU;
  // K slices: a function of the projection alone (never of the row count), so that a row's fp32 sums keep their
  // order whatever shares the launch.  Their accumulators meet in LDS: ns * RB * G KiB <= 64 KiB.
  const int rb = sw ? 2 : 1;
  const int ns = stream_waves_per_block(N / (16 * rb), K / 128, 16 / rb);
  const bool wide = M > 8 && G <= 2 && !sw && N % 64 == 0 && N / 64 >= 8 * bagel_cu_count() && ns * 4 * G <= 64;
  if (g_stream_dma && G == 1 && !wide) {
    if (sw) stream_dma_launch_one<1, 2, 2>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns, st);
    else stream_dma_launch_one<1, 1, 3>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns, st);
    return;
  }
  if (g_stream_dma && !sw && !wide) {   // more than 16 rows: rings + accumulators fit LDS without SwiGLU's second row block
    if (G == 2) stream_dma_launch_one<2, 1, 3>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns, st);
    else if (G == 3) stream_dma_launch_one<3, 1, 2>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns, st);
    else stream_dma_launch_one<4, 1, 2>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns, st);
    return;
  }
#define STREAM(G_, RB_, MAXT_) stream_launch_one<G_, RB_, MAXT_>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, ns, st)
  switch (G) {
    case 1: wide ? STREAM(1, 4, 512) : sw ? STREAM(1, 2, 1024) : STREAM(1, 1, 1024); break;
    case 2: wide ? STREAM(2, 4, 512) : sw ? STREAM(2, 2, 512) : STREAM(2, 1, 1024); break;
    case 3: sw ? STREAM(3, 2, 512) : STREAM(3, 1, 512); break;
    default: sw ? STREAM(4, 2, 512) : STREAM(4, 1, 512); break;
  }
#undef STREAM
}

static int g_skinny_gemv = 1;   // development knob: 0 forces the split-K form (bagel_set_skinny_gemv)
extern "C" void bagel_set_skinny_gemv(int on) { g_skinny_gemv = on; }

// Waves per workgroup for launches whose workgroups are all resident at once: they are spread evenly over the CUs, so
// the launch lasts as long as the CU with the most waves.  3584 rows in workgroups of four 2-row waves are 448
// workgroups: 192 CUs stream two of them and 64 CUs one.  In workgroups of seven waves they are one per CU.
static int gemv_waves_per_block(long waves) {
  const long cus = bagel_cu_count();
  if (waves > 16 * cus) return 4;
  int best = 4;
  long best_span = -1;
  for (int wpb = 4; wpb <= 16; ++wpb) {
    const long blocks = (waves + wpb - 1) / wpb;
    const long span = (blocks + cus - 1) / cus * wpb;
    if (best_span < 0 || span < best_span) {
      best_span = span;
      best = wpb;
    }
  }
  return best;
}

template <int NF, int MR, int ST, bool NORM>
static void gemv_launch_one(const bf16* X, long ldx, const bf16* W, long ldw, bf16* C, long ldc, int M, int N, int K,
                            const bf16* bias, const bf16* R, long ldr, int epilogue, const bf16* norm_w, float eps,
                            hipStream_t st) {
  const int waves = N / NF, wpb = gemv_waves_per_block(waves);
  static BagelLdsAllowed allowed;
  bagel_allow_lds(allowed, (const void*)gemv_rows_kernel<NF, MR, ST, NORM>, (int)GEMV_LDS);
  hipLaunchKernelGGL((gemv_rows_kernel<NF, MR, ST, NORM>), dim3((waves + wpb - 1) / wpb), dim3(64 * wpb), (size_t)MR * K * 2, st,
                     X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, norm_w, eps);
}

static void gemv_launch(const bf16* X, long ldx, const bf16* W, long ldw, bf16* C, long ldc, int M, int N, int K,
                        const bf16* bias, const bf16* R, long ldr, int epilogue, const bf16* norm_w, float eps,
                        hipStream_t st) {
  // Two weight rows per wave.  A projection with few waves per CU (everything but gate+up and the vocabulary) needs
  // more steps in flight per wave to keep ~64 KB per CU outstanding: eight stages there, four otherwise.
  const bool deep = (long)N / 2 < 16L * bagel_cu_count();
#define GEMV(MR, ST)                                                                                                          \
  (norm_w ? gemv_launch_one<2, MR, ST, true>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, norm_w, eps, st)        \
          : gemv_launch_one<2, MR, ST, false>(X, ldx, W, ldw, C, ldc, M, N, K, bias, R, ldr, epilogue, norm_w, eps, st))
  switch (gemv_rows_template(M)) {
    case 1: deep ? GEMV(1, 8) : GEMV(1, 4); break;
    case 2: deep ? GEMV(2, 8) : GEMV(2, 4); break;
    default: deep ? GEMV(4, 8) : GEMV(4, 4); break;
  }
#undef GEMV
}

extern "C" long bagel_skinny_gemm_workspace_bytes(int M, int N) { return (long)16 * M * N * sizeof(float); }

static int skinny_launch(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                         const void* bias, const void* residual, long ldr, int epilogue, void* workspace,
                         long workspace_bytes, const void* norm_weight, void* Xn, long ldn, float eps, bool allow_gemv,
                         void* stream) {
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
  const bool stream_ok = g_stream_mfma && !norm_weight && ldc % 4 == 0 && (epilogue != SK_RESIDUAL || ldr % 4 == 0) &&
                         ((uintptr_t)C | (uintptr_t)residual) % 8 == 0;
  // three and four rows: the MFMA form with the weights brought in by DMA streams faster than four dot products per weight
  const bool gemv_pays = M <= 2 || !(stream_ok && g_stream_dma);
  if (allow_gemv && g_skinny_gemv && !norm_weight && gemv_pays && gemv_eligible(M, N, K, epilogue)) {
    gemv_launch((const bf16*)X, ldx, (const bf16*)W, ldw, (bf16*)C, ldc, M, N, K, (const bf16*)bias, (const bf16*)residual, ldr,
                epilogue, nullptr, 0.f, (hipStream_t)stream);
    BAGEL_LAUNCH_CHECK();
    return 0;
  }
  if (stream_ok) {
    stream_launch((const bf16*)X, ldx, (const bf16*)W, ldw, (bf16*)C, ldc, M, N, K, (const bf16*)bias, (const bf16*)residual, ldr,
                  epilogue, (hipStream_t)stream);
    BAGEL_LAUNCH_CHECK();
    return 0;
  }
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
  // always the split-K form: a row's result must not depend on how many rows share the launch (a CFG branch packed
  // with others equals that branch run alone)
  return skinny_launch(X, ldx, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, workspace, workspace_bytes, nullptr,
                       nullptr, 0, 0.f, false, stream);
}

// Same contract; shapes that bagel_skinny_gemv_eligible() accepts take the one-launch form (other summation order).
extern "C" int bagel_skinny_gemv_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M,
                                      int N, int K, const void* bias, const void* residual, long ldr,
                                      int epilogue, void* workspace, long workspace_bytes, void* stream) {
  return skinny_launch(X, ldx, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, workspace, workspace_bytes, nullptr,
                       nullptr, 0, 0.f, true, stream);
}

extern "C" int bagel_skinny_gemv_eligible(int M, int N, int K, int epilogue) {
  return g_skinny_gemv && gemv_eligible(M, N, K, epilogue) ? 1 : 0;
}

extern "C" int bagel_rmsnorm_bf16(const void* x, long ldx, void* out, long ldo, int rows, int H, const void* w_first,
                                  const void* w_second, int split_row, float eps, void* stream);

extern "C" int bagel_skinny_gemm_rmsnorm_in_bf16(const void* X, long ldx, const void* norm_weight, float eps, const void* W,
                                                 long ldw, void* C, long ldc, int M, int N, int K, const void* bias,
                                                 int epilogue, void* Xn, long ldn, void* workspace, long workspace_bytes,
                                                 void* stream) {
  BAGEL_CHECK(X && W && C && norm_weight, "bagel_skinny_gemm_rmsnorm_in_bf16: null operand");
  BAGEL_CHECK(M > 0 && M <= 64, "bagel_skinny_gemm_rmsnorm_in_bf16: M=%d outside 1..64", M);
  BAGEL_CHECK(epilogue == SK_BIAS || epilogue == SK_SWIGLU || epilogue == SK_BIAS_SILU,
              "bagel_skinny_gemm_rmsnorm_in_bf16: epilogue %d takes no normalised input", epilogue);
  BAGEL_CHECK(ldx % 8 == 0 && ldw % 8 == 0 && ldx >= K && ldw >= K && K % 8 == 0, "bagel_skinny_gemm_rmsnorm_in_bf16: bad leading dims");
  BAGEL_CHECK(((uintptr_t)X | (uintptr_t)W | (uintptr_t)norm_weight) % 16 == 0, "bagel_skinny_gemm_rmsnorm_in_bf16: misaligned operand");
  if (g_skinny_gemv && N > 0 && gemv_eligible(M, N, K, epilogue)) {
    gemv_launch((const bf16*)X, ldx, (const bf16*)W, ldw, (bf16*)C, ldc, M, N, K, (const bf16*)bias, nullptr, 0, epilogue,
                (const bf16*)norm_weight, eps, (hipStream_t)stream);
    BAGEL_LAUNCH_CHECK();
    return 0;
  }
  BAGEL_CHECK(Xn && ldn >= K && ldn % 8 == 0, "bagel_skinny_gemm_rmsnorm_in_bf16: this shape needs the Xn scratch rows");
  if (int rc = bagel_rmsnorm_bf16(X, ldx, Xn, ldn, M, K, norm_weight, norm_weight, M, eps, stream)) return rc;
  return skinny_launch(Xn, ldn, W, ldw, C, ldc, M, N, K, bias, nullptr, 0, epilogue, workspace, workspace_bytes, nullptr, nullptr,
                       0, 0.f, false, stream);
}

extern "C" int bagel_skinny_gemm_residual_rmsnorm_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc,
                                                       int M, int N, int K, const void* residual, long ldr,
                                                       const void* norm_weight, void* Xn, long ldn, float eps,
                                                       void* workspace, long workspace_bytes, void* stream) {
  BAGEL_CHECK(norm_weight && Xn && ldn >= N, "bagel_skinny_gemm_residual_rmsnorm_bf16: null norm operand");
  if (g_stream_mfma) {
    // the one-launch projection leaves the finished rows in C; their norm is one small launch more, the same two
    // launches as partial sums + (reduce, residual, norm), with the faster projection
    if (int rc = skinny_launch(X, ldx, W, ldw, C, ldc, M, N, K, nullptr, residual, ldr, SK_RESIDUAL, workspace, workspace_bytes,
                               nullptr, nullptr, 0, 0.f, false, stream))
      return rc;
    return bagel_rmsnorm_bf16(C, ldc, Xn, ldn, M, N, norm_weight, norm_weight, M, eps, stream);
  }
  return skinny_launch(X, ldx, W, ldw, C, ldc, M, N, K, nullptr, residual, ldr, SK_RESIDUAL, workspace, workspace_bytes,
                       norm_weight, Xn, ldn, eps, false, stream);
}
