// FLUX VAE decoder kernels (gfx950): implicit-GEMM 3x3 convolution on the MFMA units, GroupNorm + swish,
// the single-head mid-block attention helpers, and the layout converters at both ends.
//
// Activations are NHWC bf16 ([H*W][C]), so a 1x1 convolution IS bagel_gemm_bf16 and a 3x3 convolution is
// the same GEMM tile with K running over (ky, kx, c) and the A-operand rows gathered by the LDS-DMA's
// per-lane source address (out-of-image taps point at a zero page).  Nearest-neighbour 2x upsampling
// (Upsample.forward, autoencoder.py:116-119) is folded into that gather.  Rounding points follow what
// bf16 autocast does to the reference's fp32 VAE: convolutions consume and produce bf16, GroupNorm and
// swish run in fp32 on the bf16 input and are rounded once when the next convolution casts its input.
#include "gemm_tile.h"
#include "gemm_pipelined.h"

// -------------------------------------------------------------------------------------------------
// conv3x3: stride 1, padding 1 (resample mode 0); the same on the nearest-2x upsampled input (mode 1);
// stride 2 after padding one zero row/column on the bottom/right (mode 2; Downsample.forward,
// autoencoder.py:98-108).
// nn.Conv2d calls at autoencoder.py:76,78 (ResnetBlock), :102, :114 (Upsample), :170, :248 (conv_out).
// W is pre-arranged [Cout][ky][kx][Cin]; Cin % 64 == 0.
// -------------------------------------------------------------------------------------------------
// Development knob: 0 keeps every convolution on the 128 x 128 tile (A/B timing, tools/vae_bench.py).
static int g_conv_pipelined = 1;
extern "C" void bagel_set_conv_pipelined(int on) { g_conv_pipelined = on; }

extern "C" int bagel_conv3x3_nhwc_bf16(const void* x, int Hin, int Win, int Cin, int upsample, const void* w, int Cout,
                                       const void* bias, const void* residual, void* out, const void* zero_page,
                                       void* stream) {
  BAGEL_CHECK(x && w && bias && out && zero_page, "bagel_conv3x3_nhwc_bf16: null operand");
  BAGEL_CHECK(Cin % 64 == 0 && Cout % 4 == 0, "bagel_conv3x3_nhwc_bf16: Cin=%d must be a multiple of 64, Cout=%d of 4", Cin, Cout);
  BAGEL_CHECK(Hin > 0 && Win > 0, "bagel_conv3x3_nhwc_bf16: empty image");
  BAGEL_CHECK(upsample >= 0 && upsample <= 2, "bagel_conv3x3_nhwc_bf16: resample mode %d unknown", upsample);
  BAGEL_CHECK(upsample != 2 || (Hin % 2 == 0 && Win % 2 == 0), "bagel_conv3x3_nhwc_bf16: stride 2 needs even %dx%d", Hin, Win);
  GemmParams p;
  p.ksplit = 1;
  p.slab = 0;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.upsample = upsample;
  p.Hout = upsample == 1 ? 2 * Hin : upsample == 2 ? Hin / 2 : Hin;
  p.Wout = upsample == 1 ? 2 * Win : upsample == 2 ? Win / 2 : Win;
  p.A = (const bf16*)x; p.lda = Cin;
  p.W = (const bf16*)w; p.ldw = 9L * Cin;
  p.C = (bf16*)out; p.ldc = Cout;
  p.bias = (const bf16*)bias;
  p.R = (const bf16*)residual; p.ldr = Cout;
  p.M = p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin;
  p.alpha = 1.f;
  p.zeros = (const bf16*)zero_page;
  p.diag = nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (Cout <= 32 && !residual) {
    // conv_out (three output channels): a 128-wide tile would spend its whole width on padding, the pass over the
    // input is what costs: 128 x 32 tiles, one wave per 32 rows
    constexpr int BM = 128, BN = 32;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = 1;
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, 4, 1, EPI_BIAS, A_CONV3X3>), dim3(p.tiles_m), dim3(256),
                       2 * (BM + BN) * ROW_BYTES, st, p);
    BAGEL_LAUNCH_CHECK();
    return 0;
  }
  // Large images with 256 or 512 output channels: the software-pipelined 256 x 256 tile (persistent workgroups, one
  // per CU) once its tiles fill the chip: 15-20 % faster there.  With 128 output channels a 256 x 128 tile of the same
  // loop measured no faster than two 128 x 128 workgroups per CU (381 against 370 us at 1024 x 1024: half the matrix
  // work per DMA piece), so those layers keep the small tile.  Same K order either way: same bits.
  {
    const int bn = Cout % 256 == 0 ? 256 : 0;
    const long tiles = bn ? (long)((p.M + 255) / 256) * (Cout / bn) : 0;
    const int cus = bagel_cu_count() >= 8 ? bagel_cu_count() / 8 * 8 : 256;
    if (g_conv_pipelined && tiles >= cus) {
      p.tiles_m = (p.M + 255) / 256;
      p.tiles_n = Cout / bn;
      const dim3 grid((unsigned)(tiles < cus ? tiles : cus)), block(512);
      const size_t lds = 2 * (256 + bn) * ROW_BYTES;
#define LAUNCH_PIPE(EPI, BN_, WR_)                                                                   \
      do {                                                                                           \
        auto k = gemm256_pipelined_kernel<EPI, BN_, WR_, false, A_CONV3X3>;                          \
        static BagelLdsAllowed allowed;                                                              \
        bagel_allow_lds(allowed, (const void*)k, (int)lds);                                          \
        hipLaunchKernelGGL(k, grid, block, lds, st, p);                                              \
      } while (0)
      if (residual) LAUNCH_PIPE(EPI_BIAS_RESIDUAL, 256, 2);
      else LAUNCH_PIPE(EPI_BIAS, 256, 2);
#undef LAUNCH_PIPE
      BAGEL_LAUNCH_CHECK();
      return 0;
    }
  }
  constexpr int BM = 128, BN = 128;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * ROW_BYTES;
  const dim3 grid(p.tiles_m * p.tiles_n), block(256);
  if (residual)
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, 2, 2, EPI_BIAS_RESIDUAL, A_CONV3X3>), grid, block, lds, st, p);
  else
    hipLaunchKernelGGL((gemm_bf16_nt_kernel<BM, BN, 2, 2, EPI_BIAS, A_CONV3X3>), grid, block, lds, st, p);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// GroupNorm(32 groups, eps) [+ swish] on NHWC bf16.  nn.GroupNorm at autoencoder.py:43,75,77,247 and
// swish (:34-35).  Statistics per group over H*W*C/32 values (fp32 partial sums per block, fp64 across blocks);
// three passes: partial sums per block -> per-channel scale = rstd * gamma, shift = beta - mean * scale ->
// y = swish(x * scale + shift) -> bf16.
// -------------------------------------------------------------------------------------------------
constexpr int GN_BLOCKS = 512;

// Pass 1: every thread owns one 8-channel chunk (the same one in every iteration: the block's pixel stride is a
// whole number of rows) and walks pixels with four 16-byte loads in flight.
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16* __restrict__ x, long npix, int C,
                                                         float* __restrict__ part) {
  const int cpr = C / 8;
  const int chunk = threadIdx.x % cpr;
  const int ppb = 256 / cpr;  // pixels per block-iteration
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  const long stride = (long)gridDim.x * ppb;
  const bf16* src = x + chunk * 8;
  long pix = (long)blockIdx.x * ppb + threadIdx.x / cpr;
  for (; pix + 3 * stride < npix; pix += 4 * stride) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(src + (pix + u * stride) * C);
#pragma unroll
    for (int u = 0; u < 4; ++u)      // the same order of additions as one pixel per iteration
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[u][e];
        s[e] += f;
        q[e] += f * f;
      }
  }
  for (; pix < npix; pix += stride) {
    const bf16x8 v = *(const bf16x8*)(src + pix * C);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      s[e] += f;
      q[e] += f * f;
    }
  }
  __shared__ float sh[256][17];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[threadIdx.x][e] = s[e];
    sh[threadIdx.x][8 + e] = q[e];
  }
  __syncthreads();
  // thread t < C sums channel t over the block's pixel lanes
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int pl = 0; pl < ppb; ++pl) {
      a += sh[pl * cpr + c / 8][c % 8];
      b += sh[pl * cpr + c / 8][8 + c % 8];
    }
    part[((long)blockIdx.x * C + c) * 2] = a;
    part[((long)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

// Pass 2: one workgroup per group.  Thread t sums the (block, channel) partials t, t + 256, ... of its group in
// fp64; the 256 sums are then combined through LDS in a fixed order, so the statistics do not depend on timing.
// Output per channel: scale = rstd * gamma and shift = beta - mean * scale (what pass 3 applies).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int nblocks, int C, long npix,
                                                           float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ scale_shift) {
  const int g = blockIdx.x, cpg = C / 32;
  double s = 0.0, q = 0.0;
  const int n = nblocks * cpg;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int b = i / cpg, c = g * cpg + i - b * cpg;
    const float2 v = *(const float2*)(part + ((long)b * C + c) * 2);
    s += v.x;
    q += v.y;
  }
  __shared__ double sh_s[256], sh_q[256];
  __shared__ float sh_stat[2];
  sh_s[threadIdx.x] = s;
  sh_q[threadIdx.x] = q;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      sh_s[threadIdx.x] += sh_s[threadIdx.x + w];
      sh_q[threadIdx.x] += sh_q[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double cnt = (double)npix * cpg;
    const double mean = sh_s[0] / cnt;
    const double var = sh_q[0] / cnt - mean * mean;
    sh_stat[0] = (float)mean;
    sh_stat[1] = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
  }
  __syncthreads();
  for (int c = g * cpg + threadIdx.x; c < (g + 1) * cpg; c += 256) {
    const float sc = sh_stat[1] * gamma[c];
    scale_shift[2 * c] = sc;
    scale_shift[2 * c + 1] = beta[c] - sh_stat[0] * sc;
  }
}

// Pass 3: y = swish(x * scale + shift).  A thread keeps the scale / shift of its 8 channels in registers and walks
// pixels with four 16-byte loads in flight (a block's pixel stride is a whole number of rows, so the chunk is fixed).
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16* __restrict__ x, long npix, int C,
                                                       const float* __restrict__ scale_shift, int swish,
                                                       bf16* __restrict__ out) {
  const int cpr = C / 8;
  const int chunk = threadIdx.x % cpr;
  const int ppb = 256 / cpr;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float2 ss = *(const float2*)(scale_shift + 2 * (chunk * 8 + e));
    a[e] = ss.x;
    b[e] = ss.y;
  }
  auto apply = [&](const bf16x8 v) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float y = __builtin_fmaf((float)v[e], a[e], b[e]);
      if (swish) y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
      o[e] = (bf16)y;
    }
    return o;
  };
  const long stride = (long)gridDim.x * ppb;
  const long off = chunk * 8;
  long pix = (long)blockIdx.x * ppb + threadIdx.x / cpr;
  for (; pix + 3 * stride < npix; pix += 4 * stride) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8*)(x + (pix + u * stride) * C + off);
#pragma unroll
    for (int u = 0; u < 4; ++u) *(bf16x8*)(out + (pix + u * stride) * C + off) = apply(v[u]);
  }
  for (; pix < npix; pix += stride) *(bf16x8*)(out + pix * C + off) = apply(*(const bf16x8*)(x + pix * C + off));
}

extern "C" long bagel_groupnorm_workspace_bytes(int C) { return ((long)GN_BLOCKS * C * 2 + 2L * C) * sizeof(float); }

extern "C" int bagel_groupnorm_swish_nhwc_bf16(const void* x, long npix, int C, const void* gamma_f32,
                                               const void* beta_f32, float eps, int swish, void* out, void* workspace,
                                               long workspace_bytes, void* stream) {
  BAGEL_CHECK(x && gamma_f32 && beta_f32 && out && workspace, "bagel_groupnorm_swish_nhwc_bf16: null operand");
  BAGEL_CHECK(C % 32 == 0 && C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, "bagel_groupnorm_swish_nhwc_bf16: C=%d not supported", C);
  BAGEL_CHECK(npix > 0, "bagel_groupnorm_swish_nhwc_bf16: empty image");
  BAGEL_CHECK(workspace_bytes >= bagel_groupnorm_workspace_bytes(C), "bagel_groupnorm_swish_nhwc_bf16: workspace too small");
  BAGEL_CHECK(((uintptr_t)x | (uintptr_t)out | (uintptr_t)workspace) % 16 == 0, "bagel_groupnorm_swish_nhwc_bf16: misaligned operand");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* scale_shift = part + (long)GN_BLOCKS * C * 2;
  const int ppb = 256 / (C / 8);
  const long rows = (npix + ppb - 1) / ppb;     // block-iterations the image holds
  const int nblocks = (int)(rows < GN_BLOCKS ? rows : GN_BLOCKS);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nblocks), dim3(256), 0, st, (const bf16*)x, npix, C, part);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32), dim3(256), 0, st, part, nblocks, C, npix, eps, (const float*)gamma_f32,
                     (const float*)beta_f32, scale_shift);
  // eight workgroups per CU when the image is large enough; each thread then streams four pixels at a time
  const long want = (rows + 3) / 4;
  const int ablocks = (int)(want < 8L * bagel_cu_count() ? (want > 0 ? want : 1) : 8L * bagel_cu_count());
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ablocks), dim3(256), 0, st, (const bf16*)x, npix, C, scale_shift, swish, (bf16*)out);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Row softmax of fp32 scores -> bf16 probabilities (one block per row).  With bagel_gemm_bf16's fp32
// scaled epilogue (S = Q.K^T / sqrt(C)) and a second GEMM (P.V) this is AttnBlock.attention
// (autoencoder.py:50-62): single head, head dim = 512 channels, all H*W tokens.
// -------------------------------------------------------------------------------------------------
// Rows of up to 256 x 4 x ROW_REGS values are read ONCE and kept in registers between the three passes (maximum, sum,
// normalise): 16384 keys are 64 values per thread.  Longer rows take the re-reading form.
constexpr int SOFTMAX_ROW_REGS = 16;   // f32x4 per thread
template <bool IN_REGS>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long lds_, int n,
                                                           bf16* __restrict__ p, long ldp) {
  const float* row = s + (long)blockIdx.x * lds_;
  __shared__ float sh[8];
  f32x4 keep[IN_REGS ? SOFTMAX_ROW_REGS : 1];
  float m = -INFINITY;
  if constexpr (IN_REGS) {
#pragma unroll
    for (int r = 0; r < SOFTMAX_ROW_REGS; ++r) {
      const int i = threadIdx.x * 4 + r * 1024;
      keep[r] = i < n ? *(const f32x4*)(row + i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int r = 0; r < SOFTMAX_ROW_REGS; ++r) m = fmaxf(fmaxf(m, fmaxf(keep[r][0], keep[r][1])), fmaxf(keep[r][2], keep[r][3]));
  } else {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      const f32x4 v = *(const f32x4*)(row + i);
      m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  __syncthreads();
  // the same order of additions in both forms: a thread's values in index order, then the wave, then the four waves
  float sum = 0.f;
  if constexpr (IN_REGS) {
#pragma unroll
    for (int r = 0; r < SOFTMAX_ROW_REGS; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        keep[r][e] = __expf(keep[r][e] - m);      // exp(-inf) = 0 for the slots past the row's end
        sum += keep[r][e];
      }
  } else {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      const f32x4 v = *(const f32x4*)(row + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += __expf(v[e] - m);
    }
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) sh[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (sh[4] + sh[5] + sh[6] + sh[7]);
  bf16* prow = p + (long)blockIdx.x * ldp;
  if constexpr (IN_REGS) {
#pragma unroll
    for (int r = 0; r < SOFTMAX_ROW_REGS; ++r) {
      const int i = threadIdx.x * 4 + r * 1024;
      if (i < n) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)(keep[r][e] * inv);
        *(bf16x4*)(prow + i) = o;
      }
    }
  } else {
    for (int i = threadIdx.x * 4; i < n; i += 1024) {
      const f32x4 v = *(const f32x4*)(row + i);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (bf16)(__expf(v[e] - m) * inv);
      *(bf16x4*)(prow + i) = o;
    }
  }
}

extern "C" int bagel_softmax_rows_f32_bf16(const void* scores, long lds_, int rows, int n, void* probs, long ldp,
                                           void* stream) {
  BAGEL_CHECK(scores && probs && rows > 0 && n > 0 && n % 4 == 0 && lds_ % 4 == 0 && ldp % 4 == 0,
              "bagel_softmax_rows_f32_bf16: bad arguments");
  if (n <= 1024 * SOFTMAX_ROW_REGS)
    hipLaunchKernelGGL(softmax_rows_kernel<true>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)scores, lds_, n,
                       (bf16*)probs, ldp);
  else
    hipLaunchKernelGGL(softmax_rows_kernel<false>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const float*)scores, lds_, n,
                       (bf16*)probs, ldp);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// out[c][r] = in[r][c]   (bf16, 64x64 tiles through LDS): V^T for the P.V GEMM of the mid-block attention
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ in, int R, int C,
                                                             bf16* __restrict__ out) {
  __shared__ bf16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i / 64, c = i % 64;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(long)(r0 + r) * C + c0 + c] : (bf16)0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i / 64, r = i % 64;
    if (r0 + r < R && c0 + c < C) out[(long)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

extern "C" int bagel_transpose_bf16(const void* in, int R, int C, void* out, void* stream) {
  BAGEL_CHECK(in && out && R > 0 && C > 0, "bagel_transpose_bf16: bad arguments");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)in, R, C, (bf16*)out);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Decoder input: z (NCHW fp32, Cz channels) -> im2col rows [H*W][Kpad] bf16 of bf16(z / scale + shift)
// with K ordered (ky, kx, c) and zero padding (AutoEncoder.decode, autoencoder.py:320-322, followed by
// conv_in, :252).  Cz = 16 is too thin for the implicit-GEMM gather, so conv_in runs as a dense GEMM on
// these rows.
// -------------------------------------------------------------------------------------------------
__global__ void latent_im2col_kernel(const float* __restrict__ z, int Cz, int H, int W, float inv_scale, float shift,
                                     bf16* __restrict__ out, int Kpad) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * W * Kpad) return;
  const int k = idx % Kpad;
  const long pix = idx / Kpad;
  float v = 0.f;
  if (k < 9 * Cz) {
    const int tap = k / Cz, c = k % Cz;
    const int y = (int)(pix / W) + tap / 3 - 1, x = (int)(pix % W) + tap % 3 - 1;
    if (y >= 0 && y < H && x >= 0 && x < W) v = z[((long)c * H + y) * W + x] * inv_scale + shift;
  }
  out[idx] = (bf16)v;
}

extern "C" int bagel_vae_latent_im2col_bf16(const void* z_nchw_f32, int Cz, int H, int W, float scale_factor,
                                            float shift_factor, void* out, int Kpad, void* stream) {
  BAGEL_CHECK(z_nchw_f32 && out && Kpad >= 9 * Cz && Kpad % 64 == 0, "bagel_vae_latent_im2col_bf16: bad arguments");
  const long n = (long)H * W * Kpad;
  hipLaunchKernelGGL(latent_im2col_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float*)z_nchw_f32, Cz, H, W, 1.0f / scale_factor, shift_factor, (bf16*)out, Kpad);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// Decoder output: NHWC bf16 with Cpad channels -> NCHW bf16 with C channels
__global__ void nhwc_to_nchw_kernel(const bf16* __restrict__ in, long npix, int Cpad, int C, bf16* __restrict__ out) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npix * C) return;
  const int c = idx / npix;
  const long pix = idx % npix;
  out[idx] = in[pix * Cpad + c];
}

extern "C" int bagel_nhwc_to_nchw_bf16(const void* in, long npix, int Cpad, int C, void* out, void* stream) {
  BAGEL_CHECK(in && out && npix > 0 && C > 0 && C <= Cpad, "bagel_nhwc_to_nchw_bf16: bad arguments");
  const long n = npix * C;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)in,
                     npix, Cpad, C, (bf16*)out);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// -------------------------------------------------------------------------------------------------
// Encoder output -> latent: DiagonalGaussian + scale/shift (autoencoder.py:281-287, 315-318) on the bf16
// moments the encoder's conv_out produced; every elementwise op rounds to bf16 as it does on a bf16 tensor.
// moments NHWC [H*W][ldm] (mean channels 0..Cz-1, logvar Cz..2Cz-1), noise and z NCHW bf16.
// -------------------------------------------------------------------------------------------------
__global__ void vae_sample_latent_kernel(const bf16* __restrict__ mom, int ldm, const bf16* __restrict__ noise, int Cz,
                                         long npix, float scale, float shift, bf16* __restrict__ z) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npix * Cz) return;
  const int c = idx / npix;
  const long pix = idx % npix;
  const float mean = (float)mom[pix * ldm + c];
  const float logvar = (float)mom[pix * ldm + Cz + c];
  const float std = round_bf16(__expf(round_bf16(0.5f * logvar)));
  const float v = round_bf16(mean + round_bf16(std * (float)noise[idx]));
  z[idx] = (bf16)(scale * round_bf16(v - shift));
}

extern "C" int bagel_vae_sample_latent_bf16(const void* moments_nhwc, int ldm, const void* noise_nchw, int Cz, int H,
                                            int W, float scale_factor, float shift_factor, void* z_nchw, void* stream) {
  BAGEL_CHECK(moments_nhwc && noise_nchw && z_nchw && Cz > 0 && ldm >= 2 * Cz && H > 0 && W > 0,
              "bagel_vae_sample_latent_bf16: bad arguments");
  const long n = (long)H * W * Cz;
  hipLaunchKernelGGL(vae_sample_latent_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)moments_nhwc, ldm, (const bf16*)noise_nchw, Cz, (long)H * W, scale_factor, shift_factor,
                     (bf16*)z_nchw);
  BAGEL_LAUNCH_CHECK();
  return 0;
}
