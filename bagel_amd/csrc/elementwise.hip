// HBM-bound kernels of the generation forward (gfx950): RMSNorm with per-modality weights, per-head
// q/k RMSNorm + RoPE + KV-workspace scatter, latent embedding, timestep sinusoid, CFG combine + renorm +
// Euler update.  All loads/stores of bf16 rows are 16 bytes per lane.  Rounding points follow the
// reference's bf16 graph; each kernel names the lines it restates.
#include "common.h"
#include "qk_rope.h"

// ---------------------------------------------------------------------------------------------
// RMSNorm over rows of [rows, H]; rows < split use w_first, the others w_second.
// Qwen2RMSNorm.forward (modeling_qwen2.py:54-59): fp32 statistics, normalised value rounded to bf16,
// then multiplied by the bf16 weight and rounded again.  Per-modality weights:
// Qwen2MoTDecoderLayer.forward_inference (qwen2_navit.py:798-800, 826-829), final norm (:1102-1104).
// One wave per row.  Rows of up to 4096 elements stay in registers between the two passes (one HBM read,
// one write); wider rows are read a second time.
// ---------------------------------------------------------------------------------------------
template <int MAXC>  // 16-byte chunks per lane held in registers; 0 = re-read the row
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16* __restrict__ x, long ldx, bf16* __restrict__ out,
                                                      long ldo, int rows, int H, const bf16* __restrict__ w_first,
                                                      const bf16* __restrict__ w_second, int split, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const bf16* xr = x + (long)row * ldx;
  const bf16* w = row < split ? w_first : w_second;
  bf16* orow = out + (long)row * ldo;
  float ss = 0.f;
  if constexpr (MAXC > 0) {
    // the weights are requested with the row, not after its statistics: a few-row launch (a decode step) is two
    // dependent memory round trips otherwise
    bf16x8 v[MAXC], wv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < H) {
        v[i] = *(const bf16x8*)(xr + c);
        wv[i] = *(const bf16x8*)(w + c);
      }
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += (float)v[i][j] * (float)v[i][j];
      }
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane * 8 + i * 512;
      if (c < H) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)wv[i][j] * round_bf16((float)v[i][j] * rs));
        *(bf16x8*)(orow + c) = o;
      }
    }
  } else {
    for (int c = lane * 8; c < H; c += 512) {
      const bf16x8 v = *(const bf16x8*)(xr + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)H + eps);
    for (int c = lane * 8; c < H; c += 512) {
      const bf16x8 v = *(const bf16x8*)(xr + c);
      const bf16x8 wv = *(const bf16x8*)(w + c);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)wv[j] * round_bf16((float)v[j] * rs));
      *(bf16x8*)(orow + c) = o;
    }
  }
}

extern "C" int bagel_rmsnorm_bf16(const void* x, long ldx, void* out, long ldo, int rows, int H, const void* w_first,
                                  const void* w_second, int split_row, float eps, void* stream) {
  BAGEL_CHECK(x && out && w_first, "bagel_rmsnorm_bf16: null operand");
  BAGEL_CHECK(rows > 0 && H > 0 && H % 8 == 0, "bagel_rmsnorm_bf16: bad shape rows=%d H=%d", rows, H);
  BAGEL_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && ldx >= H && ldo >= H, "bagel_rmsnorm_bf16: bad leading dims");
  BAGEL_CHECK(split_row >= rows || w_second, "bagel_rmsnorm_bf16: rows past split_row need w_second");
  BAGEL_CHECK(((uintptr_t)x | (uintptr_t)out | (uintptr_t)w_first | (uintptr_t)w_second) % 16 == 0,
              "bagel_rmsnorm_bf16: misaligned operand");
  const dim3 grid((rows + 3) / 4), block(256);
#define ARGS (const bf16*)x, ldx, (bf16*)out, ldo, rows, H, (const bf16*)w_first, (const bf16*)w_second, split_row, eps
  if (H <= 1024)
    hipLaunchKernelGGL(rmsnorm_kernel<2>, grid, block, 0, (hipStream_t)stream, ARGS);
  else if (H <= 4096)
    hipLaunchKernelGGL(rmsnorm_kernel<8>, grid, block, 0, (hipStream_t)stream, ARGS);
  else
    hipLaunchKernelGGL(rmsnorm_kernel<0>, grid, block, 0, (hipStream_t)stream, ARGS);
#undef ARGS
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Per-head RMSNorm of q and k, RoPE, and scatter into the attention operands.
// PackedAttentionMoT.forward_inference (qwen2_navit.py:549-568, gen mode): q/k are up-cast to fp32,
// normalised per head with the modality's weight (no intermediate rounding), rotated with the bf16
// cos/sin of the token's position (rotate_half form, modeling_qwen2.py:154-186) and rounded to bf16 once.
// With und_rounding != 0 the und-mode graph is followed instead (:526-530): the norm output and every
// RoPE product are rounded to bf16.
// q goes to Q[row][head][D]; k goes straight to its key slot of the KV workspace K[kvh][slot][D], which
// removes the per-layer rebuild of the merged key tensor (qwen2_navit.py:574-580).
// D/8 lanes per (row, head); lane j holds elements 8j..8j+7, its rotate_half partner is lane j ^ (D/16).
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void qk_norm_rope_block(int block, const bf16* __restrict__ qkv, long ld, int M, int Hq,
                                                           int Hkv, const bf16* __restrict__ qw_first,
                                                           const bf16* __restrict__ qw_second,
                                                           const bf16* __restrict__ kw_first,
                                                           const bf16* __restrict__ kw_second, int split,
                                                           const bf16* __restrict__ cos_t,
                                                           const bf16* __restrict__ sin_t,
                                                           const int* __restrict__ row_cs,
                                                           const int* __restrict__ row_slot, bf16* __restrict__ Q,
                                                           bf16* __restrict__ Kws, long S, float eps,
                                                           int und_rounding) {
  constexpr int LPH = D / 8;
  // Two token rows per thread (four measured slower), every load of both issued before the first result is needed: with one 16-byte load in
  // flight per lane the kernel is bound by (resident waves x bytes per wave) / memory latency, not by the memory system.
  constexpr int U = 2;
  const int NH = Hq + Hkv;
  const long item = ((long)block * 256 + threadIdx.x) / LPH;
  const long pairs = (M + U - 1) / U;
  if (item >= pairs * NH) return;  // partial waves shuffle among live lanes only (a head's lanes leave together)
  const int j = threadIdx.x % LPH;
  const int row0 = (int)(item / NH) * U, head = item % NH;
  const bool is_q = head < Hq;
  bf16x8 v[U];
  long cs[U];
  int slot[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int row = min(row0 + u, M - 1);
    v[u] = *(const bf16x8*)(qkv + (long)row * ld + (long)head * D + 8 * j);  // q heads then k heads are contiguous
    cs[u] = (long)row_cs[row] * D;
    slot[u] = row_slot[row];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int row = row0 + u;
    if (row >= M) break;
    const bf16* w = is_q ? (row < split ? qw_first : qw_second) : (row < split ? kw_first : kw_second);
    const bf16x8 o = qk_norm_rope_lane<D>(v[u], j, w, cos_t + cs[u], sin_t + cs[u], eps, und_rounding);
    if (is_q)
      *(bf16x8*)(Q + ((long)row * Hq + head) * D + 8 * j) = o;
    else
      *(bf16x8*)(Kws + ((long)(head - Hq) * S + slot[u]) * D + 8 * j) = o;
  }
}

// V rows of the fused qkv projection -> transposed value workspace Vt[kvh][d][slot] (2-byte stores,
// consecutive lanes = consecutive token rows = consecutive slots, so a wave writes 128 contiguous bytes).
template <int D>
__device__ __forceinline__ void v_scatter_transposed_block(int block_x, int block_y, int grid_y, const bf16* __restrict__ qkv,
                                                           long ld, int M, int v_col0, int Hkv,
                                                           const int* __restrict__ row_slot, bf16* __restrict__ Vt, long S) {
  const int row = block_x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6;          // 4 waves split the Hkv*D/8 chunks
  if (row >= M) return;
  const long slot = row_slot[row];
  const bf16* src = qkv + (long)row * ld + v_col0;
  const int nchunk = Hkv * D / 8;
  // gridDim.y workgroups share a 64-row block's chunks: 64 rows per workgroup alone are too few workgroups to fill the
  // chip at a few thousand rows (192 at 12288 rows)
  for (int c = part + 4 * block_y; c < nchunk; c += 4 * grid_y) {
    const bf16x8 v = *(const bf16x8*)(src + 8 * c);
    bf16* dst = Vt + (long)(8 * c) * S + slot;  // (kvh*D + d) * S with kvh*D + d == 8c + e
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[(long)e * S] = v[e];
  }
}

// One launch for both: the first v_blocks workgroups scatter the values (bound by their 2-byte store instructions),
// the others normalise and rotate q / k (bound by memory latency); side by side each fills what the other leaves idle,
// and the step pays one launch boundary instead of two.
template <int D>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const bf16* __restrict__ qkv, long ld, int M, int Hq, int Hkv,
                                                           const bf16* __restrict__ qw_first, const bf16* __restrict__ qw_second,
                                                           const bf16* __restrict__ kw_first, const bf16* __restrict__ kw_second,
                                                           int split, const bf16* __restrict__ cos_t, const bf16* __restrict__ sin_t,
                                                           const int* __restrict__ row_cs, const int* __restrict__ row_slot,
                                                           bf16* __restrict__ Q, bf16* __restrict__ Kws, long S, float eps,
                                                           int und_rounding, bf16* __restrict__ Vt, int v_blocks, int v_split) {
  const int b = blockIdx.x;   // uniform over the workgroup
  if (b < v_blocks) {
    v_scatter_transposed_block<D>(b / v_split, b % v_split, v_split, qkv, ld, M, (Hq + Hkv) * D, Hkv, row_slot, Vt, S);
    return;
  }
  qk_norm_rope_block<D>(b - v_blocks, qkv, ld, M, Hq, Hkv, qw_first, qw_second, kw_first, kw_second, split, cos_t, sin_t, row_cs,
                        row_slot, Q, Kws, S, eps, und_rounding);
}

extern "C" int bagel_qk_norm_rope_kv_bf16(const void* qkv, long ld, int M, int Hq, int Hkv, int head_dim,
                                          const void* q_norm_first, const void* q_norm_second,
                                          const void* k_norm_first, const void* k_norm_second, int split_row,
                                          const void* cos_table, const void* sin_table, const void* row_cs,
                                          const void* row_slot, void* Q, void* K_ws, void* Vt_ws, long S,
                                          float eps, int und_rounding, void* stream) {
  BAGEL_CHECK(qkv && q_norm_first && k_norm_first && cos_table && sin_table && row_cs && row_slot && Q && K_ws && Vt_ws,
              "bagel_qk_norm_rope_kv_bf16: null operand");
  BAGEL_CHECK(head_dim == 128 || head_dim == 64, "bagel_qk_norm_rope_kv_bf16: head_dim %d not supported", head_dim);
  BAGEL_CHECK(M > 0 && Hq > 0 && Hkv > 0, "bagel_qk_norm_rope_kv_bf16: bad shape");
  BAGEL_CHECK(ld % 8 == 0 && ld >= (long)(Hq + 2 * Hkv) * head_dim, "bagel_qk_norm_rope_kv_bf16: bad leading dim");
  BAGEL_CHECK(split_row >= M || (q_norm_second && k_norm_second), "bagel_qk_norm_rope_kv_bf16: rows past split_row need second weights");
  hipStream_t st = (hipStream_t)stream;
  const int lph = head_dim / 8;
  const long threads = (long)((M + 1) / 2) * (Hq + Hkv) * lph;   // two token rows per thread
  const int qk_blocks = (int)((threads + 255) / 256);
  // value scatter: up to 8 workgroups per 64-row block while that still leaves each wave a chunk (Hkv * D / 8 chunks, 4 waves)
  int v_split = 1;
  while (v_split < 8 && (long)((M + 63) / 64) * v_split < 4L * bagel_cu_count() && Hkv * head_dim / 8 >= 8 * v_split) v_split *= 2;
  const int v_blocks = (M + 63) / 64 * v_split;
  const dim3 grid(v_blocks + qk_blocks), block(256);
#define ARGS (const bf16*)qkv, ld, M, Hq, Hkv, (const bf16*)q_norm_first, (const bf16*)q_norm_second,                \
             (const bf16*)k_norm_first, (const bf16*)k_norm_second, split_row, (const bf16*)cos_table,               \
             (const bf16*)sin_table, (const int*)row_cs, (const int*)row_slot, (bf16*)Q, (bf16*)K_ws, S, eps, und_rounding, \
             (bf16*)Vt_ws, v_blocks, v_split
  if (head_dim == 128) hipLaunchKernelGGL(qk_norm_rope_kernel<128>, grid, block, 0, st, ARGS);
  else hipLaunchKernelGGL(qk_norm_rope_kernel<64>, grid, block, 0, st, ARGS);
#undef ARGS
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Latent embedding: X[r] = bf16( bf16(lin[src[r]] + temb) + pos_table[pos_id[src[r]]] )
// Bagel._forward_flow (bagel.py:1394-1399): vae2llm(x_t) + time_embedder(t) + latent_pos_embed, each sum
// rounded to bf16.  ``lin`` is the vae2llm projection (bias included) of one row per latent token;
// ``src`` lets several CFG branches of one sample share that row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_latent_kernel(const bf16* __restrict__ lin, long ldl,
                                                           const bf16* __restrict__ temb,
                                                           const bf16* __restrict__ pos_table,
                                                           const int* __restrict__ pos_id,
                                                           const int* __restrict__ src_row, bf16* __restrict__ X,
                                                           long ldx, int rows, int H) {
  const int cpr = H / 8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)rows * cpr) return;
  const int r = idx / cpr, c = (idx % cpr) * 8;
  const int s = src_row[r];
  const bf16x8 a = *(const bf16x8*)(lin + (long)s * ldl + c);
  const bf16x8 t = *(const bf16x8*)(temb + c);
  const bf16x8 pe = *(const bf16x8*)(pos_table + (long)pos_id[s] * H + c);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)(round_bf16((float)a[e] + (float)t[e]) + (float)pe[e]);
  *(bf16x8*)(X + (long)r * ldx + c) = o;
}

extern "C" int bagel_embed_latent_bf16(const void* lin, long ldl, const void* temb, const void* pos_table,
                                       const void* pos_id, const void* src_row, void* X, long ldx, int rows, int H,
                                       void* stream) {
  BAGEL_CHECK(lin && temb && pos_table && pos_id && src_row && X, "bagel_embed_latent_bf16: null operand");
  BAGEL_CHECK(rows > 0 && H % 8 == 0 && ldl % 8 == 0 && ldx % 8 == 0, "bagel_embed_latent_bf16: bad shape");
  const long n = (long)rows * (H / 8);
  hipLaunchKernelGGL(embed_latent_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)lin,
                     ldl, (const bf16*)temb, (const bf16*)pos_table, (const int*)pos_id, (const int*)src_row,
                     (bf16*)X, ldx, rows, H);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// out[r] = table[idx[r]]   (token embedding lookup, bagel.py:1389-1391; also replays saved rows)
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ table, long ldt,
                                                          const int* __restrict__ idx, bf16* __restrict__ out,
                                                          long ldo, int rows, int H) {
  const int cpr = H / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * cpr) return;
  const int r = i / cpr, c = (i % cpr) * 8;
  const long s = idx ? idx[r] : r;
  *(bf16x8*)(out + (long)r * ldo + c) = *(const bf16x8*)(table + s * ldt + c);
}

extern "C" int bagel_gather_rows_bf16(const void* table, long ldt, const void* idx, void* out, long ldo, int rows,
                                      int H, void* stream) {
  BAGEL_CHECK(table && out, "bagel_gather_rows_bf16: null operand");
  BAGEL_CHECK(rows > 0 && H % 8 == 0 && ldt % 8 == 0 && ldo % 8 == 0, "bagel_gather_rows_bf16: bad shape");
  const long n = (long)rows * (H / 8);
  hipLaunchKernelGGL(gather_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)table,
                     ldt, (const int*)idx, (bf16*)out, ldo, rows, H);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Sinusoidal timestep features for n timesteps: out[i] = [cos(t_i f_k) | sin(t_i f_k)], f_k = exp(-ln(1e4) k / half)
// TimestepEmbedder.timestep_embedding (modeling_utils.py:88-107), fp32 math, stored bf16 because the
// following nn.Linear runs on bf16 operands.
// ---------------------------------------------------------------------------------------------
__global__ void timestep_sinusoid_kernel(const float* __restrict__ t, int n, int dim, bf16* __restrict__ out) {
  const int i = blockIdx.x, half = dim / 2;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    const int k = c < half ? c : c - half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[i] * f;
    out[(long)i * dim + c] = (bf16)(c < half ? cosf(a) : sinf(a));
  }
}

extern "C" int bagel_timestep_sinusoid_bf16(const void* t, int n, int dim, void* out, void* stream) {
  BAGEL_CHECK(t && out && n > 0 && dim > 0 && dim % 2 == 0, "bagel_timestep_sinusoid_bf16: bad arguments");
  hipLaunchKernelGGL(timestep_sinusoid_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const float*)t, n, dim,
                     (bf16*)out);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *(const f32x4*)(x + i);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
    *(bf16x4*)(out + i) = o;
  } else {
    for (long k = i; k < n; ++k) out[k] = (bf16)x[k];
  }
}

extern "C" int bagel_cast_f32_bf16(const void* x, void* out, long n, void* stream) {
  BAGEL_CHECK(x && out && n > 0, "bagel_cast_f32_bf16: bad arguments");
  BAGEL_CHECK(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 8 == 0), "bagel_cast_f32_bf16: misaligned operand");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((n / 4 + 256) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const float*)x, (bf16*)out, n);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// CFG combine + renorm + Euler step.  Bagel._forward_flow (bagel.py:1466-1500) and the update
// x_t = x_t - v_t * dt (bagel.py:1339).  Every tensor op of the reference is a bf16 op whose result is
// rounded to bf16 (python scalars and 0-dim fp32 tensors enter as fp32 operands); torch.norm of a bf16
// tensor accumulates in fp32 and rounds the result.
//
//   renorm_type 0 = global (norm over every row of the packed batch), 1 = channel (per row),
//               2 = text_channel (per row, after the text stage only)
// Pass 1 (rows kernel) forms the combined velocity and per-row sums of squares; for channel/text_channel
// it also applies the row scale.  Pass 2 (one block) turns the row sums into the global scale.
// Pass 3 applies the global scale (if any) and the Euler update, and emits the bf16 copy of x_t that the
// next step's vae2llm projection reads.
// ---------------------------------------------------------------------------------------------
enum { RENORM_GLOBAL = 0, RENORM_CHANNEL = 1, RENORM_TEXT_CHANNEL = 2 };

__device__ __forceinline__ float cfg_mix(float base, float v, float scale) {
  // base + scale * (v - base) with a bf16 rounding after each op
  return round_bf16(base + round_bf16(scale * round_bf16(v - base)));
}

__device__ __forceinline__ float renorm_scale(float ss0, float ss1, float rmin) {
  const float n0 = round_bf16(sqrtf(ss0)), n1 = round_bf16(sqrtf(ss1));
  const float q = round_bf16(n0 / round_bf16(n1 + 1e-8f));
  return round_bf16(fminf(fmaxf(q, rmin), 1.0f));
}

template <int P>
__global__ __launch_bounds__(256) void cfg_rows_kernel(const bf16* __restrict__ v_cond, const bf16* __restrict__ v_text,
                                                       const bf16* __restrict__ v_img, float s_text, float s_img,
                                                       int renorm_type, float rmin, bf16* __restrict__ v_out,
                                                       float* __restrict__ row_ss, int rows) {
  constexpr int LPR = P / 8;
  const long item = ((long)blockIdx.x * 256 + threadIdx.x) / LPR;
  if (item >= rows) return;
  const int j = threadIdx.x % LPR;
  const long off = item * P + 8 * j;
  const bf16x8 c = *(const bf16x8*)(v_cond + off);
  const bf16x8 t = *(const bf16x8*)(v_text + off);
  bf16x8 im;
  const bool use_img = s_img > 1.0f;
  if (use_img) im = *(const bf16x8*)(v_img + off);
  float vc[8], a[8];
  float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    vc[e] = (float)c[e];
    a[e] = cfg_mix((float)t[e], vc[e], s_text);  // v_t_text_
    ss0 += vc[e] * vc[e];
  }
  if (renorm_type == RENORM_TEXT_CHANNEL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ss1 += a[e] * a[e];
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      ss0 += __shfl_xor(ss0, o, LPR);
      ss1 += __shfl_xor(ss1, o, LPR);
    }
    const float sc = renorm_scale(ss0, ss1, rmin);
    bf16x8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = round_bf16(a[e] * sc);
      if (use_img) v = cfg_mix((float)im[e], v, s_img);
      o8[e] = (bf16)v;
    }
    *(bf16x8*)(v_out + off) = o8;
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (use_img) a[e] = cfg_mix((float)im[e], a[e], s_img);
    ss1 += a[e] * a[e];
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) {
    ss0 += __shfl_xor(ss0, o, LPR);
    ss1 += __shfl_xor(ss1, o, LPR);
  }
  bf16x8 o8;
  if (renorm_type == RENORM_CHANNEL) {
    const float sc = renorm_scale(ss0, ss1, rmin);
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (bf16)(a[e] * sc);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (bf16)a[e];
    if (j == 0) {
      row_ss[item] = ss0;
      row_ss[rows + item] = ss1;
    }
  }
  *(bf16x8*)(v_out + off) = o8;
}

// one block: sums the per-row sums of squares in a fixed order and writes the global scale
__global__ __launch_bounds__(1024) void cfg_global_scale_kernel(const float* __restrict__ row_ss, int rows, float rmin,
                                                                 float* __restrict__ scale_out) {
  __shared__ float sh[2][16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < rows; i += 1024) {
    a += row_ss[i];
    b += row_ss[rows + i];
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) {
    sh[0][threadIdx.x >> 6] = a;
    sh[1][threadIdx.x >> 6] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int w = 0; w < 16; ++w) {
      s0 += sh[0][w];
      s1 += sh[1][w];
    }
    scale_out[0] = renorm_scale(s0, s1, rmin);
  }
}

// x -= float(bf16(v * dt)), v optionally scaled by *scale first; writes v_t (bf16) and the bf16 copy of x
__global__ __launch_bounds__(256) void euler_update_kernel(const bf16* __restrict__ v_in, const float* __restrict__ scale,
                                                           float dt, float* __restrict__ x, bf16* __restrict__ x_bf16,
                                                           bf16* __restrict__ v_out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  const bf16x8 v = *(const bf16x8*)(v_in + i);
  const float sc = scale ? scale[0] : 1.0f;
  const f32x4 x0 = *(const f32x4*)(x + i), x1 = *(const f32x4*)(x + i + 4);
  float xs[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    xs[e] = x0[e];
    xs[e + 4] = x1[e];
  }
  bf16x8 vo, xb;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float ve = (float)v[e];
    if (scale) ve = round_bf16(ve * sc);
    vo[e] = (bf16)ve;
    xs[e] = xs[e] - round_bf16(ve * dt);
    xb[e] = (bf16)xs[e];
  }
  *(f32x4*)(x + i) = f32x4{xs[0], xs[1], xs[2], xs[3]};
  *(f32x4*)(x + i + 4) = f32x4{xs[4], xs[5], xs[6], xs[7]};
  *(bf16x8*)(x_bf16 + i) = xb;
  if (v_out) *(bf16x8*)(v_out + i) = vo;
}

extern "C" long bagel_cfg_workspace_bytes(int rows, int P) { return ((long)rows * P * 2 + 2L * rows * 4 + 256 + 255) / 256 * 256; }

// Shared by both entry points below; `workspace` holds bagel_cfg_workspace_bytes(rows, P) bytes.  Every word of the
// workspace that is read has been written earlier in the same call, so nothing has to start at zero.
static int cfg_euler_launch(const char* who, const void* v_cond, const void* v_text, const void* v_img, float cfg_text_scale,
                            float cfg_img_scale, int renorm_type, float renorm_min, float dt, void* x_t, void* x_bf16,
                            void* v_t_out, int rows, int P, void* workspace, void* stream) {
  BAGEL_CHECK(v_cond && x_t && x_bf16 && workspace, "%s: null operand", who);
  BAGEL_CHECK(rows > 0 && P == 64, "%s: rows=%d P=%d (P must be 64)", who, rows, P);
  BAGEL_CHECK(renorm_type >= 0 && renorm_type <= 2, "%s: cfg_renorm_type %d is not supported", who, renorm_type);
  const bool cfg = cfg_text_scale > 1.0f;  // bagel.py:1466: image CFG only ever acts together with text CFG
  if (cfg) {
    BAGEL_CHECK(v_text, "%s: cfg_text_scale > 1 needs v_text", who);
    BAGEL_CHECK(!(cfg_img_scale > 1.0f) || v_img, "%s: cfg_img_scale > 1 needs v_img", who);
  }
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)rows * P;
  bf16* v_pre = (bf16*)workspace;
  float* row_ss = (float*)((char*)workspace + ((n * 2 + 255) / 256) * 256);
  float* scale = row_ss + 2L * rows;
  const bf16* v_final = (const bf16*)v_cond;
  const float* scale_ptr = nullptr;
  if (cfg) {
    const long threads = (long)rows * (P / 8);
    hipLaunchKernelGGL(cfg_rows_kernel<64>, dim3((threads + 255) / 256), dim3(256), 0, st, (const bf16*)v_cond,
                       (const bf16*)v_text, (const bf16*)v_img, cfg_text_scale, cfg_img_scale, renorm_type, renorm_min,
                       v_pre, row_ss, rows);
    v_final = v_pre;
    if (renorm_type == RENORM_GLOBAL) {
      hipLaunchKernelGGL(cfg_global_scale_kernel, dim3(1), dim3(1024), 0, st, row_ss, rows, renorm_min, scale);
      scale_ptr = scale;
    }
  }
  hipLaunchKernelGGL(euler_update_kernel, dim3((n / 8 + 255) / 256), dim3(256), 0, st, v_final, scale_ptr, dt,
                     (float*)x_t, (bf16*)x_bf16, (bf16*)v_t_out, n);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

extern "C" int bagel_cfg_euler_update(const void* v_cond, const void* v_text, const void* v_img, float cfg_text_scale,
                                      float cfg_img_scale, int renorm_type, float renorm_min, float dt, void* x_t,
                                      void* x_bf16, void* v_t_out, int rows, int P, void* workspace,
                                      long workspace_bytes, void* stream) {
  BAGEL_CHECK(workspace, "bagel_cfg_euler_update: null operand");
  BAGEL_CHECK(rows > 0 && P == 64, "bagel_cfg_euler_update: rows=%d P=%d (P must be 64)", rows, P);
  BAGEL_CHECK(workspace_bytes >= bagel_cfg_workspace_bytes(rows, P), "bagel_cfg_euler_update: workspace too small");
  return cfg_euler_launch("bagel_cfg_euler_update", v_cond, v_text, v_img, cfg_text_scale, cfg_img_scale, renorm_type,
                          renorm_min, dt, x_t, x_bf16, v_t_out, rows, P, workspace, stream);
}

// ---------------------------------------------------------------------------------------------
// The same update for callers that cannot allocate (a step loop under HIP graph capture, a serving loop that
// forbids hipMalloc): the pre-renorm velocity, the per-row sums of squares and the global scale live in a
// file-scope device array instead of a caller workspace.  One array per GPU, held until the process ends;
// its device address is looked up once per device and only that address is ever handed to kernels.
// ---------------------------------------------------------------------------------------------
constexpr int CFG_STATIC_MAX_ROWS = 65536;  // 16 images of 1024 x 1024 (4096 latent tokens each)
constexpr long CFG_STATIC_BYTES = ((long)CFG_STATIC_MAX_ROWS * 64 * 2 + 2L * CFG_STATIC_MAX_ROWS * 4 + 256 + 255) / 256 * 256;
static __device__ __align__(256) unsigned char scratch[CFG_STATIC_BYTES];

#include <mutex>
static int cfg_static_scratch(void** out) {
  static std::mutex mu;
  static void* ptr_of_device[64] = {nullptr};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= 64) {
    bagel_set_error("bagel_cfg_euler_update_static: hipGetDevice failed or device ordinal %d beyond 63: %s", dev,
                    hipGetErrorString(e));
    return 2;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (ptr_of_device[dev] == nullptr) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));   // this device's copy of the array (dev is current)
    if (e != hipSuccess || ptr == nullptr) {
      bagel_set_error("bagel_cfg_euler_update_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
      return 2;
    }
    ptr_of_device[dev] = ptr;
  }
  *out = ptr_of_device[dev];
  return 0;
}

extern "C" int bagel_cfg_static_max_rows(void) { return CFG_STATIC_MAX_ROWS; }

extern "C" int bagel_cfg_euler_update_static(const void* v_cond, const void* v_text, const void* v_img, float cfg_text_scale,
                                             float cfg_img_scale, int renorm_type, float renorm_min, float dt, void* x_t,
                                             void* x_bf16, void* v_t_out, int rows, int P, void* stream) {
  BAGEL_CHECK(rows > 0 && P == 64, "bagel_cfg_euler_update_static: rows=%d P=%d (P must be 64)", rows, P);
  BAGEL_CHECK(rows <= CFG_STATIC_MAX_ROWS, "bagel_cfg_euler_update_static: rows=%d exceeds the built-in scratch (%d rows); "
              "use bagel_cfg_euler_update with a caller workspace", rows, CFG_STATIC_MAX_ROWS);
  void* ws = nullptr;
  if (int rc = cfg_static_scratch(&ws)) return rc;
  return cfg_euler_launch("bagel_cfg_euler_update_static", v_cond, v_text, v_img, cfg_text_scale, cfg_img_scale, renorm_type,
                          renorm_min, dt, x_t, x_bf16, v_t_out, rows, P, ws, stream);
}

// ---------------------------------------------------------------------------------------------
// Greedy token choice: out[r] = index of the first maximum of logits[r] (torch.argmax, bagel.py:1580).
// One block per row; ties resolve to the lowest index, NaNs are never selected over numbers.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const bf16* __restrict__ logits, long ld, int n,
                                                           int* __restrict__ out) {
  const bf16* row = logits + (long)blockIdx.x * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  auto take = [&](float v, int i) {
    if (v > best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  };
  // 16-byte loads where the row allows it (a vocabulary row is ~300 KB), four of them in flight per thread; scalar
  // loads for the ragged rest
  const bool vec = (ld % 8 == 0) && ((uintptr_t)logits % 16 == 0);
  const int nvec = vec ? n / 8 : 0;
  for (int c0 = threadIdx.x; c0 < nvec; c0 += 4096) {
    bf16x8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + 1024 * u < nvec) v[u] = *(const bf16x8*)(row + 8 * (c0 + 1024 * u));
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (c0 + 1024 * u < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) take((float)v[u][e], 8 * (c0 + 1024 * u) + e);
      }
  }
  for (int i = nvec * 8 + threadIdx.x; i < n; i += 1024) take((float)row[i], i);
  // wave reduce, then the 16 wave winners through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v = __shfl_xor(best, o, 64);
    const int i = __shfl_xor(bi, o, 64);
    take(v, i);
  }
  __shared__ float sv[16];
  __shared__ int si[16];
  if ((threadIdx.x & 63) == 0) {
    sv[threadIdx.x >> 6] = best;
    si[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) take(sv[w], si[w]);
    out[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
  }
}

extern "C" int bagel_argmax_rows_bf16(const void* logits, long ld, int rows, int n, void* out_i32, void* stream) {
  BAGEL_CHECK(logits && out_i32 && rows > 0 && n > 0, "bagel_argmax_rows_bf16: bad arguments");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, (const bf16*)logits, ld, n,
                     (int*)out_i32);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// nn.LayerNorm (siglip_navit.py:267-269, 343): fp32 statistics and affine on a bf16 row, rounded to bf16
// once (the consumer is always a bf16 nn.Linear).  One wave per row, values kept in registers.
// ---------------------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16* __restrict__ x, long ldx, bf16* __restrict__ out,
                                                        long ldo, int rows, int H, const bf16* __restrict__ gamma,
                                                        const bf16* __restrict__ beta, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int nchunk = H / 8;
  float v[MAXC][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const bf16x8 t = *(const bf16x8*)(x + (long)row * ldx + 8 * c);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (float)t[e];
        sum += v[i][e];
      }
    }
  }
  const float mean = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (lane + 64 * i < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e] - mean;
        sq += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nchunk) {
      const bf16x8 g = *(const bf16x8*)(gamma + 8 * c);
      const bf16x8 b = *(const bf16x8*)(beta + 8 * c);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16)((v[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
      *(bf16x8*)(out + (long)row * ldo + 8 * c) = o;
    }
  }
}

extern "C" int bagel_layernorm_bf16(const void* x, long ldx, void* out, long ldo, int rows, int H, const void* gamma,
                                    const void* beta, float eps, void* stream) {
  BAGEL_CHECK(x && out && gamma && beta, "bagel_layernorm_bf16: null operand");
  BAGEL_CHECK(rows > 0 && H > 0 && H % 8 == 0 && H <= 4096 && ldx % 8 == 0 && ldo % 8 == 0 && ldx >= H && ldo >= H,
              "bagel_layernorm_bf16: bad shape rows=%d H=%d", rows, H);
  const dim3 grid((rows + 3) / 4), block(256);
#define ARGS (const bf16*)x, ldx, (bf16*)out, ldo, rows, H, (const bf16*)gamma, (const bf16*)beta, eps
  if (H <= 1536)
    hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, (hipStream_t)stream, ARGS);
  else
    hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, (hipStream_t)stream, ARGS);
#undef ARGS
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Plain multi-head attention inputs (no q/k norm, no RoPE: SiglipFlashAttention2 with rope off,
// siglip_navit.py:216-243) for the attention kernel: qkv [M][3][heads][Din] ->
//   Q [M][heads][D] , K workspace [heads][S][D] at slot row_slot[row], Vt workspace [heads][D][S],
// with Din (72 for SigLIP so400m) zero-padded to the kernel's head_dim D.  Zero padding leaves q.k and P.V
// unchanged; the pad rows of Vt must be zero, which the caller guarantees by zeroing the workspace once.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_heads_pad_kernel(const bf16* __restrict__ qkv, long ld, int M, int heads,
                                                              int Din, int D, const int* __restrict__ row_slot,
                                                              bf16* __restrict__ Q, bf16* __restrict__ Kws,
                                                              bf16* __restrict__ Vt, long S) {
  const int cin = Din / 8, cpd = D / 8;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per_row = (long)heads * cpd;
  if (idx >= (long)M * per_row) return;
  const int row = idx / per_row;
  const int rem = idx % per_row;
  const int head = rem / cpd, c = rem % cpd;
  const long slot = row_slot[row];
  const bf16* src = qkv + (long)row * ld + (long)head * Din + 8 * c;
  bf16x8 q, k;
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = k[e] = (bf16)0.f;
  if (c < cin) {
    q = *(const bf16x8*)src;
    k = *(const bf16x8*)(src + (long)heads * Din);
    const bf16x8 v = *(const bf16x8*)(src + 2L * heads * Din);
    bf16* dst = Vt + ((long)head * D + 8 * c) * S + slot;
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[(long)e * S] = v[e];
  }
  *(bf16x8*)(Q + ((long)row * heads + head) * D + 8 * c) = q;
  *(bf16x8*)(Kws + ((long)head * S + slot) * D + 8 * c) = k;
}

extern "C" int bagel_split_heads_pad_bf16(const void* qkv, long ld, int M, int heads, int head_dim_in, int head_dim,
                                          const void* row_slot, void* Q, void* K_ws, void* Vt_ws, long S, void* stream) {
  BAGEL_CHECK(qkv && row_slot && Q && K_ws && Vt_ws, "bagel_split_heads_pad_bf16: null operand");
  BAGEL_CHECK(M > 0 && heads > 0 && head_dim_in % 8 == 0 && head_dim % 8 == 0 && head_dim_in <= head_dim,
              "bagel_split_heads_pad_bf16: bad head dims %d -> %d", head_dim_in, head_dim);
  BAGEL_CHECK(ld % 8 == 0 && ld >= 3L * heads * head_dim_in, "bagel_split_heads_pad_bf16: bad leading dim");
  const long n = (long)M * heads * (head_dim / 8);
  hipLaunchKernelGGL(split_heads_pad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)qkv, ld, M, heads, head_dim_in, head_dim, (const int*)row_slot, (bf16*)Q, (bf16*)K_ws,
                     (bf16*)Vt_ws, S);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// TaylorSeer feature cache (modeling/cache_utils/taylorseer.py), elementwise over n bf16 values.
// F holds the factors [order][factor_stride >= n].  Every tensor op of the reference rounds to bf16; so do these.
// update (derivative_approximation, :11-29): new_0 = feature, new_{i+1} = bf16(bf16(new_i - old_i) / distance)
//   for i < n_new - 1, where the caller decides n_new (at most n_old + 1 and at most 7).
// eval (taylor_formula, :31-45): out = sum_i bf16(bf16(float(1 / i!) * F_i) * x^i), added in bf16.
// ---------------------------------------------------------------------------------------------
constexpr int TAYLOR_MAX = 7;

__global__ __launch_bounds__(256) void taylor_update_kernel(const bf16* __restrict__ feature, long ldf, int H,
                                                            bf16* __restrict__ F, long fs, long n, int n_new, float distance) {
  const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (idx >= n) return;
  const long row = idx / H;
  const int col = idx - row * H;
  bf16x8 cur = *(const bf16x8*)(feature + row * ldf + col);
#pragma unroll
  for (int i = 0; i < TAYLOR_MAX; ++i) {
    if (i >= n_new) break;
    bf16x8 nxt;
    if (i + 1 < n_new) {
      const bf16x8 old = *(const bf16x8*)(F + (long)i * fs + idx);
#pragma unroll
      for (int e = 0; e < 8; ++e) nxt[e] = (bf16)(round_bf16((float)cur[e] - (float)old[e]) / distance);
    }
    *(bf16x8*)(F + (long)i * fs + idx) = cur;
    cur = nxt;
  }
}

__global__ __launch_bounds__(256) void taylor_eval_kernel(const bf16* __restrict__ F, long fs, long n, int n_factors, int x,
                                                          bf16* __restrict__ out, long ldo, int H) {
  const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (idx >= n) return;
  const long row = idx / H;
  const int col = idx - row * H;
  bf16x8 acc = *(const bf16x8*)(F + idx);   // order 0: 1/0! * F_0 * x^0
  float coef = 1.f, xp = 1.f;
  double fact = 1.0;
#pragma unroll
  for (int i = 1; i < TAYLOR_MAX; ++i) {
    if (i >= n_factors) break;
    fact *= (double)i;
    coef = (float)(1.0 / fact);
    xp *= (float)x;
    const bf16x8 f = *(const bf16x8*)(F + (long)i * fs + idx);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (bf16)((float)acc[e] + round_bf16(round_bf16(coef * (float)f[e]) * xp));
  }
  *(bf16x8*)(out + row * ldo + col) = acc;
}

extern "C" int bagel_taylor_update_bf16(const void* feature, long ldf, int rows, int H, void* factors,
                                        long factor_stride, int n_new, int distance, void* stream) {
  BAGEL_CHECK(feature && factors && rows > 0 && H > 0 && H % 8 == 0 && ldf % 8 == 0 && ldf >= H,
              "bagel_taylor_update_bf16: bad arguments");
  BAGEL_CHECK(n_new >= 1 && n_new <= TAYLOR_MAX && (n_new == 1 || distance > 0), "bagel_taylor_update_bf16: bad order %d", n_new);
  const long n = (long)rows * H;
  BAGEL_CHECK(factor_stride >= n && factor_stride % 8 == 0, "bagel_taylor_update_bf16: bad factor stride");
  hipLaunchKernelGGL(taylor_update_kernel, dim3((n / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)feature, ldf, H, (bf16*)factors, factor_stride, n, n_new, (float)distance);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

extern "C" int bagel_taylor_eval_bf16(const void* factors, long factor_stride, int rows, int H, int n_factors, int x,
                                      void* out, long ldo, void* stream) {
  BAGEL_CHECK(factors && out && rows > 0 && H > 0 && H % 8 == 0 && ldo % 8 == 0 && ldo >= H,
              "bagel_taylor_eval_bf16: bad arguments");
  BAGEL_CHECK(n_factors >= 1 && n_factors <= TAYLOR_MAX && x >= 0, "bagel_taylor_eval_bf16: bad order %d", n_factors);
  const long n = (long)rows * H;
  BAGEL_CHECK(factor_stride >= n && factor_stride % 8 == 0, "bagel_taylor_eval_bf16: bad factor stride");
  hipLaunchKernelGGL(taylor_eval_kernel, dim3((n / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)factors, factor_stride, n, n_factors, x, (bf16*)out, ldo, H);
  BAGEL_LAUNCH_CHECK();
  return 0;
}
