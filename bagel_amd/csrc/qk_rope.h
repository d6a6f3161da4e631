// Per-head q/k RMSNorm followed by rotary embedding, as one lane's share of a head.
//
// PackedAttentionMoT.forward_inference (qwen2_navit.py:549-566): the norm is computed in fp32 from the bf16
// projection output; with und_rounding the norm output and every RoPE product are rounded to bf16 (the
// understanding-mode graph, :526-530), otherwise only the final sum is.
// D/8 lanes hold one head; lane j holds elements 8j..8j+7 and its rotate_half partner is lane j ^ (D/16).  All D/8
// lanes of a head must be active together.
#pragma once
#include "common.h"

template <int D>
__device__ __forceinline__ bf16x8 qk_norm_rope_lane(const bf16x8 v, int j, const bf16* __restrict__ w,
                                                    const bf16* __restrict__ cos_row, const bf16* __restrict__ sin_row,
                                                    float eps, int und_rounding) {
  constexpr int LPH = D / 8;
  float x[8];
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    x[e] = (float)v[e];
    ss += x[e] * x[e];
  }
#pragma unroll
  for (int o = LPH / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, LPH);
  const float rs = rsqrtf(ss / (float)D + eps);
  const bf16x8 wv = *(const bf16x8*)(w + 8 * j);
  const bf16x8 cv = *(const bf16x8*)(cos_row + 8 * j);
  const bf16x8 sv = *(const bf16x8*)(sin_row + 8 * j);
  float y[8], yr[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float n = x[e] * rs;
    y[e] = und_rounding ? round_bf16((float)wv[e] * round_bf16(n)) : (float)wv[e] * n;
  }
  const float sign = j < LPH / 2 ? -1.f : 1.f;  // rotate_half: first half gets -x2, second half +x1
#pragma unroll
  for (int e = 0; e < 8; ++e) yr[e] = sign * __shfl_xor(y[e], LPH / 2, LPH);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float c = (float)cv[e], s = (float)sv[e];
    if (und_rounding)
      o[e] = (bf16)(round_bf16(y[e] * c) + round_bf16(yr[e] * s));
    else
      o[e] = (bf16)(__fmul_rn(y[e], c) + __fmul_rn(yr[e], s));
  }
  return o;
}
