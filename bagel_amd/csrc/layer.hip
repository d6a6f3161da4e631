// One Mixture-of-Transformers decoder layer, and the whole stack, as single C calls.
//
// Qwen2MoTDecoderLayer.forward_inference (qwen2_navit.py:771-845) around PackedAttentionMoT.forward_inference
// (:510-611), and the layer loop + final norm of Qwen2Model.forward_inference (:1079-1105), for a packed
// sequence whose rows are sorted by expert: rows [0, rows_gen) use the generation expert (``*_moe_gen``
// parameters, MFMA tile GEMMs), rows [rows_gen, rows_gen + rows_und) the understanding expert (weight-streaming
// GEMM for at most 64 rows, tile GEMM beyond).  This file only sequences the kernels of the other sources on
// the caller's stream; it allocates nothing and holds no state besides the optional event recorder below.
#include "common.h"
#include "../../include/bagel_hip.h"
#include <mutex>
#include <vector>

namespace {

// ---- optional timing of the generation expert's gate+up GEMM (the dominant kernel), for the benchmark's
// roofline figure: HIP events on the launch stream around each such launch while recording is on.
struct TimedLaunch {
  hipEvent_t start, stop;
  double flops;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<TimedLaunch> g_prof;

inline char* at(void* base, long row, long width) { return (char*)base + row * width * 2; }
inline const char* at(const void* base, long row, long width) { return (const char*)base + row * width * 2; }

// One projection of the understanding expert on rows [r0, r0 + n) of `in` -> `out`.
int und_linear(const void* in, long ld_in, const void* W, void* out, long ld_out, int n, int N, int K, const void* bias,
               const void* residual, long ldr, int kind /* 0 bias, 1 residual, 2 swiglu */, const bagel_mot_rows* r,
               void* stream) {
  // Up to 64 rows stream the weights once, with bits that do not depend on the row count (the text rows of a packed
  // generation step rely on it).  A forward without generation rows (a prompt's prefill, a decode step) of more than 16
  // rows takes the tile kernels, which share every token-row fragment through LDS: 186 against 282 us per layer at 64
  // rows; its bits then depend on the row class (17 ... 512, 513 ... 1024, more), not on the row count inside it.
  const bool decode_tiles = r->rows_gen == 0 && n > 16;
  const int sk_ep = kind == 0 ? BAGEL_SK_BIAS : kind == 1 ? BAGEL_SK_RESIDUAL : BAGEL_SK_SWIGLU;
  if (n <= 64 && !decode_tiles)
    return bagel_skinny_gemm_bf16(in, ld_in, W, K, out, ld_out, n, N, K, bias, residual, ldr, sk_ep, r->skinny_ws,
                                  r->skinny_ws_bytes, stream);
  if (r->rows_gen > 0) {
    // Text rows beside generation rows (two per (sample, CFG branch) unit, so more than 32 units in one forward):
    // a unit's bits must not depend on how many other units share the launch - the same units split over ranks, or
    // run alone, stay at 64 rows or fewer - so the rows keep the weight-streaming kernel, 64 at a time.
    for (int r0 = 0; r0 < n; r0 += 64) {
      const int m = n - r0 < 64 ? n - r0 : 64;
      if (int rc = bagel_skinny_gemm_bf16(at(in, r0, ld_in), ld_in, W, K, at(out, r0, ld_out), ld_out, m, N, K, bias,
                                          residual ? at(residual, r0, ldr) : nullptr, ldr, sk_ep, r->skinny_ws,
                                          r->skinny_ws_bytes, stream))
        return rc;
    }
    return 0;
  }
  const int ep = kind == 0 ? BAGEL_EPI_BIAS : kind == 1 ? BAGEL_EPI_RESIDUAL : BAGEL_EPI_SWIGLU;
  // 17 ... 1024 rows (a text prompt's prefill, a large decode batch) split the reduction of the narrow projections over the chip
  return bagel_gemm_ws_bf16(in, ld_in, W, K, out, ld_out, n, N, K, bias, residual, ldr, ep, 1.0f, r->skinny_ws, r->skinny_ws_bytes,
                            stream);
}

}  // namespace

extern "C" void bagel_profile_gate_up(int on) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  g_prof_on = on != 0;
}

// Waits for the recorded launches, writes up to `cap` (milliseconds, flops) pairs and forgets them; returns the count.
extern "C" int bagel_profile_collect(double* ms, double* flops, int cap) {
  std::lock_guard<std::mutex> lock(g_prof_mu);
  int n = 0;
  for (auto& t : g_prof) {
    float m = 0.f;
    if (t.flops > 0 && hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&m, t.start, t.stop) == hipSuccess && n < cap) {
      ms[n] = m;
      flops[n] = t.flops;
      ++n;
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  g_prof.clear();
  return n;
}

#define TRY(call)              \
  do {                         \
    if (int rc_ = (call)) return rc_; \
  } while (0)

extern "C" int bagel_mot_layer_forward(const bagel_mot_dims* d, const bagel_mot_layer* w, const bagel_mot_rows* r,
                                       void* K_ws, void* Vt_ws, int flags, const void* next_norm_gen,
                                       const void* next_norm_und, void* stream) {
  BAGEL_CHECK(d && w && r && K_ws && Vt_ws, "bagel_mot_layer_forward: null argument");
  const int Mv = r->rows_gen, Mt = r->rows_und, M = Mv + Mt;
  BAGEL_CHECK(Mv >= 0 && Mt >= 0 && M > 0, "bagel_mot_layer_forward: rows_gen=%d rows_und=%d", Mv, Mt);
  BAGEL_CHECK(r->X && r->Xn && r->QKV && r->Q, "bagel_mot_layer_forward: null activation buffer");
  const int H = d->hidden, I = d->intermediate, Hq = d->num_heads, Hkv = d->num_kv_heads, D = d->head_dim;
  const int qd = Hq * D, qkv = (Hq + 2 * Hkv) * D;
  const float eps = d->rms_eps;
  const bagel_mot_expert &g = w->gen, &u = w->und;
  // per-modality weight pairs for the row-split kernels: rows below rows_gen take the first
  const void* ln_in_a = Mv ? g.ln_in : u.ln_in;
  const void* ln_post_a = Mv ? g.ln_post : u.ln_post;
  const void* qn_a = Mv ? g.q_norm : u.q_norm;
  const void* kn_a = Mv ? g.k_norm : u.k_norm;
  const int split = Mv ? Mv : M;
  // few understanding rows and nothing else: reduce + residual + the following RMSNorm share one launch
  const bool fuse_norm = Mv == 0 && Mt <= 16 && H <= 8192;   // und_linear() explains the exception
  // the tile route of und_linear(): o_proj / down_proj share their reduce launch with the RMSNorm that follows
  const bool tile_rows = Mv == 0 && Mt > 16 && H <= 8192;
  void* X_u = at(r->X, Mv, H);
  void* Xn_u = at(r->Xn, Mv, H);
  // last layer of a generation forward: the text rows' keys / values are still needed (the latent rows attend to
  // them), what the layer would add to their hidden state is not
  const bool drop_und_out = (flags & BAGEL_LAYER_UND_OUT_UNUSED) && Mv > 0;

  if (flags & BAGEL_LAYER_NORM_IN_GEMM) {
    // understanding rows only, a handful of them (a decode step): both RMSNorms are applied inside the
    // weight-streaming kernels that consume them, so the layer is seven launches (five with the one-launch decode attention) and Xn is never written
    BAGEL_CHECK(Mv == 0 && Mt <= 64, "bagel_mot_layer_forward: BAGEL_LAYER_NORM_IN_GEMM needs rows_gen == 0 and rows_und <= 64");
    TRY(bagel_skinny_gemm_rmsnorm_in_bf16(r->X, H, u.ln_in, eps, u.wqkv, H, r->QKV, qkv, Mt, qkv, H, u.bqkv, BAGEL_SK_BIAS, r->Xn,
                                          H, r->skinny_ws, r->skinny_ws_bytes, stream));
    const bool one_launch = r->decode_ws && !(flags & BAGEL_LAYER_KV_ONLY) && r->n_tiles == M &&
                            bagel_decode_attention_supported(M, Hq, Hkv, D);
    if (one_launch) {
      BAGEL_CHECK(r->O && r->Hm && r->tiles, "bagel_mot_layer_forward: null attention / MLP buffer");
      TRY(bagel_decode_attention_bf16(r->QKV, qkv, M, Hq, Hkv, D, u.q_norm, u.k_norm, r->cos_table, r->sin_table, r->row_cs,
                                      r->row_slot, r->tiles, K_ws, Vt_ws, r->S, r->O, eps, r->und_rounding, r->decode_max_kv, r->decode_ws,
                                      r->decode_ws_bytes, stream));
    } else {
      TRY(bagel_qk_norm_rope_kv_bf16(r->QKV, qkv, M, Hq, Hkv, D, u.q_norm, u.q_norm, u.k_norm, u.k_norm, M, r->cos_table,
                                     r->sin_table, r->row_cs, r->row_slot, r->Q, K_ws, Vt_ws, r->S, eps, r->und_rounding, stream));
      if (flags & BAGEL_LAYER_KV_ONLY) return 0;
      BAGEL_CHECK(r->O && r->Hm && r->tiles && r->n_tiles > 0, "bagel_mot_layer_forward: null attention / MLP buffer");
      TRY(bagel_varlen_attention_bf16(r->Q, K_ws, Vt_ws, r->O, r->tiles, r->n_tiles, Hq, Hkv, D, r->S, stream));
    }
    TRY(bagel_skinny_gemv_bf16(r->O, qd, u.wo, qd, r->X, H, Mt, H, qd, nullptr, r->X, H, BAGEL_SK_RESIDUAL, r->skinny_ws,
                               r->skinny_ws_bytes, stream));
    TRY(bagel_skinny_gemm_rmsnorm_in_bf16(r->X, H, u.ln_post, eps, u.wgu, H, r->Hm, I, Mt, 2 * I, H, nullptr, BAGEL_SK_SWIGLU,
                                          r->Xn, H, r->skinny_ws, r->skinny_ws_bytes, stream));
    TRY(bagel_skinny_gemv_bf16(r->Hm, I, u.wd, I, r->X, H, Mt, H, I, nullptr, r->X, H, BAGEL_SK_RESIDUAL, r->skinny_ws,
                               r->skinny_ws_bytes, stream));
    if (next_norm_und) TRY(bagel_rmsnorm_bf16(r->X, H, r->Xn, H, M, H, next_norm_und, next_norm_und, M, eps, stream));
    return 0;
  }
  if (!(flags & BAGEL_LAYER_XN_READY)) TRY(bagel_rmsnorm_bf16(r->X, H, r->Xn, H, M, H, ln_in_a, u.ln_in, split, eps, stream));
  if (Mv) TRY(bagel_gemm_bf16(r->Xn, H, g.wqkv, H, r->QKV, qkv, Mv, qkv, H, g.bqkv, nullptr, 0, BAGEL_EPI_BIAS, 1.0f, stream));
  if (Mt) TRY(und_linear(Xn_u, H, u.wqkv, at(r->QKV, Mv, qkv), qkv, Mt, qkv, H, u.bqkv, nullptr, 0, 0, r, stream));
  // a decode step of more than four samples (the caller's decode workspace says every row is one sample's new token)
  const bool decode_rows = Mv == 0 && r->decode_ws && !(flags & BAGEL_LAYER_KV_ONLY) && r->n_tiles == M &&
                           bagel_decode_attention_supported(M, Hq, Hkv, D);
  if (decode_rows) {
    BAGEL_CHECK(r->O && r->Hm && r->tiles, "bagel_mot_layer_forward: null attention / MLP buffer");
    TRY(bagel_decode_attention_bf16(r->QKV, qkv, M, Hq, Hkv, D, u.q_norm, u.k_norm, r->cos_table, r->sin_table, r->row_cs,
                                    r->row_slot, r->tiles, K_ws, Vt_ws, r->S, r->O, eps, r->und_rounding, r->decode_max_kv, r->decode_ws,
                                    r->decode_ws_bytes, stream));
  } else {
    TRY(bagel_qk_norm_rope_kv_bf16(r->QKV, qkv, M, Hq, Hkv, D, qn_a, u.q_norm, kn_a, u.k_norm, split, r->cos_table, r->sin_table,
                                   r->row_cs, r->row_slot, r->Q, K_ws, Vt_ws, r->S, eps, r->und_rounding, stream));
    if (flags & BAGEL_LAYER_KV_ONLY) return 0;
    BAGEL_CHECK(r->O && r->Hm && r->tiles && r->n_tiles > 0, "bagel_mot_layer_forward: null attention / MLP buffer");
    TRY(bagel_varlen_attention_bf16(r->Q, K_ws, Vt_ws, r->O, r->tiles, r->n_tiles, Hq, Hkv, D, r->S, stream));
  }
  if (Mv) TRY(bagel_gemm_bf16(r->O, qd, g.wo, qd, r->X, H, Mv, H, qd, nullptr, r->X, H, BAGEL_EPI_RESIDUAL, 1.0f, stream));
  if (fuse_norm) {
    TRY(bagel_skinny_gemm_residual_rmsnorm_bf16(r->O, qd, u.wo, qd, r->X, H, Mt, H, qd, r->X, H, u.ln_post, r->Xn, H, eps,
                                                r->skinny_ws, r->skinny_ws_bytes, stream));
  } else if (tile_rows) {
    TRY(bagel_gemm_ws_residual_rmsnorm_bf16(r->O, qd, u.wo, qd, r->X, H, Mt, H, qd, r->X, H, u.ln_post, r->Xn, H, eps, r->skinny_ws,
                                            r->skinny_ws_bytes, stream));
  } else {
    if (Mt && !drop_und_out) TRY(und_linear(at(r->O, Mv, qd), qd, u.wo, X_u, H, Mt, H, qd, nullptr, X_u, H, 1, r, stream));
    TRY(bagel_rmsnorm_bf16(r->X, H, r->Xn, H, drop_und_out ? Mv : M, H, ln_post_a, u.ln_post, split, eps, stream));
  }
  if (Mv) {
    TimedLaunch t{};
    bool timed = false;
    {
      std::lock_guard<std::mutex> lock(g_prof_mu);
      if (g_prof_on && hipEventCreate(&t.start) == hipSuccess) {
        timed = hipEventCreate(&t.stop) == hipSuccess;
        if (!timed) (void)hipEventDestroy(t.start);
      }
    }
    if (timed) (void)hipEventRecord(t.start, (hipStream_t)stream);
    const int rc_gu = bagel_gemm_bf16(r->Xn, H, g.wgu, H, r->Hm, I, Mv, 2 * I, H, nullptr, nullptr, 0, BAGEL_EPI_SWIGLU, 1.0f, stream);
    if (timed) {
      // recorded and handed to the collector even when the launch failed, so the pair is always destroyed there
      (void)hipEventRecord(t.stop, (hipStream_t)stream);
      t.flops = rc_gu ? 0.0 : 2.0 * Mv * 2.0 * I * H;
      std::lock_guard<std::mutex> lock(g_prof_mu);
      g_prof.push_back(t);
    }
    if (rc_gu) return rc_gu;
  }
  if (Mt && !drop_und_out) TRY(und_linear(Xn_u, H, u.wgu, at(r->Hm, Mv, I), I, Mt, 2 * I, H, nullptr, nullptr, 0, 2, r, stream));
  if (Mv) TRY(bagel_gemm_bf16(r->Hm, I, g.wd, I, r->X, H, Mv, H, I, nullptr, r->X, H, BAGEL_EPI_RESIDUAL, 1.0f, stream));
  if (drop_und_out) {
    if (next_norm_gen) TRY(bagel_rmsnorm_bf16(r->X, H, r->Xn, H, Mv, H, next_norm_gen, next_norm_gen, Mv, eps, stream));
    return 0;
  }
  if (fuse_norm && next_norm_und) {
    TRY(bagel_skinny_gemm_residual_rmsnorm_bf16(r->Hm, I, u.wd, I, r->X, H, Mt, H, I, r->X, H, next_norm_und, r->Xn, H, eps,
                                                r->skinny_ws, r->skinny_ws_bytes, stream));
    return 0;
  }
  if (tile_rows && next_norm_und) {
    TRY(bagel_gemm_ws_residual_rmsnorm_bf16(r->Hm, I, u.wd, I, r->X, H, Mt, H, I, r->X, H, next_norm_und, r->Xn, H, eps, r->skinny_ws,
                                            r->skinny_ws_bytes, stream));
    return 0;
  }
  if (Mt) TRY(und_linear(at(r->Hm, Mv, I), I, u.wd, X_u, H, Mt, H, I, nullptr, X_u, H, 1, r, stream));
  if (next_norm_gen || next_norm_und)
    TRY(bagel_rmsnorm_bf16(r->X, H, r->Xn, H, M, H, Mv ? next_norm_gen : next_norm_und, next_norm_und, split, eps, stream));
  return 0;
}

extern "C" int bagel_mot_forward(const bagel_mot_dims* d, const bagel_mot_layer* layers, int n_layers,
                                 const bagel_mot_rows* r, void* K_ws, void* Vt_ws, long layer_stride, int kv_only_last,
                                 const void* final_norm_gen, const void* final_norm_und, int final_norm_rows, void* stream) {
  BAGEL_CHECK(d && layers && r && n_layers > 0 && layer_stride > 0, "bagel_mot_forward: bad argument");
  const int Mv = r->rows_gen;
  // a decode step (up to four understanding rows, nothing else): the norms move into the weight-streaming kernels
  const bool lazy_norm = Mv == 0 && r->rows_und <= 4 &&
                         bagel_skinny_gemv_eligible(r->rows_und, (d->num_heads + 2 * d->num_kv_heads) * d->head_dim, d->hidden, BAGEL_SK_BIAS);
  if (lazy_norm) {
    for (int l = 0; l < n_layers; ++l) {
      const bool last = l + 1 == n_layers;
      const int flags = BAGEL_LAYER_NORM_IN_GEMM | (last && kv_only_last ? BAGEL_LAYER_KV_ONLY : 0);
      // Xn is only produced where the caller asked for the final norm
      const void* nu = last && !kv_only_last ? final_norm_und : nullptr;
      TRY(bagel_mot_layer_forward(d, &layers[l], r, (char*)K_ws + (long)l * layer_stride * 2,
                                  (char*)Vt_ws + (long)l * layer_stride * 2, flags, nullptr, nu, stream));
    }
    return 0;
  }
  for (int l = 0; l < n_layers; ++l) {
    const bool last = l + 1 == n_layers;
    int flags = l > 0 ? BAGEL_LAYER_XN_READY : 0;     // every layer leaves the next one's input norm in Xn
    if (last && kv_only_last) flags |= BAGEL_LAYER_KV_ONLY;
    // the caller reads the generation rows only (the velocity head, bagel.py:1425-1426)
    if (last && !kv_only_last && Mv > 0 && final_norm_rows > 0 && final_norm_rows <= Mv) flags |= BAGEL_LAYER_UND_OUT_UNUSED;
    // the norm that follows layer l: the next layer's input norm, or the model's final norm when it covers all rows
    const bool final_here = last && !kv_only_last && final_norm_rows <= 0;
    const void* ng = last ? (final_here ? final_norm_gen : nullptr) : layers[l + 1].gen.ln_in;
    const void* nu = last ? (final_here ? final_norm_und : nullptr) : layers[l + 1].und.ln_in;
    TRY(bagel_mot_layer_forward(d, &layers[l], r, (char*)K_ws + (long)l * layer_stride * 2, (char*)Vt_ws + (long)l * layer_stride * 2,
                                flags, ng, nu, stream));
  }
  if (!kv_only_last && final_norm_rows > 0 && (final_norm_gen || final_norm_und)) {
    const int rows = final_norm_rows;
    const int split = Mv ? Mv : rows;
    TRY(bagel_rmsnorm_bf16(r->X, d->hidden, r->Xn, d->hidden, rows, d->hidden, Mv ? final_norm_gen : final_norm_und,
                           final_norm_und, split, d->rms_eps, stream));
  }
  return 0;
}
