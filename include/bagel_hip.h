/* C ABI of libbagel_hip.so: the gfx950 (MI355X) kernels behind BAGEL's image-generation hot path.
 *
 * The reference has no FFI of its own: its hot path is Python methods on nn.Modules that call
 * torch / flash_attn library kernels.  Each entry point below replaces the library call(s) made at the
 * cited reference lines.  Conventions:
 *   - every pointer is a DEVICE pointer unless named otherwise; nothing is allocated or freed here;
 *   - bf16 tensors are row-major with element (not byte) leading dimensions;
 *   - `stream` is a hipStream_t; kernels are enqueued and the call returns immediately;
 *   - return 0 on success, non-zero on failure, message from bagel_last_error() (thread-local).
 */
#ifndef BAGEL_HIP_H
#define BAGEL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int bagel_abi_version(void);
const char* bagel_last_error(void);

/* epilogues of bagel_gemm_bf16 */
#define BAGEL_EPI_BIAS 0     /* C = bf16(acc + bias)  (bias may be NULL) */
#define BAGEL_EPI_RESIDUAL 1 /* C = bf16(bf16(acc) + R) */
#define BAGEL_EPI_SWIGLU 2   /* C[:, f] = bf16(bf16(silu(bf16(gate_f))) * bf16(up_f)); W rows interleaved
                                [16 gate rows | 16 up rows] per 16 features; C has N/2 columns */
#define BAGEL_EPI_BIAS_RESIDUAL 3 /* C = bf16(bf16(acc + bias) + R) */
#define BAGEL_EPI_F32_SCALED 4    /* C is fp32: C = alpha * acc */
#define BAGEL_EPI_BIAS_GELU 5     /* C = bf16(gelu_tanh(bf16(acc + bias))): SiglipMLP.fc1 + activation
                                     (siglip_navit.py:255-258), MLPconnector.fc1 (modeling_utils.py:122-125) */

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T), bf16 operands, fp32 accumulation on the MFMA units.
 * Replaces the nn.Linear calls of the generation expert: q/k/v_proj_moe_gen (qwen2_navit.py:541-547),
 * o_proj_moe_gen (:605), mlp_moe_gen gate/up/down (:833, modeling_qwen2.py:200-201), vae2llm and
 * llm2vae (bagel.py:1396,1425), and the residual adds (:818,836).  K % 64 == 0, N % 4 == 0,
 * lda and ldw (elements) multiples of 8 and below 2^21. */
int bagel_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                    const void* bias, const void* residual, long ldr, int epilogue, float alpha, void* stream);

/* The same product for callers that own a workspace.  17 ... 1024 rows against a projection of few output features
 * (the understanding expert's q/k/v, o and down projections during a text prompt's prefill, qwen2_navit.py:541-547,
 * 605, 833) are too few tiles to fill the chip: the reduction is then split over workgroups, the fp32 partial tiles go
 * through the workspace and a second launch adds them in slice order and applies the epilogue (bias, residual,
 * bias + residual, bias + GELU).  The split depends on (N, K) and on the row class, (16, 512] or (512, 1024], only.  With a null or too small workspace, and for every
 * other shape, this is bagel_gemm_bf16.  bagel_gemm_workspace_bytes: bytes the split form needs (0 = not used). */
int bagel_gemm_ws_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                       const void* bias, const void* residual, long ldr, int epilogue, float alpha, void* workspace,
                       long workspace_bytes, void* stream);
long bagel_gemm_workspace_bytes(int M, int N, int K, int epilogue);

/* bagel_gemm_ws_bf16 with the residual epilogue, followed by the RMSNorm of the result: C = bf16(bf16(A . W^T) + R),
 * Xn = norm_weight * bf16(C * rsqrt(mean(C^2) + eps)) - o_proj / down_proj of the understanding expert with the norm
 * that follows them (qwen2_navit.py:605, 818-823, 833-836; modeling_qwen2.py:54-59).  Where the reduction is split
 * (see above; N <= 8192) the launch that adds the partial tiles applies residual and norm too; C may be R. */
int bagel_gemm_ws_residual_rmsnorm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N,
                                        int K, const void* residual, long ldr, const void* norm_weight, void* Xn, long ldn,
                                        float eps, void* workspace, long workspace_bytes, void* stream);

/* epilogues of bagel_skinny_gemm_bf16 */
#define BAGEL_SK_BIAS 0
#define BAGEL_SK_RESIDUAL 1
#define BAGEL_SK_SWIGLU 2
#define BAGEL_SK_BIAS_SILU 3 /* C = bf16(silu(bf16(acc + bias))) */

/* Same contraction for M <= 64 rows, streaming every weight byte once (HBM-bound).
 * Replaces the understanding expert's nn.Linear calls on the two text tokens of each sample
 * (qwen2_navit.py:540,543,546,604,832) and TimestepEmbedder.mlp (modeling_utils.py:80-84,111).
 * N % 16 == 0, K % 128 == 0.  One launch: the K slices of a block of 16 weight rows are the waves of one workgroup
 * and meet in LDS.  A row's result does not depend on M (the K split is a function of N and K alone).
 * `workspace` (bagel_skinny_gemm_workspace_bytes) is only touched by the split-K fallback, taken when ldc / ldr are
 * not multiples of 4 elements or C / residual are not 8-byte aligned. */
long bagel_skinny_gemm_workspace_bytes(int M, int N);
int bagel_skinny_gemm_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                           const void* bias, const void* residual, long ldr, int epilogue, void* workspace,
                           long workspace_bytes, void* stream);

/* The same product with the SK_RESIDUAL epilogue, followed in the same launch by Qwen2RMSNorm of the result row:
 *   C = bf16(bf16(X . W^T) + residual),  Xn = norm_weight * bf16(C * rsqrt(mean(C^2) + eps))
 * (o_proj / down_proj + residual add + the next RMSNorm of Qwen2MoTDecoderLayer.forward_inference,
 * qwen2_navit.py:818-829, 836, and modeling_qwen2.py:54-59) for the one-token-per-sample decode loop of
 * Bagel.generate_text (bagel.py:1551-1587).  N <= 8192. */
int bagel_skinny_gemm_residual_rmsnorm_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M,
                                            int N, int K, const void* residual, long ldr, const void* norm_weight,
                                            void* Xn, long ldn, float eps, void* workspace, long workspace_bytes,
                                            void* stream);

/* C = epilogue(RMSNorm(X; norm_weight, eps) . W^T): Qwen2RMSNorm (modeling_qwen2.py:54-59) followed by nn.Linear, for
 * the one-token-per-sample decode loop (input_layernorm + q/k/v_proj, post_attention_layernorm + gate/up_proj, final
 * norm + lm_head: qwen2_navit.py:798-801, 826-832, 1098-1105, 1204).  With at most 4 rows (2 for SwiGLU and for
 * N > 16384) and K % 512 == 0 the rows are normalised inside the weight-streaming kernel and nothing else is
 * launched; otherwise RMSNorm runs into the caller's Xn rows ([M][ldn], may be NULL when the fused form applies:
 * bagel_skinny_gemv_eligible) and bagel_skinny_gemm_bf16 follows.  epilogue: BAGEL_SK_BIAS, BAGEL_SK_SWIGLU, BAGEL_SK_BIAS_SILU. */
int bagel_skinny_gemv_eligible(int M, int N, int K, int epilogue);

/* bagel_skinny_gemm_bf16's contract in the one-launch form where bagel_skinny_gemv_eligible() accepts the shape (a wave
 * owns whole weight rows: other fp32 summation order), bagel_skinny_gemm_bf16 itself otherwise.  For the decode
 * loop; bagel_skinny_gemm_bf16 never takes this form, so that a row's result never depends on how many
 * rows share the launch. */
int bagel_skinny_gemv_bf16(const void* X, long ldx, const void* W, long ldw, void* C, long ldc, int M, int N, int K,
                           const void* bias, const void* residual, long ldr, int epilogue, void* workspace,
                           long workspace_bytes, void* stream);
int bagel_skinny_gemm_rmsnorm_in_bf16(const void* X, long ldx, const void* norm_weight, float eps, const void* W, long ldw,
                                      void* C, long ldc, int M, int N, int K, const void* bias, int epilogue, void* Xn,
                                      long ldn, void* workspace, long workspace_bytes, void* stream);

/* out[r] = w * bf16(x[r] * rsqrt(mean(x[r]^2) + eps)); rows < split_row use w_first, the rest w_second.
 * Replaces Qwen2RMSNorm (modeling_qwen2.py:54-59) as used per modality at qwen2_navit.py:798-800,
 * 826-829 and 1102-1104. */
int bagel_rmsnorm_bf16(const void* x, long ldx, void* out, long ldo, int rows, int H, const void* w_first,
                       const void* w_second, int split_row, float eps, void* stream);

/* Per-head RMSNorm of q and k + RoPE + scatter of q, k, v into the attention operands.
 * Replaces qwen2_navit.py:549-585: the fp32 q/k norms per modality, apply_rotary_pos_emb, the bf16 casts
 * and the per-layer rebuild of merged_key_states / merged_value_states.
 *   qkv      [M][(Hq+2Hkv)*D]  fused projection output (q heads, k heads, v heads)
 *   row_cs   [M] int32         row of cos_table / sin_table ([*][D] bf16) for each token
 *   row_slot [M] int32         key slot of each token in the KV workspace
 *   Q        [M][Hq][D]        K_ws [Hkv][S][D]        Vt_ws [Hkv][D][S]
 * und_rounding != 0 follows the understanding-mode graph (:526-530: bf16 norm output, bf16 RoPE). */
int bagel_qk_norm_rope_kv_bf16(const void* qkv, long ld, int M, int Hq, int Hkv, int head_dim,
                               const void* q_norm_first, const void* q_norm_second, const void* k_norm_first,
                               const void* k_norm_second, int split_row, const void* cos_table,
                               const void* sin_table, const void* row_cs, const void* row_slot, void* Q, void* K_ws,
                               void* Vt_ws, long S, float eps, int und_rounding, void* stream);

/* Packed varlen GQA attention: replaces flash_attn_varlen_func (qwen2_navit.py:590-599).
 *   tiles [n_tiles] of 8 x int32 {q_row0, n_rows (<=128), kv_start (multiple of 64), kv_len, causal_base, 0, 0, 0}
 * Every query row of a tile attends key slots [kv_start, kv_start + kv_len) of its kv head; with
 * causal_base >= 0, row r of the tile only sees the first causal_base + r + 1 of them (the bottom-right
 * aligned causal mask of flash-attn >= 2.1; causal_base = context length + index of the tile's first row
 * among the sample's new tokens).  causal_base = -1 for the non-causal generation forward.
 * S (key slots per kv head, a multiple of 64) is limited to S * head_dim * 2 < 2^31 bytes.
 * Precondition: keys are consumed in tiles of 64 slots, so the slots from kv_start + kv_len up to the next
 * multiple of 64 are read.  Their scores are replaced by -inf (K_ws may hold anything there), but their zero
 * probabilities still multiply Vt_ws: those Vt_ws slots must hold FINITE values (0 x NaN is NaN). */
int bagel_varlen_attention_bf16(const void* Q, const void* K_ws, const void* Vt_ws, void* O, const void* tiles,
                                int n_tiles, int Hq, int Hkv, int head_dim, long S, void* stream);

/* The same kernel with the softmax scale given by the caller (bagel_varlen_attention_bf16 uses
 * 1/sqrt(head_dim)): for heads narrower than the kernel's head_dim that were zero-padded, e.g. the 72-wide
 * heads of flash_attn_varlen_func at siglip_navit.py:232-241. */
int bagel_varlen_attention_scaled_bf16(const void* Q, const void* K_ws, const void* Vt_ws, void* O, const void* tiles,
                                       int n_tiles, int Hq, int Hkv, int head_dim, long S, float softmax_scale,
                                       void* stream);

/* nn.LayerNorm with fp32 statistics on bf16 rows (H % 8 == 0, H <= 4096), bf16 affine:
 * siglip_navit.py:267-269, 283, 296, 343, 370. */
int bagel_layernorm_bf16(const void* x, long ldx, void* out, long ldo, int rows, int H, const void* gamma,
                         const void* beta, float eps, void* stream);

/* qkv [M][3][heads][head_dim_in] -> Q [M][heads][head_dim], K_ws [heads][S][head_dim] and Vt_ws
 * [heads][head_dim][S] at slot row_slot[row], zero-padding head_dim_in up to head_dim (no q/k norm, no
 * RoPE): the reshapes of SiglipFlashAttention2.forward, siglip_navit.py:216-222.  The pad rows of Vt_ws are
 * not written: zero the workspace once. */
int bagel_split_heads_pad_bf16(const void* qkv, long ld, int M, int heads, int head_dim_in, int head_dim,
                               const void* row_slot, void* Q, void* K_ws, void* Vt_ws, long S, void* stream);

/* ---- one decoder layer / the whole layer stack as single calls ---------------------------------------------
 * Qwen2MoTDecoderLayer.forward_inference (qwen2_navit.py:771-845) with PackedAttentionMoT.forward_inference
 * (:510-611) inside, and the layer loop + final norm of Qwen2Model.forward_inference (:1079-1105).
 * The packed sequence is sorted by expert: rows [0, rows_gen) use the generation expert, the following rows_und
 * rows the understanding expert (mode="und": rows_gen == 0).  All pointers are device pointers; the weight layouts
 * are those of bagel_gemm_bf16 / bagel_skinny_gemm_bf16 (wqkv = q|k|v rows fused, wgu = gate/up interleaved per 16
 * features for BAGEL_EPI_SWIGLU); activations are row-contiguous bf16 at their natural widths. */
typedef struct bagel_mot_dims {
  int hidden, intermediate, num_heads, num_kv_heads, head_dim;
  float rms_eps;
} bagel_mot_dims;

typedef struct bagel_mot_expert {      /* one expert's parameters of one layer */
  const void *wqkv, *bqkv, *wo, *wgu, *wd; /* [(Hq+2Hkv)D, H], [(Hq+2Hkv)D], [H, Hq D], [2I, H], [H, I] */
  const void *ln_in, *ln_post;             /* input_layernorm, post_attention_layernorm [H] */
  const void *q_norm, *k_norm;             /* [D] */
} bagel_mot_expert;

typedef struct bagel_mot_layer {
  bagel_mot_expert gen, und;           /* ``*_moe_gen`` parameters / plain parameters */
} bagel_mot_layer;

typedef struct bagel_mot_rows {        /* one packed forward: activations, tables, scratch */
  int rows_gen, rows_und;
  void *X, *Xn;                        /* [M][H] residual stream / normalised copy (M = rows_gen + rows_und) */
  void *QKV, *Q, *O, *Hm;              /* [M][(Hq+2Hkv)D], [M][Hq D], [M][Hq D], [M][I] */
  const void *cos_table, *sin_table;   /* RoPE rows, see bagel_qk_norm_rope_kv_bf16 */
  const void *row_cs, *row_slot;       /* [M] int32 */
  const void* tiles;                   /* attention tile table, see bagel_varlen_attention_bf16 */
  int n_tiles;
  long S;                              /* key slots per kv head in K_ws / Vt_ws */
  void* skinny_ws;                     /* fp32 split-K partials: bagel_skinny_gemm_workspace_bytes(min(rows_und, 64), 2I) */
  long skinny_ws_bytes;
  int und_rounding;                    /* mode="und" rounding of the q/k norm and RoPE (qwen2_navit.py:526-530) */
  void* decode_ws;                     /* optional: bagel_decode_attention_workspace_bytes(), zeroed once by the caller.
                                          Says that every row is one sample's single new token and tiles[i] describes
                                          row i; the layer then runs bagel_decode_attention_bf16 where it applies */
  long decode_ws_bytes;
  long decode_max_kv;                  /* upper bound of a row's kv_len in decode mode (0: unknown) */
} bagel_mot_rows;

#define BAGEL_LAYER_KV_ONLY 1  /* stop once the layer's keys/values are in K_ws / Vt_ws (last layer of a prefill) */
#define BAGEL_LAYER_XN_READY 2 /* Xn already holds input_layernorm(X): the previous layer's call produced it */
#define BAGEL_LAYER_UND_OUT_UNUSED 8 /* rows_gen > 0: nobody reads the understanding rows of X / Xn after this layer (last
                                       layer of a generation forward, whose head takes the latent rows only,
                                       bagel.py:1425-1426): their o_proj and MLP are skipped, their keys/values still
                                       land in K_ws / Vt_ws; X and Xn of those rows are undefined on return */
#define BAGEL_LAYER_NORM_IN_GEMM 4 /* rows_gen == 0, a few rows: both RMSNorms run inside the projections that consume
                                      them (bagel_skinny_gemm_rmsnorm_in_bf16); Xn is written only for next_norm_und */

/* A decode step's attention in one launch: q/k RMSNorm + RoPE of row i, its new key / value written to slot
 * row_slot[i] of K_ws / Vt_ws, attention of the row over keys [kv_start, kv_start + kv_len) of tiles[i] (the new
 * slot included), O[i] = result.  Equivalent to bagel_qk_norm_rope_kv_bf16 + bagel_varlen_attention_bf16 on rows that
 * are each one sample's single new token (Bagel.generate_text, bagel.py:1551-1587; qwen2_navit.py:549-599), with the
 * keys of a sample split over workgroups and merged by the last one to finish.
 * workspace: bagel_decode_attention_workspace_bytes(), its first 1024 bytes zero before the first call (the kernel
 * leaves them zero); calls sharing a workspace must not overlap.  max_kv_len: an upper bound of every row's kv_len
 * (sizes the grid; 0 = unknown, S is used; a bound below a row's kv_len only makes that row's chunks longer).  Same finite-padding precondition on Vt_ws as
 * bagel_varlen_attention_bf16.  bagel_decode_attention_supported() != 0: rows * Hkv <= 256, Hq / Hkv 7 or 2,
 * head_dim 128 or 64. */
int bagel_decode_attention_supported(int rows, int Hq, int Hkv, int head_dim);
long bagel_decode_attention_workspace_bytes(int rows, int Hq, int Hkv, int head_dim);
int bagel_decode_attention_bf16(const void* qkv, long ld, int rows, int Hq, int Hkv, int head_dim, const void* q_norm,
                                const void* k_norm, const void* cos_table, const void* sin_table, const void* row_cs,
                                const void* row_slot, const void* tiles, void* K_ws, void* Vt_ws, long S, void* O,
                                float eps, int und_rounding, long max_kv_len, void* workspace, long workspace_bytes,
                                void* stream);

/* One layer, in place on r->X.  K_ws [Hkv][S][D] and Vt_ws [Hkv][D][S] are THIS layer's workspace slices.
 * next_norm_gen / next_norm_und (optional): weights of the RMSNorm that follows the layer (the next layer's
 * input_layernorm); when given, Xn holds that norm of the layer's output on return. */
int bagel_mot_layer_forward(const bagel_mot_dims* dims, const bagel_mot_layer* layer, const bagel_mot_rows* rows,
                            void* K_ws, void* Vt_ws, int flags, const void* next_norm_gen, const void* next_norm_und,
                            void* stream);

/* All layers: layer l uses K_ws + l * layer_stride and Vt_ws + l * layer_stride (elements).  kv_only_last != 0
 * skips the last layer's attention and MLP (a context prefill only keeps keys/values).  Otherwise Xn holds
 * norm(X) (final_norm_gen on the first rows_gen rows, final_norm_und on the rest; first final_norm_rows rows
 * only when > 0) on return: Qwen2Model.norm / norm_moe_gen, qwen2_navit.py:1098-1105.  With 0 < final_norm_rows <=
 * rows_gen the caller reads generation rows only and the last layer runs with BAGEL_LAYER_UND_OUT_UNUSED. */
int bagel_mot_forward(const bagel_mot_dims* dims, const bagel_mot_layer* layers, int n_layers, const bagel_mot_rows* rows,
                      void* K_ws, void* Vt_ws, long layer_stride, int kv_only_last, const void* final_norm_gen,
                      const void* final_norm_und, int final_norm_rows, void* stream);

/* Measurement hook (no counterpart in the reference): while on, every generation-expert gate+up GEMM launched by
 * bagel_mot_layer_forward is bracketed by HIP events on its stream; bagel_profile_collect waits for them and
 * returns (milliseconds, algorithmic flops) per launch. */
void bagel_profile_gate_up(int on);
int bagel_profile_collect(double* ms, double* flops, int cap);

/* X[r] = bf16(bf16(lin[src_row[r]] + temb) + pos_table[pos_id[src_row[r]]]): bagel.py:1394-1399. */
int bagel_embed_latent_bf16(const void* lin, long ldl, const void* temb, const void* pos_table, const void* pos_id,
                            const void* src_row, void* X, long ldx, int rows, int H, void* stream);

/* out[r] = table[idx[r]] (idx == NULL: identity): embed_tokens lookup, bagel.py:1389-1391. */
int bagel_gather_rows_bf16(const void* table, long ldt, const void* idx, void* out, long ldo, int rows, int H,
                           void* stream);

/* out[i] = [cos(t_i f) | sin(t_i f)], f_k = exp(-ln(1e4) k / (dim/2)): modeling_utils.py:88-107. */
int bagel_timestep_sinusoid_bf16(const void* t, int n, int dim, void* out, void* stream);

int bagel_cast_f32_bf16(const void* x, void* out, long n, void* stream);

/* TaylorSeer feature cache of the last decoder layer (modeling/cache_utils/taylorseer.py:11-45 as used at
 * qwen2_navit.py:838-843): factors [7][factor_stride] bf16 (factor_stride >= rows * H elements), every elementwise op rounded to bf16 like the
 * reference's tensor arithmetic.
 *   update: F_0 = feature, F_{i+1} = ((F_i_new - F_i_old) / distance) for i < n_new - 1   (a full step)
 *   eval:   out = sum_{i < n_factors} (1 / i!) * F_i * x^i                                 (a skipped step) */
int bagel_taylor_update_bf16(const void* feature, long ldf, int rows, int H, void* factors, long factor_stride,
                             int n_new, int distance, void* stream);
int bagel_taylor_eval_bf16(const void* factors, long factor_stride, int rows, int H, int n_factors, int x, void* out,
                           long ldo, void* stream);

/* out[r] = index of the first maximum of logits[r][0..n) (int32): torch.argmax at bagel.py:1580. */
int bagel_argmax_rows_bf16(const void* logits, long ld, int rows, int n, void* out_i32, void* stream);

/* CFG combine + renorm + Euler update: bagel.py:1466-1500 and :1339.
 *   renorm_type 0 "global", 1 "channel", 2 "text_channel"; v_* are [rows][P] bf16 (P == 64);
 *   x_t fp32 [rows][P] is updated in place: x_t -= bf16(v_t * dt); x_bf16 receives bf16(x_t);
 *   v_t_out (optional) receives the combined velocity.  CFG is applied iff cfg_text_scale > 1. */
long bagel_cfg_workspace_bytes(int rows, int P);
int bagel_cfg_euler_update(const void* v_cond, const void* v_text, const void* v_img, float cfg_text_scale,
                           float cfg_img_scale, int renorm_type, float renorm_min, float dt, void* x_t, void* x_bf16,
                           void* v_t_out, int rows, int P, void* workspace, long workspace_bytes, void* stream);

/* bagel_cfg_euler_update without a workspace argument and without any allocation.  No counterpart in the
 * reference: it exists for callers that cannot allocate device memory (a flow-step loop under HIP graph capture,
 * serving loops and harnesses that forbid hipMalloc).  The scratch (pre-renorm velocity, per-row sums of squares,
 * global scale: 8.5 MiB) is a file-scope device array of the library, one copy per GPU, held until the process
 * ends.  rows <= bagel_cfg_static_max_rows() (65536 = 16 images of 1024x1024), otherwise an error is returned.
 * Results are bit-identical to bagel_cfg_euler_update.  Every word of the scratch that a call reads it has written
 * first, so nothing is cleared between calls; all calls on one GPU share the array, so calls that overlap on
 * one GPU (different streams) race - serialise them. */
int bagel_cfg_static_max_rows(void);
int bagel_cfg_euler_update_static(const void* v_cond, const void* v_text, const void* v_img, float cfg_text_scale,
                                  float cfg_img_scale, int renorm_type, float renorm_min, float dt, void* x_t,
                                  void* x_bf16, void* v_t_out, int rows, int P, void* stream);

/* ---- FLUX VAE decoder (modeling/autoencoder.py:196-272, 320-322); activations are NHWC bf16 ---- */

/* 3x3 convolution as an implicit GEMM on the MFMA units.  `upsample` selects the resampling fused into the
 * input gather: 0 = stride 1, padding 1; 1 = the same after nearest-neighbour 2x upsampling of x
 * (Upsample.forward, autoencoder.py:116-119); 2 = stride 2 after one zero row/column of padding on the
 * bottom/right (Downsample.forward, autoencoder.py:104-108; Hout = Hin / 2).
 *   x [Hin][Win][Cin], w [Cout][3][3][Cin], bias [Cout], residual/out [Hout*Wout][Cout]
 *   out = bf16(acc + bias), or bf16(bf16(acc + bias) + residual) when residual != NULL
 *   zero_page: >= 128 zero bytes of device memory (source of the padding taps).
 * Replaces nn.Conv2d at autoencoder.py:76,78,102,114,170,248 and the `x + h` of ResnetBlock (:95). */
int bagel_conv3x3_nhwc_bf16(const void* x, int Hin, int Win, int Cin, int upsample, const void* w, int Cout,
                            const void* bias, const void* residual, void* out, const void* zero_page, void* stream);

/* GroupNorm(32, eps) with fp32 affine, optional swish, fp32 statistics, bf16 in/out.
 * Replaces nn.GroupNorm + swish at autoencoder.py:84-89, 51, 269-270. */
long bagel_groupnorm_workspace_bytes(int C);
int bagel_groupnorm_swish_nhwc_bf16(const void* x, long npix, int C, const void* gamma_f32, const void* beta_f32,
                                    float eps, int swish, void* out, void* workspace, long workspace_bytes,
                                    void* stream);

/* probs[r] = softmax(scores[r]) (fp32 in, bf16 out): the softmax of AttnBlock.attention (autoencoder.py:60). */
int bagel_softmax_rows_f32_bf16(const void* scores, long lds, int rows, int n, void* probs, long ldp, void* stream);

/* out[C][R] = in[R][C]^T (bf16). */
int bagel_transpose_bf16(const void* in, int R, int C, void* out, void* stream);

/* im2col rows [H*W][Kpad] of bf16(z / scale_factor + shift_factor) for conv_in (autoencoder.py:321,252);
 * z is NCHW fp32, K order (ky, kx, c), zero padded to Kpad (multiple of 64). */
int bagel_vae_latent_im2col_bf16(const void* z_nchw_f32, int Cz, int H, int W, float scale_factor, float shift_factor,
                                 void* out, int Kpad, void* stream);

/* z = scale * ((mean + exp(0.5 logvar) * noise) - shift), bf16 with a rounding after every elementwise op:
 * DiagonalGaussian.forward + AutoEncoder.encode (autoencoder.py:281-287, 315-318).
 *   moments NHWC [H*W][ldm] bf16 (mean | logvar channels), noise and z NCHW [Cz][H][W] bf16. */
int bagel_vae_sample_latent_bf16(const void* moments_nhwc, int ldm, const void* noise_nchw, int Cz, int H, int W,
                                 float scale_factor, float shift_factor, void* z_nchw, void* stream);

/* NHWC bf16 (Cpad channels) -> NCHW bf16 (first C channels). */
int bagel_nhwc_to_nchw_bf16(const void* in, long npix, int Cpad, int C, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BAGEL_HIP_H */
