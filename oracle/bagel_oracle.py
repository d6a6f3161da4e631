"""CPU restatement of BAGEL's packed flow-matching denoise step (TEST INFRASTRUCTURE).

This file is a checker, not a product path: only ``tests/``, the smoke check and
the benchmark's CPU-baseline leg may import it.  It restates, in plain PyTorch on
the CPU, the arithmetic the reference performs for image generation, following the
reference's rounding points (bf16 storage, fp32 internals) one by one.  Every
function cites the reference lines it follows.  It is pinned against outputs of
the reference itself (tests/golden/*.safetensors, produced by
tests/golden/make_golden.py which imports the reference in the authoring
container).

The third-party ``flash_attn_varlen_func`` (pinned ``flash_attn==2.5.8`` in the
reference's requirements.txt:16, source absent) is restated by its published
definition: per-sample softmax(QK^T/sqrt(D))V with grouped KV heads.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


class exact_arithmetic:
    """Context manager: evaluate the same graph with every bf16 rounding point turned into fp32.

    Used with an fp32 copy of the weights to obtain the exact-arithmetic answer, against which
    the reference's own bf16 rounding noise is measured (the "noise floor" of a scenario).
    """

    def __enter__(self):
        global BF16
        self._saved = BF16
        BF16 = torch.float32
        return self

    def __exit__(self, *exc):
        global BF16
        BF16 = self._saved
        return False


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """Qwen2RMSNorm.forward, modeling/qwen2/modeling_qwen2.py:54-59."""
    in_dtype = x.dtype
    h = x.to(torch.float32)
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return w * h.to(in_dtype)


def rotary_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float,
                   dtype=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Qwen2RotaryEmbedding.forward, modeling_qwen2.py:130-150 (default rope, scaling 1.0).

    inv_freq = 1/theta^(arange(0,D,2)/D); cos/sin are rounded to the activation dtype.
    """
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    freqs = (inv_freq[None, :, None].float() @ position_ids[None, None, :].float()).transpose(1, 2)[0]
    emb = torch.cat((freqs, freqs), dim=-1)
    dtype = BF16 if dtype is None else dtype
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """modeling_qwen2.py:154-158."""
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """apply_rotary_pos_emb with unsqueeze_dim=1, modeling_qwen2.py:162-186."""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear under bf16 autocast: operands cast to bf16, fp32 accumulate, bf16 result."""
    return F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))


def mlp(x, p: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """Qwen2MLP.forward, modeling_qwen2.py:200-201 (silu)."""
    g = linear(x, p[prefix + "gate_proj.weight"])
    u = linear(x, p[prefix + "up_proj.weight"])
    return linear(F.silu(g) * u, p[prefix + "down_proj.weight"])


def varlen_attention(q, k, v, q_lens: Sequence[int], k_lens: Sequence[int], causal: bool) -> torch.Tensor:
    """Definition of flash_attn_varlen_func as called at qwen2_navit.py:590-599.

    q [sum(q_lens), Hq, D], k/v [sum(k_lens), Hkv, D]; scale 1/sqrt(D); with
    ``causal`` the mask is aligned to the bottom-right corner (flash-attn >= 2.1).
    Follows the published FlashAttention-2 arithmetic: fp32 scores and softmax
    statistics, probabilities rounded to the input dtype before the P.V product,
    fp32 accumulation, normalisation by the fp32 row sum, result in the input dtype.
    """
    Hq, Hkv = q.shape[1], k.shape[1]
    rep = Hq // Hkv
    outs = []
    qo = ko = 0
    for lq, lk in zip(q_lens, k_lens):
        qs = q[qo:qo + lq].transpose(0, 1).float()            # [Hq, lq, D]
        ks = k[ko:ko + lk].transpose(0, 1).float().repeat_interleave(rep, dim=0)
        vs = v[ko:ko + lk].transpose(0, 1).float().repeat_interleave(rep, dim=0)
        s = qs @ ks.transpose(1, 2) / math.sqrt(q.shape[-1])
        if causal:
            i = torch.arange(lq)[:, None]
            j = torch.arange(lk)[None, :]
            s = s.masked_fill(j > i + (lk - lq), float("-inf"))
        pexp = torch.exp(s - s.max(dim=-1, keepdim=True).values)
        o = (pexp.to(q.dtype).float() @ vs) / pexp.sum(dim=-1, keepdim=True)
        outs.append(o.transpose(0, 1).to(q.dtype))
        qo += lq
        ko += lk
    return torch.cat(outs, dim=0)


class Cache:
    """NaiveCache, qwen2_navit.py:214-228."""

    def __init__(self, num_layers: int):
        self.key_cache = {k: None for k in range(num_layers)}
        self.value_cache = {k: None for k in range(num_layers)}

    @property
    def num_layers(self):
        return len(self.key_cache)

    @property
    def seq_lens(self):
        return 0 if self.key_cache[0] is None else self.key_cache[0].shape[0]


# ----------------------------------------------------------------------------
# MoT layer and model forward
# ----------------------------------------------------------------------------
def attention_inference(p, pre, dims, layer_idx, x, query_lens, cos, sin, packed_query_indexes,
                        past: Optional[Cache], key_values_lens, packed_key_value_indexes,
                        update_past_key_values, is_causal, mode, vae_idx, text_idx):
    """PackedAttentionMoT.forward_inference, qwen2_navit.py:510-611."""
    Hq, Hkv, D, eps = dims.num_heads, dims.num_kv_heads, dims.head_dim, dims.rms_norm_eps
    a = pre + "self_attn."
    if mode == "und":
        q = linear(x, p[a + "q_proj.weight"], p[a + "q_proj.bias"]).view(-1, Hq, D)
        k = linear(x, p[a + "k_proj.weight"], p[a + "k_proj.bias"]).view(-1, Hkv, D)
        v = linear(x, p[a + "v_proj.weight"], p[a + "v_proj.bias"]).view(-1, Hkv, D)
        q = rms_norm(q, p[a + "q_norm.weight"], eps)
        k = rms_norm(k, p[a + "k_norm.weight"], eps)
    else:
        x = x.to(BF16)
        n = x.shape[0]
        q = x.new_zeros((n, Hq * D))
        k = x.new_zeros((n, Hkv * D))
        v = x.new_zeros((n, Hkv * D))
        xt, xv = x[text_idx], x[vae_idx]
        q[text_idx] = linear(xt, p[a + "q_proj.weight"], p[a + "q_proj.bias"])
        q[vae_idx] = linear(xv, p[a + "q_proj_moe_gen.weight"], p[a + "q_proj_moe_gen.bias"])
        k[text_idx] = linear(xt, p[a + "k_proj.weight"], p[a + "k_proj.bias"])
        k[vae_idx] = linear(xv, p[a + "k_proj_moe_gen.weight"], p[a + "k_proj_moe_gen.bias"])
        v[text_idx] = linear(xt, p[a + "v_proj.weight"], p[a + "v_proj.bias"])
        v[vae_idx] = linear(xv, p[a + "v_proj_moe_gen.weight"], p[a + "v_proj_moe_gen.bias"])
        q = q.view(-1, Hq, D).to(torch.float32)
        k = k.view(-1, Hkv, D).to(torch.float32)
        v = v.view(-1, Hkv, D)
        q[text_idx] = rms_norm(q[text_idx], p[a + "q_norm.weight"], eps)
        q[vae_idx] = rms_norm(q[vae_idx], p[a + "q_norm_moe_gen.weight"], eps)
        k[text_idx] = rms_norm(k[text_idx], p[a + "k_norm.weight"], eps)
        k[vae_idx] = rms_norm(k[vae_idx], p[a + "k_norm_moe_gen.weight"], eps)

    q, k = apply_rope(q, k, cos, sin)
    q, k, v = q.to(BF16), k.to(BF16), v.to(BF16)

    q_lens = [int(t) for t in query_lens]
    if past is not None and past.key_cache[layer_idx] is not None:
        pk, pv = past.key_cache[layer_idx], past.value_cache[layer_idx]
        kv_lens = [int(t) for t in key_values_lens]
        total = sum(q_lens) + sum(kv_lens)
        mk = pk.new_zeros((total, Hkv, D))
        mv = pk.new_zeros((total, Hkv, D))
        mk[packed_query_indexes] = k
        mk[packed_key_value_indexes] = pk
        mv[packed_query_indexes] = v
        mv[packed_key_value_indexes] = pv
        k_lens = [a_ + b_ for a_, b_ in zip(kv_lens, q_lens)]
    else:
        mk, mv = k, v
        k_lens = q_lens

    o = varlen_attention(q, mk, mv, q_lens, k_lens, is_causal).reshape(-1, Hq * D)
    if mode == "und":
        o = linear(o, p[a + "o_proj.weight"])
    else:
        o2 = o.clone()
        o2[text_idx] = linear(o[text_idx], p[a + "o_proj.weight"])
        o2[vae_idx] = linear(o[vae_idx], p[a + "o_proj_moe_gen.weight"])
        o = o2
    if update_past_key_values:
        past.key_cache[layer_idx] = mk
        past.value_cache[layer_idx] = mv
    return o, past


def layer_inference(p, dims, layer_idx, x, query_lens, cos, sin, packed_query_indexes, past,
                    key_values_lens, packed_key_value_indexes, update_past_key_values, is_causal,
                    mode, vae_idx, text_idx):
    """Qwen2MoTDecoderLayer.forward_inference, qwen2_navit.py:771-845 (TaylorSeer off)."""
    pre = f"language_model.model.layers.{layer_idx}."
    eps = dims.rms_norm_eps
    residual = x
    if mode == "und":
        h = rms_norm(x, p[pre + "input_layernorm.weight"], eps)
    else:
        h = torch.zeros_like(x)
        h[text_idx] = rms_norm(x[text_idx], p[pre + "input_layernorm.weight"], eps)
        h[vae_idx] = rms_norm(x[vae_idx], p[pre + "input_layernorm_moe_gen.weight"], eps)
    h, past = attention_inference(p, pre, dims, layer_idx, h, query_lens, cos, sin, packed_query_indexes,
                                  past, key_values_lens, packed_key_value_indexes,
                                  update_past_key_values, is_causal, mode, vae_idx, text_idx)
    x = residual + h
    residual = x
    if mode == "und":
        h = mlp(rms_norm(x, p[pre + "post_attention_layernorm.weight"], eps), p, pre + "mlp.")
    else:
        ht = rms_norm(x[text_idx], p[pre + "post_attention_layernorm.weight"], eps).to(BF16)
        hv = rms_norm(x[vae_idx], p[pre + "post_attention_layernorm_moe_gen.weight"], eps).to(BF16)
        h = torch.zeros_like(x).to(BF16)
        h[text_idx] = mlp(ht, p, pre + "mlp.")
        h[vae_idx] = mlp(hv, p, pre + "mlp_moe_gen.")
    return residual + h, past


class TaylorState:
    """One branch's TaylorSeer bookkeeping: modeling/cache_utils/taylorseer.py:58-153 with the constants of
    cache_init (fresh_threshold 3, max_order 6, first_enhance 5, taylor_cache on).

    On a 'Taylor' step every decoder layer returns the Taylor expansion of its own cached output and ignores its
    input (qwen2_navit.py:838-843), so only the LAST layer's factors reach the model output; they are the only
    ones kept here."""

    def __init__(self, num_steps: int):
        self.step = 0
        self.num_steps = num_steps
        self.activated_steps = [0]
        self.cache_counter = 0
        self.fresh_threshold, self.max_order, self.first_enhance = 3, 6, 5
        self.cal_threshold = None
        self.factors: Dict[int, torch.Tensor] = {}
        self.type = "full"

    def cal_type(self):
        """taylorseer.py:83-117 (the taylor_cache branch) + force_scheduler (:61-77, step weight 0)."""
        first = self.step < self.first_enhance
        interval = self.fresh_threshold if first else self.cal_threshold
        if first or self.cache_counter == interval - 1:
            self.type = "full"
            self.cache_counter = 0
            self.activated_steps.append(self.step)
            self.cal_threshold = round(self.fresh_threshold / 1.0)
        else:
            self.cache_counter += 1
            self.type = "Taylor"

    def update(self, feature: torch.Tensor):
        """derivative_approximation, taylorseer.py:11-29."""
        distance = self.activated_steps[-1] - self.activated_steps[-2]
        new = {0: feature}
        for i in range(self.max_order):
            if self.factors.get(i) is not None and self.step > self.first_enhance - 2:
                new[i + 1] = (new[i] - self.factors[i]) / distance
            else:
                break
        self.factors = new

    def expand(self) -> torch.Tensor:
        """taylor_formula, taylorseer.py:31-45."""
        x = self.step - self.activated_steps[-1]
        out = 0
        for i in range(len(self.factors)):
            out = out + (1 / math.factorial(i)) * self.factors[i] * (x ** i)
        return out


def llm_forward_inference(p, dims, x, query_lens, packed_query_position_ids, packed_query_indexes,
                          past=None, key_values_lens=None, packed_key_value_indexes=None,
                          update_past_key_values=True, is_causal=True, mode="und",
                          packed_vae_token_indexes=None, packed_text_indexes=None, taylor: Optional[TaylorState] = None):
    """Qwen2Model.forward_inference, qwen2_navit.py:1041-1115; ``taylor``: this branch's TaylorSeer state."""
    cos, sin = rotary_cos_sin(packed_query_position_ids, dims.head_dim, dims.rope_theta, x.dtype)
    if taylor is not None:
        taylor.cal_type()
        if taylor.type == "full" and taylor.step == 0:
            taylor.factors = {}                      # taylor_cache_init, taylorseer.py:47-56
    for i in range(dims.num_layers):
        if taylor is not None and taylor.type == "Taylor":
            x = taylor.expand()
            break
        x, past = layer_inference(p, dims, i, x, query_lens, cos, sin, packed_query_indexes, past,
                                  key_values_lens, packed_key_value_indexes, update_past_key_values,
                                  is_causal, mode, packed_vae_token_indexes, packed_text_indexes)
    if taylor is not None:
        if taylor.type == "full":
            taylor.update(x)
        taylor.step += 1
    eps = dims.rms_norm_eps
    if mode == "und":
        x = rms_norm(x, p["language_model.model.norm.weight"], eps)
    else:
        y = torch.zeros_like(x)
        y[packed_text_indexes] = rms_norm(x[packed_text_indexes], p["language_model.model.norm.weight"], eps)
        y[packed_vae_token_indexes] = rms_norm(x[packed_vae_token_indexes],
                                               p["language_model.model.norm_moe_gen.weight"], eps)
        x = y
    return x, past


# ----------------------------------------------------------------------------
# embedders
# ----------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """TimestepEmbedder.timestep_embedding, modeling/bagel/modeling_utils.py:88-107."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.dtype)
    args = t[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def time_embedder(p, t: torch.Tensor, dim: int) -> torch.Tensor:
    """TimestepEmbedder.forward, modeling_utils.py:109-112 (Linear, SiLU, Linear under bf16)."""
    f = timestep_embedding(t, dim)
    h = linear(f, p["time_embedder.mlp.0.weight"], p["time_embedder.mlp.0.bias"])
    return linear(F.silu(h), p["time_embedder.mlp.2.weight"], p["time_embedder.mlp.2.bias"])


def flattened_position_ids_extrapolate(img_h, img_w, patch_size, max_num_patches_per_side):
    """data/data_utils.py:53-58."""
    nh, nw = img_h // patch_size, img_w // patch_size
    return (torch.arange(0, nh)[:, None] * max_num_patches_per_side + torch.arange(0, nw)).flatten()


# ----------------------------------------------------------------------------
# host index builders
# ----------------------------------------------------------------------------
def prepare_prompts(curr_kvlens, curr_rope, prompt_token_ids: List[List[int]], new_token_ids):
    """Bagel.prepare_prompts, bagel.py:806-841, with the tokenizer call replaced by given ids."""
    ids, pos, lens, idx, kvidx = [], [], [], [], []
    curr = 0
    newlens, new_rope = [], []
    for toks, kvlen, rope in zip(prompt_token_ids, curr_kvlens, curr_rope):
        kvidx.extend(range(curr, curr + kvlen))
        curr += kvlen
        t = [new_token_ids["bos_token_id"]] + list(toks) + [new_token_ids["eos_token_id"]]
        lens.append(len(t))
        ids.extend(t)
        pos.extend(range(rope, rope + len(t)))
        idx.extend(range(curr, curr + len(t)))
        newlens.append(kvlen + len(t))
        new_rope.append(rope + len(t))
        curr += len(t)
    gi = {
        "text_token_lens": torch.tensor(lens, dtype=torch.int),
        "packed_text_ids": torch.tensor(ids, dtype=torch.long),
        "packed_text_position_ids": torch.tensor(pos, dtype=torch.long),
        "packed_text_indexes": torch.tensor(idx, dtype=torch.long),
        "packed_key_value_indexes": torch.tensor(kvidx, dtype=torch.long),
        "key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
    }
    return gi, newlens, new_rope


def forward_cache_update_text(p, dims, past, packed_text_ids, packed_text_position_ids, text_token_lens,
                              packed_text_indexes, packed_key_value_indexes, key_values_lens):
    """Bagel.forward_cache_update_text, bagel.py:844-874."""
    emb = p["language_model.model.embed_tokens.weight"][packed_text_ids]
    _, past = llm_forward_inference(p, dims, emb, text_token_lens, packed_text_position_ids,
                                    packed_text_indexes, past, key_values_lens, packed_key_value_indexes,
                                    update_past_key_values=True, is_causal=True, mode="und")
    return past


def prepare_start_tokens(curr_kvlens, curr_rope, new_token_ids):
    """Bagel.prepare_start_tokens, bagel.py:1502-1523."""
    toks, kvidx, pos = [], [], []
    curr = 0
    for kvlen, rope in zip(curr_kvlens, curr_rope):
        kvidx.extend(range(curr, curr + kvlen))
        toks.append(new_token_ids["bos_token_id"])
        pos.append(rope)
        curr += kvlen
    return {
        "packed_start_tokens": torch.tensor(toks, dtype=torch.long),
        "packed_query_position_ids": torch.tensor(pos, dtype=torch.long),
        "key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
        "packed_key_value_indexes": torch.tensor(kvidx, dtype=torch.long),
    }


def generate_text(p, dims, past, packed_key_value_indexes, key_values_lens, packed_start_tokens,
                  packed_query_position_ids, max_length, end_token_id=None, logits_trace=None):
    """Bagel.generate_text with do_sample=False (greedy), bagel.py:1526-1596."""
    step = 0
    seq = []
    curr = packed_start_tokens
    while step < max_length:
        seq.append(curr)
        emb = p["language_model.model.embed_tokens.weight"][curr]
        query_lens = torch.ones_like(curr)
        q_idx = torch.cumsum(key_values_lens, dim=0) + torch.arange(0, len(key_values_lens), dtype=key_values_lens.dtype)
        unp = list(packed_key_value_indexes.split(key_values_lens.tolist(), dim=0))
        for i in range(len(unp)):
            unp[i] = unp[i] + i
        packed_key_value_indexes = torch.cat(unp, dim=0)
        out, past = llm_forward_inference(p, dims, emb, query_lens, packed_query_position_ids, q_idx, past,
                                          key_values_lens, packed_key_value_indexes, update_past_key_values=True,
                                          is_causal=True, mode="und")
        logits = linear(out, p["language_model.lm_head.weight"])
        if logits_trace is not None:
            logits_trace.append(logits)
        curr = torch.argmax(logits, dim=-1)
        unp = list(packed_key_value_indexes.split(key_values_lens.tolist(), dim=0))
        for i in range(len(unp)):
            unp[i] = torch.cat([unp[i], torch.tensor([unp[i][-1] + 1])], dim=0)
        packed_key_value_indexes = torch.cat(unp, dim=0)
        key_values_lens = key_values_lens + 1
        packed_query_position_ids = packed_query_position_ids + 1
        step += 1
        if end_token_id is not None and curr[0] == end_token_id:
            break
    return torch.stack(seq, dim=0)


def prepare_vae_latent(dims, curr_kvlens, curr_rope, image_sizes, new_token_ids):
    """Bagel.prepare_vae_latent, bagel.py:1145-1201 (one CPU randn per image, in order)."""
    text_ids, text_idx, vae_pos, vae_idx, noises = [], [], [], [], []
    pos_ids, seqlens, idx, kvidx = [], [], [], []
    qc = c = 0
    for (H, W), kvlen, rope in zip(image_sizes, curr_kvlens, curr_rope):
        kvidx.extend(range(c, c + kvlen))
        c += kvlen
        text_ids.append(new_token_ids["start_of_image"])
        text_idx.append(qc)
        idx.append(c)
        c += 1
        qc += 1
        vae_pos.append(flattened_position_ids_extrapolate(H, W, dims.latent_downsample, dims.max_latent_size))
        h, w = H // dims.latent_downsample, W // dims.latent_downsample
        n = h * w
        noises.append(torch.randn(n, dims.patch_latent_dim))
        vae_idx.extend(range(qc, qc + n))
        idx.extend(range(c, c + n))
        c += n
        qc += n
        text_ids.append(new_token_ids["end_of_image"])
        text_idx.append(qc)
        idx.append(c)
        c += 1
        qc += 1
        pos_ids.extend([rope] * (n + 2))
        seqlens.append(n + 2)
    return {
        "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
        "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
        "packed_init_noises": torch.cat(noises, dim=0),
        "packed_vae_position_ids": torch.cat(vae_pos, dim=0),
        "packed_vae_token_indexes": torch.tensor(vae_idx, dtype=torch.long),
        "packed_seqlens": torch.tensor(seqlens, dtype=torch.int),
        "packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
        "key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
        "packed_indexes": torch.tensor(idx, dtype=torch.long),
        "packed_key_value_indexes": torch.tensor(kvidx, dtype=torch.long),
    }


def prepare_vae_latent_cfg(dims, curr_kvlens, curr_rope, image_sizes):
    """Bagel.prepare_vae_latent_cfg, bagel.py:1203-1234."""
    pos_ids, idx, kvidx = [], [], []
    c = 0
    for (H, W), kvlen, rope in zip(image_sizes, curr_kvlens, curr_rope):
        kvidx.extend(range(c, c + kvlen))
        c += kvlen
        n = (H // dims.latent_downsample) * (W // dims.latent_downsample)
        idx.extend(range(c, c + n + 2))
        c += n + 2
        pos_ids.extend([rope] * (n + 2))
    return {
        "cfg_packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
        "cfg_key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
        "cfg_packed_query_indexes": torch.tensor(idx, dtype=torch.long),
        "cfg_packed_key_value_indexes": torch.tensor(kvidx, dtype=torch.long),
    }


# ----------------------------------------------------------------------------
# flow step and sampler
# ----------------------------------------------------------------------------
def cfg_combine(v_t, cfg_text_v_t, cfg_img_v_t, cfg_text_scale, cfg_img_scale, cfg_renorm_min, cfg_renorm_type):
    """CFG combine + renorm, bagel.py:1466-1500."""
    if cfg_text_scale > 1.0:
        if cfg_renorm_type == "text_channel":
            v_text_ = cfg_text_v_t + cfg_text_scale * (v_t - cfg_text_v_t)
            n0 = torch.norm(v_t, dim=-1, keepdim=True)
            n1 = torch.norm(v_text_, dim=-1, keepdim=True)
            scale = (n0 / (n1 + 1e-8)).clamp(min=cfg_renorm_min, max=1.0)
            v_text = v_text_ * scale
            if cfg_img_scale > 1.0:
                v_t = cfg_img_v_t + cfg_img_scale * (v_text - cfg_img_v_t)
            else:
                v_t = v_text
        else:
            v_text_ = cfg_text_v_t + cfg_text_scale * (v_t - cfg_text_v_t)
            if cfg_img_scale > 1.0:
                v_ = cfg_img_v_t + cfg_img_scale * (v_text_ - cfg_img_v_t)
            else:
                v_ = v_text_
            if cfg_renorm_type == "global":
                n0 = torch.norm(v_t)
                n1 = torch.norm(v_)
            elif cfg_renorm_type == "channel":
                n0 = torch.norm(v_t, dim=-1, keepdim=True)
                n1 = torch.norm(v_, dim=-1, keepdim=True)
            else:
                raise NotImplementedError(f"{cfg_renorm_type} is not suppoprted")
            scale = (n0 / (n1 + 1e-8)).clamp(min=cfg_renorm_min, max=1.0)
            v_t = v_ * scale
    return v_t


def forward_flow(p, dims, x_t, timestep, packed_vae_token_indexes, packed_vae_position_ids, packed_text_ids,
                 packed_text_indexes, packed_indexes, packed_position_ids, packed_seqlens, key_values_lens,
                 past_key_values, packed_key_value_indexes, cfg_renorm_min=0.0, cfg_renorm_type="global",
                 cfg_text_scale=1.0, cfg_text_packed_position_ids=None, cfg_text_packed_query_indexes=None,
                 cfg_text_key_values_lens=None, cfg_text_past_key_values=None,
                 cfg_text_packed_key_value_indexes=None,
                 cfg_img_scale=1.0, cfg_img_packed_position_ids=None, cfg_img_packed_query_indexes=None,
                 cfg_img_key_values_lens=None, cfg_img_past_key_values=None,
                 cfg_img_packed_key_value_indexes=None, return_branches=False, taylor=None):
    """Bagel._forward_flow, bagel.py:1350-1500.  ``taylor``: {"cond", "text", "img"} -> TaylorState."""
    emb = p["language_model.model.embed_tokens.weight"][packed_text_ids]
    seq = emb.new_zeros((int(sum(packed_seqlens)), dims.hidden_size))
    seq[packed_text_indexes] = emb
    assert timestep.unique().shape[0] == 1
    pos = p["latent_pos_embed.pos_embed"][packed_vae_position_ids]
    temb = time_embedder(p, timestep, dims.timestep_freq_dim)
    h = linear(x_t, p["vae2llm.weight"], p["vae2llm.bias"]) + temb + pos
    seq[packed_vae_token_indexes] = h.to(seq.dtype)

    def branch(pos_ids, q_idx, past, kvl, kv_idx, which):
        out, _ = llm_forward_inference(p, dims, seq, packed_seqlens, pos_ids, q_idx, past, kvl, kv_idx,
                                       update_past_key_values=False, is_causal=False, mode="gen",
                                       packed_vae_token_indexes=packed_vae_token_indexes,
                                       packed_text_indexes=packed_text_indexes,
                                       taylor=None if taylor is None else taylor[which])
        return linear(out, p["llm2vae.weight"], p["llm2vae.bias"])[packed_vae_token_indexes]

    v_t = branch(packed_position_ids, packed_indexes, past_key_values, key_values_lens, packed_key_value_indexes, "cond")
    v_text = v_img = None
    if cfg_text_scale > 1.0:
        v_text = branch(cfg_text_packed_position_ids, cfg_text_packed_query_indexes, cfg_text_past_key_values,
                        cfg_text_key_values_lens, cfg_text_packed_key_value_indexes, "text")
    if cfg_img_scale > 1.0:
        v_img = branch(cfg_img_packed_position_ids, cfg_img_packed_query_indexes, cfg_img_past_key_values,
                       cfg_img_key_values_lens, cfg_img_packed_key_value_indexes, "img")
    out = cfg_combine(v_t, v_text, v_img, cfg_text_scale, cfg_img_scale, cfg_renorm_min, cfg_renorm_type)
    if return_branches:
        return out, (v_t, v_text, v_img)
    return out


def flow_timesteps(num_timesteps: int, timestep_shift: float):
    """bagel.py:1286-1289."""
    t = torch.linspace(1, 0, num_timesteps)
    t = timestep_shift * t / (1 + (timestep_shift - 1) * t)
    return t[:-1], t[:-1] - t[1:]


def generate_image(p, dims, packed_text_ids, packed_text_indexes, packed_init_noises, packed_vae_position_ids,
                   packed_vae_token_indexes, packed_seqlens, packed_position_ids, packed_indexes,
                   past_key_values, key_values_lens, packed_key_value_indexes, num_timesteps=24,
                   timestep_shift=1.0, cfg_renorm_min=0.0, cfg_renorm_type="global", cfg_interval=(0, 1),
                   cfg_text_scale=1.0, cfg_img_scale=1.0, trace: Optional[list] = None, max_steps=None,
                   enable_taylorseer=False, teacher_x=None, **cfg_kwargs):
    """Bagel.generate_image, bagel.py:1237-1347.  ``trace`` collects per-step tensors; ``teacher_x`` replaces the
    state before every step (a list of per-step inputs)."""
    x_t = packed_init_noises
    taylor = {k: TaylorState(num_timesteps) for k in ("cond", "text", "img")} if enable_taylorseer else None
    timesteps, dts = flow_timesteps(num_timesteps, timestep_shift)
    for i, t in enumerate(timesteps):
        if max_steps is not None and i >= max_steps:
            break
        if teacher_x is not None:
            x_t = teacher_x[i]
        timestep = torch.tensor([t] * x_t.shape[0])
        if t > cfg_interval[0] and t <= cfg_interval[1]:
            ts_, is_ = cfg_text_scale, cfg_img_scale
        else:
            ts_, is_ = 1.0, 1.0
        v_t, br = forward_flow(p, dims, x_t, timestep, packed_vae_token_indexes, packed_vae_position_ids,
                               packed_text_ids, packed_text_indexes, packed_indexes, packed_position_ids,
                               packed_seqlens, key_values_lens, past_key_values, packed_key_value_indexes,
                               cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type,
                               cfg_text_scale=ts_, cfg_img_scale=is_, return_branches=True, taylor=taylor, **cfg_kwargs)
        if trace is not None:
            trace.append({"x_t": x_t, "v_t": v_t, "v_cond": br[0], "v_text": br[1], "v_img": br[2]})
        x_t = x_t - v_t.to(x_t.device) * dts[i]
    return x_t.split((packed_seqlens - 2).tolist())


# ----------------------------------------------------------------------------
# FLUX VAE decoder
# ----------------------------------------------------------------------------
def _conv(x, p, name, padding, bf16):
    w, b = p[name + ".weight"], p[name + ".bias"]
    if bf16:
        return F.conv2d(x.to(BF16), w.to(BF16), b.to(BF16), padding=padding)
    return F.conv2d(x, w, b, padding=padding)


def _gn_swish(x, p, name, swish=True):
    """GroupNorm(32, eps 1e-6) evaluated in fp32 (autocast keeps normalisation in fp32), then x*sigmoid(x)."""
    h = F.group_norm(x.float(), 32, p[name + ".weight"].float(), p[name + ".bias"].float(), eps=1e-6)
    return h * torch.sigmoid(h) if swish else h


def _resblock(x, p, name, bf16):
    """ResnetBlock.forward, modeling/autoencoder.py:82-95."""
    h = _conv(_gn_swish(x, p, name + ".norm1"), p, name + ".conv1", 1, bf16)
    h = _conv(_gn_swish(h, p, name + ".norm2"), p, name + ".conv2", 1, bf16)
    if (name + ".nin_shortcut.weight") in p:
        x = _conv(x, p, name + ".nin_shortcut", 0, bf16)
    return x + h


def _attnblock(x, p, name, bf16):
    """AttnBlock.forward, autoencoder.py:50-65: single-head attention over h*w tokens, head dim = channels."""
    h = _gn_swish(x, p, name + ".norm", swish=False)
    q = _conv(h, p, name + ".q", 0, bf16)
    k = _conv(h, p, name + ".k", 0, bf16)
    v = _conv(h, p, name + ".v", 0, bf16)
    b, c, hh, ww = q.shape
    q, k, v = (t.reshape(b, 1, c, hh * ww).transpose(2, 3).contiguous() for t in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(2, 3).reshape(b, c, hh, ww)
    return x + _conv(o, p, name + ".proj_out", 0, bf16)


def vae_decode(p, vdims, z: torch.Tensor, bf16: bool = True) -> torch.Tensor:
    """AutoEncoder.decode + Decoder.forward, autoencoder.py:250-272,320-322.

    ``bf16`` = convolutions and attention on bf16 operands (what bf16 autocast does to the fp32 VAE).
    """
    z = z / vdims.scale_factor + vdims.shift_factor
    h = _conv(z, p, "decoder.conv_in", 1, bf16)
    h = _resblock(h, p, "decoder.mid.block_1", bf16)
    h = _attnblock(h, p, "decoder.mid.attn_1", bf16)
    h = _resblock(h, p, "decoder.mid.block_2", bf16)
    nres = len(vdims.ch_mult)
    for lvl in reversed(range(nres)):
        for b in range(vdims.num_res_blocks + 1):
            h = _resblock(h, p, f"decoder.up.{lvl}.block.{b}", bf16)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, p, f"decoder.up.{lvl}.upsample.conv", 1, bf16)
    h = _gn_swish(h, p, "decoder.norm_out")
    return _conv(h, p, "decoder.conv_out", 1, bf16)


def unpatchify_latent(dims, latent: torch.Tensor, image_shape) -> torch.Tensor:
    """InterleaveInferencer.decode_image, inferencer.py:201-207."""
    H, W = image_shape
    h, w = H // dims.latent_downsample, W // dims.latent_downsample
    p_, c = dims.latent_patch_size, dims.latent_channel
    latent = latent.reshape(1, h, w, p_, p_, c)
    latent = torch.einsum("nhwpqc->nchpwq", latent)
    return latent.reshape(1, c, h * p_, w * p_)


def to_uint8_image(image: torch.Tensor) -> torch.Tensor:
    """inferencer.py:209-210."""
    return ((image * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8)


# ----------------------------------------------------------------------------
# SigLIP NaViT vision tower + connector (image understanding tokens)
# ----------------------------------------------------------------------------
def layer_norm(x, w, b, eps):
    """nn.LayerNorm under autocast: fp32 statistics and affine; the consumer rounds the result to bf16."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def gelu_tanh(x):
    """ACT2FN["gelu_pytorch_tanh"] on a bf16 tensor: fp32 math, result in the input dtype."""
    return F.gelu(x.float(), approximate="tanh").to(x.dtype)


def patchify(image, patch_size):
    """data/data_utils.py:43-50."""
    c, h, w = image.shape
    q = patch_size
    image = image.reshape(c, h // q, q, w // q, q)
    return torch.einsum("chpwq->hwpqc", image).reshape(-1, q * q * c)


def siglip_forward(p, vit, packed_pixel_values, packed_flattened_position_ids, seqlens):
    """SiglipVisionModel.forward with rope off and the patch embedding converted to a linear layer
    (siglip_navit.py:167-195, 202-244, 255-300, 345-371).  ``vit`` is a ViTDims."""
    pre = "vit_model.vision_model."
    Hh, D = vit.num_heads, vit.hidden_size // vit.num_heads
    x = linear(packed_pixel_values, p[pre + "embeddings.patch_embedding.weight"], p[pre + "embeddings.patch_embedding.bias"])
    x = x + p[pre + "embeddings.position_embedding.weight"].to(BF16)[packed_flattened_position_ids]
    lens = [int(n) for n in seqlens]
    for i in range(vit.num_layers):
        l = pre + f"encoder.layers.{i}."
        h = layer_norm(x, p[l + "layer_norm1.weight"], p[l + "layer_norm1.bias"], vit.layer_norm_eps)
        q = linear(h, p[l + "self_attn.q_proj.weight"], p[l + "self_attn.q_proj.bias"]).view(-1, Hh, D)
        k = linear(h, p[l + "self_attn.k_proj.weight"], p[l + "self_attn.k_proj.bias"]).view(-1, Hh, D)
        v = linear(h, p[l + "self_attn.v_proj.weight"], p[l + "self_attn.v_proj.bias"]).view(-1, Hh, D)
        o = varlen_attention(q, k, v, lens, lens, causal=False).reshape(-1, Hh * D)
        x = x + linear(o, p[l + "self_attn.out_proj.weight"], p[l + "self_attn.out_proj.bias"])
        h = layer_norm(x, p[l + "layer_norm2.weight"], p[l + "layer_norm2.bias"], vit.layer_norm_eps)
        h = gelu_tanh(linear(h, p[l + "mlp.fc1.weight"], p[l + "mlp.fc1.bias"]))
        x = x + linear(h, p[l + "mlp.fc2.weight"], p[l + "mlp.fc2.bias"])
    return layer_norm(x, p[pre + "post_layernorm.weight"], p[pre + "post_layernorm.bias"], vit.layer_norm_eps)


def vit_token_embed(p, vit, packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens):
    """vit_model -> connector -> + vit_pos_embed, bagel.py:974-987 (MLPconnector: modeling_utils.py:115-126)."""
    h = siglip_forward(p, vit, packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens)
    h = gelu_tanh(linear(h, p["connector.fc1.weight"], p["connector.fc1.bias"]))
    h = linear(h, p["connector.fc2.weight"], p["connector.fc2.bias"])
    return h + p["vit_pos_embed.pos_embed"].to(BF16)[packed_vit_position_ids]


def _prepare_image_tokens(curr_kvlens, curr_rope, token_counts, new_token_ids):
    """Index bookkeeping shared by prepare_vit_images and prepare_vae_images (bagel.py:883-920, 1017-1058)."""
    img_idx, text_ids, text_idx, seqlens, pos_ids, idx, kvidx = [], [], [], [], [], [], []
    _curr = curr = 0
    newlens, new_rope = [], []
    for n, kvlen, rope in zip(token_counts, curr_kvlens, curr_rope):
        kvidx.extend(range(curr, curr + kvlen))
        curr += kvlen
        text_ids.append(new_token_ids["start_of_image"])
        text_idx.append(_curr)
        idx.append(curr)
        curr += 1
        _curr += 1
        img_idx.extend(range(_curr, _curr + n))
        idx.extend(range(curr, curr + n))
        curr += n
        _curr += n
        text_ids.append(new_token_ids["end_of_image"])
        text_idx.append(_curr)
        idx.append(curr)
        curr += 1
        _curr += 1
        pos_ids.extend([rope] * (n + 2))
        seqlens.append(n + 2)
        newlens.append(kvlen + n + 2)
        new_rope.append(rope + 1)
    gi = {
        "packed_text_ids": torch.tensor(text_ids, dtype=torch.long),
        "packed_text_indexes": torch.tensor(text_idx, dtype=torch.long),
        "packed_position_ids": torch.tensor(pos_ids, dtype=torch.long),
        "packed_seqlens": torch.tensor(seqlens, dtype=torch.int),
        "packed_indexes": torch.tensor(idx, dtype=torch.long),
        "packed_key_value_indexes": torch.tensor(kvidx, dtype=torch.long),
        "key_values_lens": torch.tensor(curr_kvlens, dtype=torch.int),
    }
    return gi, img_idx, newlens, new_rope


def prepare_vit_images(dims, curr_kvlens, curr_rope, image_tensors, new_token_ids):
    """Bagel.prepare_vit_images, bagel.py:876-936, on already transformed [3, H, W] tensors."""
    vit = dims.vit
    toks = [patchify(t, vit.patch_size) for t in image_tensors]
    pos = [flattened_position_ids_extrapolate(t.shape[1], t.shape[2], vit.patch_size, vit.max_patches_per_side)
           for t in image_tensors]
    gi, img_idx, newlens, new_rope = _prepare_image_tokens(curr_kvlens, curr_rope, [t.shape[0] for t in toks], new_token_ids)
    gi.update({
        "vit_token_seqlens": torch.tensor([t.shape[0] for t in toks], dtype=torch.int),
        "packed_vit_tokens": torch.cat(toks, dim=0),
        "packed_vit_position_ids": torch.cat(pos, dim=0),
        "packed_vit_token_indexes": torch.tensor(img_idx, dtype=torch.long),
    })
    return gi, newlens, new_rope


def forward_cache_update_vit(p, dims, past, packed_text_ids, packed_text_indexes, packed_vit_tokens,
                             packed_vit_token_indexes, packed_vit_position_ids, vit_token_seqlens, packed_position_ids,
                             packed_seqlens, packed_indexes, packed_key_value_indexes, key_values_lens, embed_trace=None):
    """Bagel.forward_cache_update_vit, bagel.py:938-1008."""
    emb = p["language_model.model.embed_tokens.weight"][packed_text_ids]
    seq = emb.new_zeros((int(sum(packed_seqlens)), dims.hidden_size))
    seq[packed_text_indexes] = emb
    ve = vit_token_embed(p, dims.vit, packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens).to(seq.dtype)
    if embed_trace is not None:
        embed_trace.append(ve)
    seq[packed_vit_token_indexes] = ve
    _, past = llm_forward_inference(p, dims, seq, packed_seqlens, packed_position_ids, packed_indexes, past,
                                    key_values_lens, packed_key_value_indexes, update_past_key_values=True,
                                    is_causal=False, mode="und")
    return past


# ----------------------------------------------------------------------------
# FLUX VAE encoder and the VAE-token prefill
# ----------------------------------------------------------------------------
def vae_encode_moments(p, vdims, x: torch.Tensor, bf16: bool = True) -> torch.Tensor:
    """Encoder.forward, autoencoder.py:172-193: [B, 3, H, W] -> [B, 2 z, H/8, W/8] (mean | logvar)."""
    h = _conv(x, p, "encoder.conv_in", 1, bf16)
    nres = len(vdims.ch_mult)
    for lvl in range(nres):
        for b in range(vdims.num_res_blocks):
            h = _resblock(h, p, f"encoder.down.{lvl}.block.{b}", bf16)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)       # Downsample, autoencoder.py:104-108
            w, b_ = p[f"encoder.down.{lvl}.downsample.conv.weight"], p[f"encoder.down.{lvl}.downsample.conv.bias"]
            h = F.conv2d(h.to(BF16), w.to(BF16), b_.to(BF16), stride=2) if bf16 else F.conv2d(h, w, b_, stride=2)
    h = _resblock(h, p, "encoder.mid.block_1", bf16)
    h = _attnblock(h, p, "encoder.mid.attn_1", bf16)
    h = _resblock(h, p, "encoder.mid.block_2", bf16)
    h = _gn_swish(h, p, "encoder.norm_out")
    return _conv(h, p, "encoder.conv_out", 1, bf16)


def vae_encode(p, vdims, x, noise, bf16: bool = True):
    """AutoEncoder.encode, autoencoder.py:275-287, 315-318, with the Gaussian draw given by the caller."""
    mean, logvar = torch.chunk(vae_encode_moments(p, vdims, x, bf16), 2, dim=1)
    z = mean + torch.exp(0.5 * logvar) * noise.to(mean.dtype)
    return vdims.scale_factor * (z - vdims.shift_factor)


def prepare_vae_images(dims, curr_kvlens, curr_rope, image_tensors, new_token_ids, timestep=0):
    """Bagel.prepare_vae_images, bagel.py:1010-1081, on already transformed [3, H, W] tensors."""
    ds = dims.latent_downsample
    shapes = [(t.shape[1] // ds, t.shape[2] // ds) for t in image_tensors]
    pos = [flattened_position_ids_extrapolate(t.shape[1], t.shape[2], ds, dims.max_latent_size) for t in image_tensors]
    gi, img_idx, newlens, new_rope = _prepare_image_tokens(curr_kvlens, curr_rope, [h * w for h, w in shapes], new_token_ids)
    size = [max(s) for s in zip(*[t.shape for t in image_tensors])]
    padded = torch.zeros((len(image_tensors), *size))
    for i, t in enumerate(image_tensors):
        padded[i, :, :t.shape[1], :t.shape[2]] = t
    gi.update({
        "padded_images": padded,
        "patchified_vae_latent_shapes": shapes,
        "packed_vae_position_ids": torch.cat(pos, dim=0),
        "packed_timesteps": torch.tensor([timestep]),
        "packed_vae_token_indexes": torch.tensor(img_idx, dtype=torch.long),
    })
    return gi, newlens, new_rope


def forward_cache_update_vae(p, vae_p, dims, past, padded_images, patchified_vae_latent_shapes, packed_vae_position_ids,
                             packed_timesteps, packed_vae_token_indexes, packed_text_ids, packed_text_indexes,
                             packed_position_ids, packed_seqlens, packed_indexes, key_values_lens,
                             packed_key_value_indexes, noise, latent_trace=None):
    """Bagel.forward_cache_update_vae, bagel.py:1083-1143."""
    emb = p["language_model.model.embed_tokens.weight"][packed_text_ids]
    seq = emb.new_zeros((int(sum(packed_seqlens)), dims.hidden_size))
    seq[packed_text_indexes] = emb
    padded_latent = vae_encode(vae_p, dims.vae, padded_images, noise)
    q, c = dims.latent_patch_size, dims.latent_channel
    packed = []
    for latent, (h, w) in zip(padded_latent, patchified_vae_latent_shapes):
        latent = latent[:, :h * q, :w * q].reshape(c, h, q, w, q)
        packed.append(torch.einsum("chpwq->hwpqc", latent).reshape(-1, q * q * c))
    packed = torch.cat(packed, dim=0)
    if latent_trace is not None:
        latent_trace.append(packed)
    t = packed_timesteps
    half = dims.timestep_freq_dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(t.dtype)
    args = t[:, None] * freqs[None]                       # modeling_utils.py:98-103 (freqs take t's dtype)
    f = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    te = linear(F.silu(linear(f, p["time_embedder.mlp.0.weight"], p["time_embedder.mlp.0.bias"])),
                p["time_embedder.mlp.2.weight"], p["time_embedder.mlp.2.bias"])
    lat = linear(packed, p["vae2llm.weight"], p["vae2llm.bias"]) + te + p["latent_pos_embed.pos_embed"][packed_vae_position_ids]
    seq[packed_vae_token_indexes] = lat.to(seq.dtype)
    _, past = llm_forward_inference(p, dims, seq, packed_seqlens, packed_position_ids, packed_indexes, past,
                                    key_values_lens, packed_key_value_indexes, update_past_key_values=True,
                                    is_causal=False, mode="gen", packed_vae_token_indexes=packed_vae_token_indexes,
                                    packed_text_indexes=packed_text_indexes)
    return past
