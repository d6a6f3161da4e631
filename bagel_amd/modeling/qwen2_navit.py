"""``model.language_model``: the surface of the reference's ``Qwen2ForCausalLM`` that inference callers touch
(modeling/bagel/qwen2_navit.py:1117-1211): ``forward_inference`` with the reference's argument names,
``model.embed_tokens`` and ``lm_head`` as callables, over the HIP engine.  ``Bagel``'s own methods do not go
through this class (they use specialised packed forwards); it exists for code written against the reference
that drives the language model directly (inferencer.py:333-429, 569-603).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib as K

BF16 = torch.bfloat16


@dataclass
class BaseNavitOutputWithPast:
    """qwen2_navit.py:231-234."""
    packed_query_sequence: torch.Tensor = None
    past_key_values: Optional[object] = None


class _EmbedTokens:
    def __init__(self, engine):
        self._engine = engine

    @property
    def weight(self):
        return self._engine.w.embed_tokens

    @K.on_device
    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        eng = self._engine
        flat = ids.reshape(-1).to(device=eng.device, dtype=torch.int32).contiguous()
        out = torch.empty((flat.numel(), eng.dims.hidden_size), device=eng.device, dtype=BF16)
        if flat.numel():
            K.gather_rows(eng.w.embed_tokens, flat, out, flat.numel())
        return out.view(*ids.shape, eng.dims.hidden_size)


class _LMHead:
    def __init__(self, engine):
        self._engine = engine

    @property
    def weight(self):
        return self._engine.w.lm_head

    @K.on_device
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        eng = self._engine
        if eng.w.lm_head is None:
            raise ValueError("the state dict held no language_model.lm_head.weight")
        rows = x.reshape(-1, x.shape[-1]).to(device=eng.device, dtype=BF16).contiguous()
        V = eng.w.lm_head.shape[0]
        out = torch.empty((rows.shape[0], V), device=eng.device, dtype=BF16)
        if rows.shape[0] <= 64:      # weight-streaming GEMM (HBM-bound), else the MFMA tile GEMM
            ws = eng._get("lm_head_ws", (16 * rows.shape[0] * V,), dtype=torch.float32)
            K.skinny_gemm(rows, eng.w.lm_head, out, ws)
        else:
            K.gemm(rows, eng.w.lm_head, out)
        return out.view(*x.shape[:-1], V)


class _Qwen2Model:
    def __init__(self, engine):
        self.embed_tokens = _EmbedTokens(engine)
        self.enable_taylorseer = False


class Qwen2ForCausalLM:
    def __init__(self, engine, cache_cls):
        self._engine = engine
        self._cache_cls = cache_cls
        self.model = _Qwen2Model(engine)
        self.lm_head = _LMHead(engine)

    @K.on_device
    @torch.no_grad()
    def forward_inference(self, packed_query_sequence, query_lens, packed_query_position_ids, packed_query_indexes,
                          past_key_values=None, key_values_lens=None, packed_key_value_indexes=None,
                          update_past_key_values=True, is_causal=True, mode="und", packed_vae_token_indexes=None,
                          packed_text_indexes=None) -> BaseNavitOutputWithPast:
        """qwen2_navit.py:1180-1211 -> Qwen2Model.forward_inference (:1041-1115)."""
        eng = self._engine
        L = eng.dims.num_layers
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        lens = [int(x) for x in cpu(query_lens).tolist()]
        M = sum(lens)
        if mode == "gen":
            # the reference asserts these (qwen2_navit.py:1072-1073)
            assert packed_vae_token_indexes is not None and packed_text_indexes is not None
            gen_rows = cpu(packed_vae_token_indexes).tolist()
            text_rows = cpu(packed_text_indexes).tolist()
            if sorted(gen_rows + text_rows) != list(range(M)):
                raise NotImplementedError("packed_vae_token_indexes and packed_text_indexes must partition the sequence")
        elif mode == "und":
            gen_rows = None
        else:
            raise ValueError(f"unknown mode {mode!r}")
        has_ctx = past_key_values is not None and past_key_values.key_cache[0] is not None
        kvl = [int(x) for x in cpu(key_values_lens).tolist()] if (has_ctx and key_values_lens is not None) else [0] * len(lens)
        qi = cpu(packed_query_indexes).tolist()
        ki = cpu(packed_key_value_indexes).tolist() if (has_ctx and packed_key_value_indexes is not None) else []
        ctx_k, ctx_v = [], []
        q = c = tot = 0
        for b, n in enumerate(lens):
            # merged layout [context of sample b | queries of sample b] (qwen2_navit.py:574-581)
            if ki[c:c + kvl[b]] != list(range(tot, tot + kvl[b])) or qi[q:q + n] != list(range(tot + kvl[b], tot + kvl[b] + n)):
                raise NotImplementedError("packed indexes must follow the contiguous per-sample layout "
                                          "[context | new tokens] that the prepare_* helpers produce")
            if kvl[b] > 0:
                ctx_k.append([past_key_values.key_cache[l][c:c + kvl[b]].to(eng.device, BF16) for l in range(L)])
                ctx_v.append([past_key_values.value_cache[l][c:c + kvl[b]].to(eng.device, BF16) for l in range(L)])
            else:
                ctx_k.append(None)
                ctx_v.append(None)
            q += n
            c += kvl[b]
            tot += kvl[b] + n
        hidden, keys, values = eng.forward_sequence(
            packed_query_sequence, lens, cpu(packed_query_position_ids).tolist(), kvl, ctx_k, ctx_v, bool(is_causal),
            gen_rows=gen_rows, need_kv=bool(update_past_key_values))
        if update_past_key_values:
            if past_key_values is None:
                past_key_values = self._cache_cls(L)
            for l in range(L):
                past_key_values.key_cache[l] = keys[l]
                past_key_values.value_cache[l] = values[l]
        return BaseNavitOutputWithPast(packed_query_sequence=hidden, past_key_values=past_key_values)
