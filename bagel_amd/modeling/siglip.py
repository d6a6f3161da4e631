"""SigLIP NaViT vision tower + MLP connector on HIP kernels (image-understanding tokens of the context).

Follows ``SiglipVisionModel.forward`` as BAGEL configures it (rope off, patch embedding in its linear
form: modeling/bagel/siglip_navit.py:167-195, 202-244, 255-300, 345-371), then ``MLPconnector`` and
``vit_pos_embed`` (modeling/bagel/bagel.py:974-987, modeling_utils.py:115-126).

The tower's heads are 72 wide.  The attention kernel works on 64- or 128-wide heads, so q, k and v are
zero-padded to 128 when they are split out of the fused projection (zero columns change neither q.k nor
P.V) and the softmax keeps its 1/sqrt(72); the output projection's weight gets matching zero columns.
The MLP's intermediate size (4304) is not a multiple of the GEMM's K step, so fc2's weight is zero-padded
along K and fc1 writes into a buffer whose pad columns stay zero.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch

from .. import _lib as K
from ..config import ViTDims

BF16 = torch.bfloat16
PAD_D = 128


def _pad_cols(w: torch.Tensor, cols: int) -> torch.Tensor:
    out = torch.zeros((w.shape[0], cols), dtype=w.dtype, device=w.device)
    out[:, :w.shape[1]] = w
    return out


class SiglipTower:
    def __init__(self, vit: ViTDims, llm_hidden: int, sd: Dict[str, torch.Tensor], device):
        K.lib()
        self.vit, self.H, self.device = vit, llm_hidden, torch.device(device)
        assert vit.head_dim % 8 == 0 and vit.head_dim <= PAD_D, "head width must be a multiple of 8 and at most 128"
        assert vit.hidden_size % 64 == 0, "tower width must be a multiple of 64"
        dev = self.device
        Hv, Iv, Hh, D = vit.hidden_size, vit.intermediate_size, vit.num_heads, vit.head_dim

        def g(name):
            return sd[name].detach().to(device=dev, dtype=BF16).contiguous()

        pre = "vit_model.vision_model."
        self.kpatch = (vit.patch_dim + 63) // 64 * 64
        self.ipad = (Iv + 63) // 64 * 64
        self.patch_w = _pad_cols(g(pre + "embeddings.patch_embedding.weight"), self.kpatch)
        self.patch_b = g(pre + "embeddings.patch_embedding.bias")
        self.pos = g(pre + "embeddings.position_embedding.weight")
        self.layers: List[dict] = []
        for i in range(vit.num_layers):
            l = pre + f"encoder.layers.{i}."
            a = l + "self_attn."
            wo = g(a + "out_proj.weight").view(Hv, Hh, D)
            wo_pad = torch.zeros((Hv, Hh, PAD_D), dtype=BF16, device=dev)
            wo_pad[:, :, :D] = wo
            self.layers.append(dict(
                ln1_w=g(l + "layer_norm1.weight"), ln1_b=g(l + "layer_norm1.bias"),
                wqkv=torch.cat([g(a + "q_proj.weight"), g(a + "k_proj.weight"), g(a + "v_proj.weight")]).contiguous(),
                bqkv=torch.cat([g(a + "q_proj.bias"), g(a + "k_proj.bias"), g(a + "v_proj.bias")]).contiguous(),
                wo=wo_pad.view(Hv, Hh * PAD_D).contiguous(), bo=g(a + "out_proj.bias"),
                ln2_w=g(l + "layer_norm2.weight"), ln2_b=g(l + "layer_norm2.bias"),
                fc1_w=g(l + "mlp.fc1.weight"), fc1_b=g(l + "mlp.fc1.bias"),
                fc2_w=_pad_cols(g(l + "mlp.fc2.weight"), self.ipad), fc2_b=g(l + "mlp.fc2.bias")))
        self.post_w, self.post_b = g(pre + "post_layernorm.weight"), g(pre + "post_layernorm.bias")
        self.c1_w, self.c1_b = g("connector.fc1.weight"), g("connector.fc1.bias")
        self.c2_w, self.c2_b = g("connector.fc2.weight"), g("connector.fc2.bias")
        self.vit_pos_embed = g("vit_pos_embed.pos_embed")
        self._buf: Dict[str, torch.Tensor] = {}

    def _get(self, name, shape, dtype=BF16):
        n = 1
        for s in shape:
            n *= s
        cur = self._buf.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            cur = torch.empty(max(n, 1), device=self.device, dtype=dtype)
            self._buf[name] = cur
        return cur[:n].view(*shape)

    @K.on_device
    @torch.no_grad()
    def tower(self, packed_pixel_values: torch.Tensor, position_ids: torch.Tensor, seqlens: Sequence[int]) -> torch.Tensor:
        """SiglipVisionModel.forward: [N, 3 p p] pixels -> [N, Hv] (post_layernorm output, bf16)."""
        vit, dev = self.vit, self.device
        Hv, Hh, D = vit.hidden_size, vit.num_heads, vit.head_dim
        N = int(packed_pixel_values.shape[0])
        assert N == sum(seqlens) and packed_pixel_values.shape[1] == vit.patch_dim
        tok32 = torch.zeros((N, self.kpatch), device=dev, dtype=torch.float32)
        tok32[:, :vit.patch_dim] = packed_pixel_values.to(dev, torch.float32)
        tok = self._get("tok", (N, self.kpatch))
        K.cast_f32_bf16(tok32, tok)
        ids = position_ids.to(torch.int32).to(dev)
        self._ids = ids
        R = self._get("R", (N, Hv))
        K.gather_rows(self.pos, ids, R, N)
        X = self._get("X", (N, Hv))
        K.gemm(tok, self.patch_w, X, bias=self.patch_b, residual=R, epilogue=K.EPI_BIAS_RESIDUAL)

        # key slots: one 64-aligned range per image; one attention tile per 128 query rows
        kv_start, tiles, slots = [], [], []
        s = r = 0
        for n in seqlens:
            kv_start.append(s)
            slots += list(range(s, s + n))
            for q0 in range(0, n, 128):
                tiles.append((r + q0, min(128, n - q0), s, n, -1, 0, 0, 0))
            s += (n + 63) // 64 * 64
            r += n
        S = s
        row_slot = torch.tensor(slots, dtype=torch.int32, device=dev)
        tiles_t = torch.tensor(tiles, dtype=torch.int32, device=dev)
        kws = self._get("kws", (Hh, S, PAD_D))
        vtws = self._get("vtws", (Hh, PAD_D, S))
        kws.zero_()
        vtws.zero_()
        Xn = self._get("Xn", (N, Hv))
        QKV = self._get("QKV", (N, 3 * Hv))
        Q = self._get("Q", (N, Hh * PAD_D))
        O = self._get("O", (N, Hh * PAD_D))
        Hm = self._get("Hm", (N, self.ipad))
        Hm.zero_()
        eps = vit.layer_norm_eps
        scale = float(D) ** -0.5
        for lw in self.layers:
            K.layernorm(X, Xn, lw["ln1_w"], lw["ln1_b"], eps)
            K.gemm(Xn, lw["wqkv"], QKV, bias=lw["bqkv"])
            K.split_heads_pad(QKV, N, Hh, D, PAD_D, row_slot, Q, kws, vtws, S)
            K.varlen_attention(Q, kws, vtws, O, tiles_t, len(tiles), Hh, Hh, PAD_D, S, softmax_scale=scale)
            K.gemm(O, lw["wo"], X, bias=lw["bo"], residual=X, epilogue=K.EPI_BIAS_RESIDUAL)
            K.layernorm(X, Xn, lw["ln2_w"], lw["ln2_b"], eps)
            K.gemm(Xn, lw["fc1_w"], Hm, bias=lw["fc1_b"], epilogue=K.EPI_BIAS_GELU)
            K.gemm(Hm, lw["fc2_w"], X, bias=lw["fc2_b"], residual=X, epilogue=K.EPI_BIAS_RESIDUAL)
        K.layernorm(X, Xn, self.post_w, self.post_b, eps)
        return Xn

    @K.on_device
    @torch.no_grad()
    def embed(self, packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens) -> torch.Tensor:
        """vit_model -> connector -> + vit_pos_embed (bagel.py:977-987): [N, H] bf16 rows for the LLM."""
        seqlens = [int(n) for n in vit_token_seqlens]
        h = self.tower(packed_vit_tokens, packed_vit_position_ids, seqlens)
        N = h.shape[0]
        c1 = self._get("c1", (N, self.H))
        K.gemm(h, self.c1_w, c1, bias=self.c1_b, epilogue=K.EPI_BIAS_GELU)
        pe = self._get("pe", (N, self.H))
        K.gather_rows(self.vit_pos_embed, self._ids, pe, N)
        out = torch.empty((N, self.H), device=self.device, dtype=BF16)
        K.gemm(c1, self.c2_w, out, bias=self.c2_b, residual=pe, epilogue=K.EPI_BIAS_RESIDUAL)
        return out
