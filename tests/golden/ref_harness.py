"""Import the reference's PyTorch path on CPU (authoring container only).

Used by make_golden.py to capture golden vectors.  Nothing here edits the
reference; five shims make its modules importable with the libraries present:

1. ``transformers.modeling_utils`` is imported before the ``flash_attn`` stand-in
   is registered (transformers probes ``flash_attn.__spec__``).
2. ``flash_attn.flash_attn_varlen_func`` is replaced by its mathematical
   definition, evaluated per sample with ``F.scaled_dot_product_attention``.
3. ``llm_config.pad_token_id = None`` (read at qwen2_navit.py:960).
4. ``ROPE_INIT_FUNCTIONS["default"]`` is restored (looked up at modeling_qwen2.py:105).
5. The model is cast to bf16 and called under CPU bf16 autocast; the rotary
   ``inv_freq`` buffer is restored to fp32 afterwards (the canonical loader in
   app.py:105-113 never casts it).

The GPU box never sees /root/reference; only the captured tensors travel.
"""
from __future__ import annotations

import sys
import types

import torch
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


def _install_shims():
    import transformers.modeling_utils  # noqa: F401  (shim 1)
    import transformers.modeling_rope_utils as mru

    if "flash_attn" not in sys.modules:
        import importlib.machinery
        fa = types.ModuleType("flash_attn")
        fa.__spec__ = importlib.machinery.ModuleSpec("flash_attn", None)

        def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                                   causal=False, **kw):
            rep = q.shape[1] // k.shape[1]
            outs = []
            for i in range(len(cu_seqlens_q) - 1):
                qs = q[cu_seqlens_q[i]:cu_seqlens_q[i + 1]].transpose(0, 1)[None]
                ks = k[cu_seqlens_k[i]:cu_seqlens_k[i + 1]].transpose(0, 1)[None].repeat_interleave(rep, dim=1)
                vs = v[cu_seqlens_k[i]:cu_seqlens_k[i + 1]].transpose(0, 1)[None].repeat_interleave(rep, dim=1)
                lq, lk = qs.shape[2], ks.shape[2]
                mask = None
                if causal:
                    ii = torch.arange(lq)[:, None]
                    jj = torch.arange(lk)[None, :]
                    mask = jj <= ii + (lk - lq)
                o = F.scaled_dot_product_attention(qs, ks, vs, attn_mask=mask)
                outs.append(o[0].transpose(0, 1))
            return torch.cat(outs, dim=0)

        fa.flash_attn_varlen_func = flash_attn_varlen_func
        sys.modules["flash_attn"] = fa

    if "default" not in mru.ROPE_INIT_FUNCTIONS:
        def _default_rope(config, device=None, seq_len=None, **kw):
            base = config.rope_theta
            dim = config.hidden_size // config.num_attention_heads
            inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.int64).float().to(device) / dim))
            return inv_freq, 1.0
        mru.ROPE_INIT_FUNCTIONS["default"] = _default_rope

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


class StandInTokenizer:
    """``encode`` maps a string of space-separated integers to token ids."""

    def encode(self, text):
        return [int(t) for t in text.split()]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def build_reference_model(dims, state_dict, lean=False):
    """Reference ``Bagel`` (generation half only) carrying the given bf16 state dict.

    ``lean``: the modules are constructed on the meta device and the state dict's tensors become the parameters
    (``load_state_dict(assign=True)``), so a 28-layer model at BAGEL-7B width costs its 29 GB of bf16 weights
    once instead of an fp32 initialisation (58 GB) plus the state dict.  Same parameter values either way."""
    _install_shims()
    from modeling.bagel.qwen2_navit import Qwen2Config, Qwen2ForCausalLM
    from modeling.bagel.bagel import Bagel, BagelConfig
    from modeling.autoencoder import AutoEncoderParams

    llm_config = Qwen2Config(
        vocab_size=dims.vocab_size, hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size,
        num_hidden_layers=dims.num_layers, num_attention_heads=dims.num_heads,
        num_key_value_heads=dims.num_kv_heads, rms_norm_eps=dims.rms_norm_eps, rope_theta=dims.rope_theta,
        qk_norm=True, layer_module="Qwen2MoTDecoderLayer", tie_word_embeddings=False,
    )
    llm_config.pad_token_id = None  # shim 3
    v = dims.vae
    vae_config = AutoEncoderParams(resolution=v.resolution, in_channels=v.in_channels, downsample=v.downsample,
                                   ch=v.ch, out_ch=v.out_ch, ch_mult=list(v.ch_mult),
                                   num_res_blocks=v.num_res_blocks, z_channels=v.z_channels,
                                   scale_factor=v.scale_factor, shift_factor=v.shift_factor)
    vit_model, vit_config, extra = None, None, {}
    if dims.vit is not None:
        # the tower as app.py:45-48, 96-97 sets it up: rope off, patch embedding converted to a linear layer
        from modeling.bagel.siglip_navit import SiglipVisionConfig, SiglipVisionModel
        t = dims.vit
        vit_config = SiglipVisionConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
                                        num_hidden_layers=t.num_layers, num_attention_heads=t.num_heads,
                                        num_channels=t.num_channels, image_size=t.patch_size * t.max_patches_per_side,
                                        patch_size=t.patch_size, layer_norm_eps=t.layer_norm_eps, rope=False)
        vit_model = SiglipVisionModel(vit_config)
        vit_model.vision_model.embeddings.convert_conv2d_to_linear(vit_config)
        extra = dict(vit_max_num_patch_per_side=t.max_patches_per_side, connector_act="gelu_pytorch_tanh")
    config = BagelConfig(visual_gen=True, visual_und=dims.vit is not None, llm_config=llm_config, vit_config=vit_config,
                         vae_config=vae_config, latent_patch_size=dims.latent_patch_size,
                         max_latent_size=dims.max_latent_size, **extra)
    if lean:
        assert vit_model is None, "lean construction is for the generation half only"
        with torch.device("meta"):
            lm = Qwen2ForCausalLM(llm_config)
            model = Bagel(lm, vit_model, config)
        missing, unexpected = model.load_state_dict(state_dict, strict=False, assign=True)
        missing = [m for m in missing if "lm_head" not in m]
        assert not missing and not unexpected, (missing, unexpected)
        model.language_model.lm_head = None          # never called by the generation path; still on meta
        model = model.eval()
        assert all(p.dtype == torch.bfloat16 and not p.is_meta for p in model.parameters())
        # the rotary buffers were made on meta: same formula as the "default" rope initialiser above, fp32
        rot = model.language_model.model.rotary_emb
        D = dims.hidden_size // dims.num_heads
        inv_freq = 1.0 / (dims.rope_theta ** (torch.arange(0, D, 2, dtype=torch.int64).float() / D))
        rot.inv_freq = inv_freq
        rot.original_inv_freq = inv_freq.clone()
        leftovers = [n for n, b in model.named_buffers() if b.is_meta]
        assert not leftovers, leftovers
        return model
    lm = Qwen2ForCausalLM(llm_config)
    model = Bagel(lm, vit_model, config)
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    missing = [m for m in missing if "lm_head" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    model = model.to(torch.bfloat16).eval()  # shim 5 (first half)
    # ``.to(bf16)`` also rounds the rotary ``inv_freq`` buffer, which the canonical loader
    # (accelerate ``load_checkpoint_and_dispatch(dtype=bf16)``, app.py:105-113) leaves in fp32
    # because it only casts tensors that come from the checkpoint.  Put the fp32 values back.
    rot = model.language_model.model.rotary_emb
    rot.inv_freq = rot.original_inv_freq.float()
    return model


def build_reference_vae(vdims, state_dict):
    _install_shims()
    from modeling.autoencoder import AutoEncoder, AutoEncoderParams
    params = AutoEncoderParams(resolution=vdims.resolution, in_channels=vdims.in_channels,
                               downsample=vdims.downsample, ch=vdims.ch, out_ch=vdims.out_ch,
                               ch_mult=list(vdims.ch_mult), num_res_blocks=vdims.num_res_blocks,
                               z_channels=vdims.z_channels, scale_factor=vdims.scale_factor,
                               shift_factor=vdims.shift_factor)
    ae = AutoEncoder(params)
    missing, unexpected = ae.load_state_dict(state_dict, strict=False)
    if not any(k.startswith("encoder.") for k in state_dict):
        missing = [m for m in missing if not m.startswith("encoder.")]
    assert not missing and not unexpected, (missing, unexpected)
    return ae.eval()


def autocast():
    return torch.autocast("cpu", dtype=torch.bfloat16)  # shim 5 (second half)
