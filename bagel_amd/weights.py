"""Deterministic synthetic weights keyed by the reference's parameter names.

No BAGEL checkpoint exists offline, so every parity check runs on seeded
random weights.  The generator is order independent: each tensor draws from its
own CPU ``torch.Generator`` seeded from (seed, crc32(name)), so the checker in
the authoring container, the HIP host code on the GPU box and the benchmark all
see identical values without shipping any weight file.

Key names follow the module tree of the reference
(/root/reference/modeling/bagel/bagel.py:62-95,
/root/reference/modeling/bagel/qwen2_navit.py:388-405,698-716,957-973,
/root/reference/modeling/autoencoder.py:196-248), which is also the layout of
``ema.safetensors`` / ``ae.safetensors``.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import numpy as np
import torch

from .config import BagelDims, VAEDims, ViTDims


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
    return g


def _randn(seed: int, name: str, shape, std: float, mean: float = 0.0, dtype=torch.bfloat16) -> torch.Tensor:
    t = torch.randn(*shape, generator=_gen(seed, name), dtype=torch.float32)
    return (t * std + mean).to(dtype)


def get_1d_sincos(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """/root/reference/modeling/bagel/modeling_utils.py:48-66 (float64 table)."""
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int) -> np.ndarray:
    """/root/reference/modeling/bagel/modeling_utils.py:24-45 (w goes first in the meshgrid)."""
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    emb_h = get_1d_sincos(embed_dim // 2, grid[0])
    emb_w = get_1d_sincos(embed_dim // 2, grid[1])
    return np.concatenate([emb_h, emb_w], axis=1)


def llm_layer_names(i: int) -> Iterator[Tuple[str, str]]:
    """(name, kind) for every parameter of decoder layer ``i``."""
    p = f"language_model.model.layers.{i}."
    for suf in ("", "_moe_gen"):
        yield p + f"self_attn.q_proj{suf}.weight", "wq"
        yield p + f"self_attn.q_proj{suf}.bias", "bq"
        yield p + f"self_attn.k_proj{suf}.weight", "wk"
        yield p + f"self_attn.k_proj{suf}.bias", "bk"
        yield p + f"self_attn.v_proj{suf}.weight", "wk"
        yield p + f"self_attn.v_proj{suf}.bias", "bk"
        yield p + f"self_attn.o_proj{suf}.weight", "wo"
        yield p + f"self_attn.q_norm{suf}.weight", "nd"
        yield p + f"self_attn.k_norm{suf}.weight", "nd"
        yield p + f"mlp{suf}.gate_proj.weight", "wi"
        yield p + f"mlp{suf}.up_proj.weight", "wi"
        yield p + f"mlp{suf}.down_proj.weight", "wd"
        yield p + f"input_layernorm{suf}.weight", "nh"
        yield p + f"post_attention_layernorm{suf}.weight", "nh"


def make_llm_state_dict(dims: BagelDims, seed: int = 0, device="cpu",
                        include_lm_head: bool = False, threads: int = 1) -> Dict[str, torch.Tensor]:
    """bf16 state dict of the language model + generation head.

    Matrices ~ N(0, 0.02); biases ~ N(0, 0.02); norm weights ~ N(1, 0.1) so that a
    mix-up between the understanding and generation experts' norms is visible;
    ``llm2vae`` (zero-initialised by the reference, bagel.py:97-100) is drawn
    with std 0.8/sqrt(hidden) so the velocity has RMS ~0.8 at every width, the
    scale of a real flow-matching velocity, which is what the absolute parity
    tolerance on velocities presumes.  ``threads`` > 1 draws the tensors in parallel
    (same values: each tensor has its own seeded generator).
    """
    H, I, D = dims.hidden_size, dims.intermediate_size, dims.head_dim
    shapes = {
        "wq": (dims.q_dim, H), "bq": (dims.q_dim,), "wk": (dims.kv_dim, H), "bk": (dims.kv_dim,),
        "wo": (H, dims.q_dim), "wi": (I, H), "wd": (H, I), "nd": (D,), "nh": (H,),
    }
    sd: Dict[str, torch.Tensor] = {}
    jobs = []

    def put(name, shape, std, mean=0.0):
        sd[name] = None                      # keeps the insertion order of the sequential generator
        jobs.append((name, shape, std, mean))

    def run_jobs():
        # every tensor has its own generator, so the draws are independent of the order and of the thread that
        # makes them; torch.randn releases the GIL
        def one(job):
            name, shape, std, mean = job
            return name, _randn(seed, name, shape, std, mean).to(device)
        if threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(threads) as ex:
                for name, t in ex.map(one, jobs):
                    sd[name] = t
        else:
            for job in jobs:
                sd[job[0]] = one(job)[1]

    put("language_model.model.embed_tokens.weight", (dims.vocab_size, H), 0.02)
    for i in range(dims.num_layers):
        for name, kind in llm_layer_names(i):
            if kind in ("nd", "nh"):
                put(name, shapes[kind], 0.1, 1.0)
            else:
                put(name, shapes[kind], 0.02)
    put("language_model.model.norm.weight", (H,), 0.1, 1.0)
    put("language_model.model.norm_moe_gen.weight", (H,), 0.1, 1.0)
    if include_lm_head:
        put("language_model.lm_head.weight", (dims.vocab_size, H), 0.02)
    F = dims.timestep_freq_dim
    put("time_embedder.mlp.0.weight", (H, F), 0.02)
    put("time_embedder.mlp.0.bias", (H,), 0.02)
    put("time_embedder.mlp.2.weight", (H, H), 0.02)
    put("time_embedder.mlp.2.bias", (H,), 0.02)
    P = dims.patch_latent_dim
    put("vae2llm.weight", (H, P), 0.02)
    put("vae2llm.bias", (H,), 0.02)
    put("llm2vae.weight", (P, H), 0.8 / H ** 0.5)
    put("llm2vae.bias", (P,), 0.02)
    run_jobs()
    pe = get_2d_sincos_pos_embed(H, dims.max_latent_size)
    sd["latent_pos_embed.pos_embed"] = torch.from_numpy(pe).float().to(torch.bfloat16).to(device)
    return sd


def make_vit_state_dict(dims: BagelDims, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """bf16 weights of the SigLIP tower (patch embedding already in its linear form,
    siglip_navit.py:167-182), the connector and ``vit_pos_embed`` (bagel.py:82-87).
    Matrices ~ N(0, 1/fan_in) so activations keep unit scale; LayerNorm affine ~ N(1, 0.1) / N(0, 0.1)."""
    vit: ViTDims = dims.vit
    Hv, Iv, H = vit.hidden_size, vit.intermediate_size, dims.hidden_size
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, nout, nin, gain=1.0):
        sd[name + ".weight"] = _randn(seed, name + ".weight", (nout, nin), gain * nin ** -0.5).to(device)
        sd[name + ".bias"] = _randn(seed, name + ".bias", (nout,), 0.1).to(device)

    def ln(name, n):
        sd[name + ".weight"] = _randn(seed, name + ".weight", (n,), 0.1, 1.0).to(device)
        sd[name + ".bias"] = _randn(seed, name + ".bias", (n,), 0.1).to(device)

    pre = "vit_model.vision_model."
    lin(pre + "embeddings.patch_embedding", Hv, vit.patch_dim)
    sd[pre + "embeddings.position_embedding.weight"] = _randn(
        seed, pre + "embeddings.position_embedding.weight", (vit.max_patches_per_side ** 2, Hv), 0.5).to(device)
    for i in range(vit.num_layers):
        l = pre + f"encoder.layers.{i}."
        ln(l + "layer_norm1", Hv)
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(l + "self_attn." + n, Hv, Hv)
        ln(l + "layer_norm2", Hv)
        lin(l + "mlp.fc1", Iv, Hv)
        lin(l + "mlp.fc2", Hv, Iv)
    ln(pre + "post_layernorm", Hv)
    lin("connector.fc1", H, Hv)
    lin("connector.fc2", H, H, gain=0.5)
    pe = get_2d_sincos_pos_embed(H, vit.max_patches_per_side)
    sd["vit_pos_embed.pos_embed"] = torch.from_numpy(pe).float().to(torch.bfloat16).to(device)
    return sd


def vae_encoder_param_shapes(v: VAEDims) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of ``AutoEncoder.encoder`` (autoencoder.py:122-170)."""
    shapes: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def res(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    nres = len(v.ch_mult)
    conv("encoder.conv_in", v.in_channels, v.ch, 3)
    in_mult = (1,) + tuple(v.ch_mult)
    block_in = v.ch
    for lvl in range(nres):
        block_in = v.ch * in_mult[lvl]
        block_out = v.ch * v.ch_mult[lvl]
        for b in range(v.num_res_blocks):
            res(f"encoder.down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != nres - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", block_in, block_in, 3)
    res("encoder.mid.block_1", block_in, block_in)
    norm("encoder.mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"encoder.mid.attn_1.{n}", block_in, block_in, 1)
    res("encoder.mid.block_2", block_in, block_in)
    norm("encoder.norm_out", block_in)
    conv("encoder.conv_out", block_in, 2 * v.z_channels, 3)
    return shapes


def _make_vae_state_dict(shapes, seed, device):
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = _randn(seed, name, shape, fan_in ** -0.5, dtype=torch.float32).to(device)
        elif ".norm" in name and name.endswith(".weight"):
            sd[name] = _randn(seed, name, shape, 0.1, 1.0, dtype=torch.float32).to(device)
        else:
            sd[name] = _randn(seed, name, shape, 0.1, dtype=torch.float32).to(device)
    return sd


def make_vae_encoder_state_dict(v: VAEDims, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """fp32 encoder weights, drawn like the decoder's."""
    return _make_vae_state_dict(vae_encoder_param_shapes(v), seed, device)


def vae_decoder_param_shapes(v: VAEDims) -> Dict[str, Tuple[int, ...]]:
    """Parameter names/shapes of ``AutoEncoder.decoder`` (autoencoder.py:196-248)."""
    shapes: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cin, cout, k):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)

    def norm(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)

    def res(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    nres = len(v.ch_mult)
    block_in = v.ch * v.ch_mult[nres - 1]
    conv("decoder.conv_in", v.z_channels, block_in, 3)
    res("decoder.mid.block_1", block_in, block_in)
    norm("decoder.mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv(f"decoder.mid.attn_1.{n}", block_in, block_in, 1)
    res("decoder.mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = v.ch * v.ch_mult[lvl]
        for b in range(v.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", block_in, block_in, 3)
    norm("decoder.norm_out", block_in)
    conv("decoder.conv_out", block_in, v.out_ch, 3)
    return shapes


def make_vae_decoder_state_dict(v: VAEDims, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """fp32 decoder weights (``load_ae`` keeps the VAE in fp32, autoencoder.py:354).

    Convolutions ~ N(0, 1/sqrt(fan_in)) so activations keep unit scale through
    the 30-odd layers; norm affine ~ N(1, 0.1) / N(0, 0.1).
    """
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in vae_decoder_param_shapes(v).items():
        if len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = _randn(seed, name, shape, fan_in ** -0.5, dtype=torch.float32).to(device)
        elif ".norm" in name and name.endswith(".weight"):
            sd[name] = _randn(seed, name, shape, 0.1, 1.0, dtype=torch.float32).to(device)
        else:
            sd[name] = _randn(seed, name, shape, 0.1, dtype=torch.float32).to(device)
    return sd
