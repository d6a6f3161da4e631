"""Checkpoint directory -> loader -> editing front end, end to end on the GPU with a synthetic tiny checkpoint
(no BAGEL checkpoint exists offline: configs and safetensors are written by the repo's weight generator)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image
from safetensors.torch import save_file

pytestmark = pytest.mark.gpu


class Tok:
    """Stand-in tokenizer with the surface the loader and the inferencer use."""

    def __init__(self):
        self.vocab = {}
        self.special_tokens_map = {}

    def add_tokens(self, toks):
        for t in toks:
            self.vocab[t] = 1 + len(self.vocab)
        return len(toks)

    def convert_tokens_to_ids(self, t):
        return self.vocab[t]

    def encode(self, text):
        return [5 + (ord(c) % 200) for c in text][:24]

    def decode(self, ids):
        return "<|im_start|>" + " ".join(str(int(i)) for i in ids[1:]) + "<|im_end|>"


@pytest.fixture(scope="module")
def checkpoint_dir(tmp_path_factory):
    from bagel_amd.config import BagelDims, VAEDims, ViTDims
    from bagel_amd.weights import (make_llm_state_dict, make_vit_state_dict, make_vae_encoder_state_dict,
                                   make_vae_decoder_state_dict)
    d = tmp_path_factory.mktemp("ckpt")
    vit = ViTDims(hidden_size=576, intermediate_size=272, num_layers=2, num_heads=8, max_patches_per_side=70)
    dims = BagelDims(hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, vocab_size=512, vit=vit)
    (d / "llm_config.json").write_text(json.dumps(dict(
        hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
        vocab_size=512, rope_theta=1000000.0, rms_norm_eps=1e-6)))
    (d / "vit_config.json").write_text(json.dumps(dict(
        hidden_size=576, intermediate_size=272, num_hidden_layers=3, num_attention_heads=8, patch_size=14, image_size=980)))
    sd = make_llm_state_dict(dims, seed=0, include_lm_head=True)
    sd.update(make_vit_state_dict(dims, seed=0))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "ema.safetensors"))
    vsd = make_vae_encoder_state_dict(VAEDims(), seed=0)
    vsd.update(make_vae_decoder_state_dict(VAEDims(), seed=0))
    save_file(vsd, str(d / "ae.safetensors"))
    rng = np.random.default_rng(0)
    Image.fromarray(rng.integers(0, 256, (60, 90, 3), dtype=np.uint8)).save(str(d / "input.png"))
    return d, dims


def test_loader_returns_reference_tuple(checkpoint_dir):
    from bagel_amd.loader import load_bagel_model_for_inference
    d, dims = checkpoint_dir
    model, vae, tok, vae_t, vit_t, ids = load_bagel_model_for_inference(str(d), device="cuda", tokenizer=Tok())
    assert model.dims == dims and model.vit_tower is not None and vae.has_encoder
    assert set(ids) == {"bos_token_id", "eos_token_id", "start_of_image", "end_of_image"}
    assert vae_t.stride == 16 and vit_t.stride == 14
    with pytest.raises(NotImplementedError):
        load_bagel_model_for_inference(str(d), mode=2, tokenizer=Tok())


def test_edit_image_end_to_end(checkpoint_dir, tmp_path):
    """Image + prompt -> edited image through VAE encode, SigLIP prefill, text prefill, 3-branch CFG denoise and
    VAE decode, all on HIP kernels; then the batch and multi-step front ends on top of it."""
    from bagel_amd.loader import UnifiedImageEditingInference
    d, _ = checkpoint_dir
    eng = UnifiedImageEditingInference(str(d), device="cuda", tokenizer=Tok())
    torch.manual_seed(0)
    out = eng.edit_image(str(d / "input.png"), "make it blue", think=False, num_timesteps=4)
    assert len(out) == 1 and isinstance(out[0], Image.Image), out
    # the output takes the resized input's shape: 90x60 -> short edge raised to 512, both sides multiples of 16
    assert out[0].size == (768, 512)
    a = np.asarray(out[0]).astype(np.float32)
    assert np.isfinite(a).all() and a.std() > 1.0
    # think mode adds the decoded plan before the image
    torch.manual_seed(0)
    out2 = eng.edit_image(str(d / "input.png"), "make it blue", think=True, num_timesteps=3, max_think_token_n=6)
    assert isinstance(out2[0], str) and isinstance(out2[1], Image.Image)
    # errors are reported in the result list, like the reference does
    assert "failed" in eng.edit_image(str(d / "missing.png"), "x")[0]
    res = eng.multi_step_edit(str(d / "input.png"), ["a", "b"], output_dir=str(tmp_path / "ms"), think=False, num_timesteps=3)
    assert len(res) == 2 and os.path.exists(tmp_path / "ms" / "step_2" / "result_0.png")
    with pytest.raises(NotImplementedError):
        eng.unified_edit_image(str(d / "input.png"), "x", use_autoregressive=True)
