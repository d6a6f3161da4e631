"""Host logic of the checkpoint loader and the image transforms (no GPU)."""
import json

import numpy as np
import torch
from PIL import Image
from safetensors.torch import save_file

from bagel_amd.config import BAGEL_7B
from bagel_amd.loader import LazySafetensors, add_special_tokens, dims_from_config_files
from bagel_amd.transforms import ImageTransform, MaxLongEdgeMinShortEdgeResize, pil_img2rgb


class FakeTokenizer:
    def __init__(self):
        self.vocab = {"<|im_start|>": 7}
        self.special_tokens_map = {"eos_token": "<|endoftext|>", "additional_special_tokens": ["<|im_start|>"]}

    def add_tokens(self, toks):
        for t in toks:
            self.vocab[t] = 100 + len(self.vocab)
        return len(toks)

    def convert_tokens_to_ids(self, t):
        return self.vocab[t]


def test_dims_from_config_files_reproduces_7b(tmp_path):
    # the public BAGEL-7B-MoT numbers (Qwen2.5-7B backbone, SigLIP so400m-patch14 with 27 layers on disk)
    (tmp_path / "llm_config.json").write_text(json.dumps(dict(
        hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
        vocab_size=152064, rope_theta=1000000.0, rms_norm_eps=1e-6)))
    (tmp_path / "vit_config.json").write_text(json.dumps(dict(
        hidden_size=1152, intermediate_size=4304, num_hidden_layers=27, num_attention_heads=16, patch_size=14,
        image_size=980, num_channels=3)))
    assert dims_from_config_files(str(tmp_path)) == BAGEL_7B


def test_add_special_tokens_keeps_existing_ids():
    tok, ids, n_new = add_special_tokens(FakeTokenizer())
    assert n_new == 3 and ids["bos_token_id"] == 7
    assert set(ids) == {"bos_token_id", "eos_token_id", "start_of_image", "end_of_image"}
    assert len({*ids.values()}) == 4


def test_lazy_safetensors(tmp_path):
    p = str(tmp_path / "w.safetensors")
    save_file({"a.weight": torch.arange(6.).view(2, 3), "b": torch.ones(2)}, p)
    sd = LazySafetensors(p)
    assert "a.weight" in sd and "c" not in sd and sorted(sd.keys()) == ["a.weight", "b"]
    assert torch.equal(sd["a.weight"], torch.arange(6.).view(2, 3))
    assert dict(sd.items())["b"].sum() == 2


def test_resize_rule_matches_reference_cases():
    # worked by hand from data/transforms.py:49-85
    r = MaxLongEdgeMinShortEdgeResize(max_size=1024, min_size=512, stride=16, max_pixels=14 * 14 * 9 * 1024)
    assert r.target_size(2048, 1024) == (1024, 512)       # long edge capped
    assert r.target_size(300, 200) == (768, 512)          # short edge raised to 512
    assert r.target_size(1000, 1000) == (992, 992)        # 62.5 strides round half to even, as Python's round does
    v = MaxLongEdgeMinShortEdgeResize(max_size=980, min_size=224, stride=14, max_pixels=14 * 14 * 9 * 1024)
    assert v.target_size(4000, 3000) == (980, 728)        # 735 / 14 = 52.5 patches rounds to 52
    assert v.target_size(100, 100) == (224, 224)
    assert all(s % 14 == 0 for s in v.target_size(1234, 777))


def test_image_transform_output():
    img = Image.fromarray((np.arange(64 * 48 * 3) % 256).astype(np.uint8).reshape(48, 64, 3))
    t = ImageTransform(128, 32, 16)(img)
    assert t.shape == (3, 48, 64) and t.dtype == torch.float32
    assert torch.allclose(t, torch.from_numpy(np.array(img)).permute(2, 0, 1).float() / 255 * 2 - 1)
    rgba = Image.new("RGBA", (4, 4), (255, 0, 0, 0))
    assert pil_img2rgb(rgba).getpixel((0, 0)) == (255, 255, 255)   # transparent -> white
