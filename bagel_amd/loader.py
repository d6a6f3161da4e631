"""Checkpoint directory -> HIP-backed model objects.

``load_bagel_model_for_inference`` mirrors the reference's loader of the same name
(unified_image_editing_inference.py:867-1016; the same steps as app.py:39-113): read ``llm_config.json`` and
``vit_config.json``, build the autoencoder from ``ae.safetensors``, the model from ``ema.safetensors``, the
tokenizer with BAGEL's four special tokens and the two image transforms, and return the same 6-tuple.
Tensors are read one at a time from the safetensors files (nothing is unpickled, nothing is materialised
twice on the host).  The quantised modes (bitsandbytes NF4 / INT8) are not offered.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Tuple, Union

from PIL import Image
from safetensors import safe_open

from .config import BagelDims, VAEDims, ViTDims
from .transforms import ImageTransform


class LazySafetensors:
    """Read-only mapping over a .safetensors file that loads a tensor when it is asked for."""

    def __init__(self, path: str):
        self.path = path
        self._f = safe_open(path, framework="pt", device="cpu")
        self._keys = set(self._f.keys())

    def __contains__(self, name):
        return name in self._keys

    def __getitem__(self, name):
        if name not in self._keys:
            raise KeyError(f"{name} not found in {self.path}")
        return self._f.get_tensor(name)

    def __iter__(self):
        return iter(self._keys)

    def keys(self):
        return self._keys

    def items(self):
        for k in self._keys:
            yield k, self._f.get_tensor(k)


def dims_from_config_files(model_path: str) -> BagelDims:
    """``llm_config.json`` (Qwen2 keys) and ``vit_config.json`` (SigLIP keys) -> BagelDims, with the fixed choices
    of the reference's loaders: the tower's last layer dropped, 70 patches per side, latent patch 2, 64 latent
    positions per side (unified_image_editing_inference.py:888-912; FLUX VAE dims from autoencoder.py:340-351)."""
    with open(os.path.join(model_path, "llm_config.json")) as f:
        c = json.load(f)
    vit = None
    vit_path = os.path.join(model_path, "vit_config.json")
    if os.path.exists(vit_path):
        with open(vit_path) as f:
            v = json.load(f)
        vit = ViTDims(hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"],
                      num_layers=v["num_hidden_layers"] - 1, num_heads=v["num_attention_heads"],
                      patch_size=v.get("patch_size", 14), num_channels=v.get("num_channels", 3),
                      max_patches_per_side=70, layer_norm_eps=v.get("layer_norm_eps", 1e-6))
    return BagelDims(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                     num_layers=c["num_hidden_layers"], num_heads=c["num_attention_heads"],
                     num_kv_heads=c.get("num_key_value_heads", c["num_attention_heads"]), vocab_size=c["vocab_size"],
                     rope_theta=float(c.get("rope_theta", 10000.0)), rms_norm_eps=float(c.get("rms_norm_eps", 1e-6)),
                     latent_patch_size=2, max_latent_size=64, vae=VAEDims(), vit=vit)


def add_special_tokens(tokenizer):
    """data/data_utils.py:130-163."""
    all_special_tokens = []
    for v in tokenizer.special_tokens_map.values():
        if isinstance(v, str):
            all_special_tokens.append(v)
        elif isinstance(v, list):
            all_special_tokens += v
    new_tokens = [t for t in ("<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>")
                  if t not in all_special_tokens]
    num_new_tokens = tokenizer.add_tokens(new_tokens)
    new_token_ids = dict(bos_token_id=tokenizer.convert_tokens_to_ids("<|im_start|>"),
                         eos_token_id=tokenizer.convert_tokens_to_ids("<|im_end|>"),
                         start_of_image=tokenizer.convert_tokens_to_ids("<|vision_start|>"),
                         end_of_image=tokenizer.convert_tokens_to_ids("<|vision_end|>"))
    return tokenizer, new_token_ids, num_new_tokens


def load_bagel_model_for_inference(model_path: str, mode: int = 1, device: str = "cuda", tokenizer=None):
    """Returns ``(model, vae_model, tokenizer, vae_transform, vit_transform, new_token_ids)``.
    ``tokenizer``: an already built tokenizer (otherwise ``Qwen2Tokenizer.from_pretrained(model_path)``)."""
    if mode != 1:
        raise NotImplementedError(f"load mode {mode}: the bitsandbytes NF4 / INT8 modes are not offered (bf16 only)")
    from .modeling import AutoEncoder, Bagel
    if device == "cuda":
        device = "cuda:0"
    dims = dims_from_config_files(model_path)
    vae_model = AutoEncoder(dims.vae, LazySafetensors(os.path.join(model_path, "ae.safetensors")), device=device)
    model = Bagel(dims, state_dict=LazySafetensors(os.path.join(model_path, "ema.safetensors")), device=device)
    if tokenizer is None:
        from transformers import Qwen2Tokenizer
        tokenizer = Qwen2Tokenizer.from_pretrained(model_path)
    tokenizer, new_token_ids, _ = add_special_tokens(tokenizer)
    vae_transform = ImageTransform(1024, 512, 16)
    vit_transform = ImageTransform(980, 224, 14)
    return model, vae_model, tokenizer, vae_transform, vit_transform, new_token_ids


class UnifiedImageEditingInference:
    """The editing front end of unified_image_editing_inference.py:1019-1435 over ``InterleaveInferencer``:
    ``edit_image``, ``unified_edit_image`` (interleave path), ``batch_edit_images``, ``multi_step_edit``.
    The experimental token-by-token "unified autoregressive" generator (:47-865) is not offered."""

    def __init__(self, model_path: str, mode: int = 1, device: str = "cuda", tokenizer=None):
        from .inferencer import InterleaveInferencer
        self.device, self.model_path = device, model_path
        (self.model, self.vae_model, self.tokenizer, self.vae_transform, self.vit_transform,
         self.new_token_ids) = load_bagel_model_for_inference(model_path, mode=mode, device=device, tokenizer=tokenizer)
        self.inferencer = InterleaveInferencer(model=self.model, vae_model=self.vae_model, tokenizer=self.tokenizer,
                                               vae_transform=self.vae_transform, vit_transform=self.vit_transform,
                                               new_token_ids=self.new_token_ids)

    def edit_image(self, image_path: str, edit_prompt: str, think: bool = True, cfg_text_scale: float = 3.0,
                   cfg_img_scale: float = 1.5, num_timesteps: int = 50, image_shapes: Tuple[int, int] = (1024, 1024),
                   **kwargs) -> List[Union[str, Image.Image]]:
        """:1073-1135; errors are reported in the result list, as there."""
        try:
            input_image = Image.open(image_path).convert("RGB")
        except Exception as e:  # noqa: BLE001
            return [f"image load failed: {e}"]
        try:
            return self.inferencer.interleave_inference(
                input_lists=[input_image, edit_prompt], think=think, understanding_output=False,
                cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, num_timesteps=num_timesteps,
                image_shapes=image_shapes, **kwargs)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return [f"image edit failed: {e}"]

    def unified_edit_image(self, image_path: str, edit_prompt: str, use_autoregressive: bool = True, max_length: int = 500,
                           do_sample: bool = True, temperature: float = 0.8, cfg_text_scale: float = 3.0,
                           cfg_img_scale: float = 1.5, num_timesteps: int = 50,
                           image_shapes: Tuple[int, int] = (1024, 1024), **kwargs):
        """:1137-1221 with ``use_autoregressive=False`` (the interleave path)."""
        if use_autoregressive:
            raise NotImplementedError("the unified token-by-token generator is not offered; pass use_autoregressive=False")
        try:
            input_image = Image.open(image_path).convert("RGB")
        except Exception as e:  # noqa: BLE001
            return [f"image load failed: {e}"]
        try:
            return self.inferencer.interleave_inference(
                input_lists=[input_image, edit_prompt], understanding_output=False, cfg_text_scale=cfg_text_scale,
                cfg_img_scale=cfg_img_scale, num_timesteps=num_timesteps, image_shapes=image_shapes, **kwargs)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            return [f"unified image edit failed: {e}"]

    @staticmethod
    def _save(results, directory, stem):
        last_image = None
        for j, result in enumerate(results):
            if isinstance(result, Image.Image):
                last_image = os.path.join(directory, f"{stem}result_{j}.png")
                result.save(last_image)
            elif isinstance(result, str):
                with open(os.path.join(directory, f"{stem}text_{j}.txt"), "w", encoding="utf-8") as f:
                    f.write(result)
        return last_image

    def batch_edit_images(self, image_paths: List[str], edit_prompts: List[str], output_dir: str = "edited_images",
                          use_autoregressive: bool = False, **kwargs) -> Dict[str, list]:
        """:1303-1364."""
        os.makedirs(output_dir, exist_ok=True)
        results = {}
        for i, (image_path, edit_prompt) in enumerate(zip(image_paths, edit_prompts)):
            if use_autoregressive:
                r = self.unified_edit_image(image_path=image_path, edit_prompt=edit_prompt, use_autoregressive=True, **kwargs)
            else:
                r = self.edit_image(image_path=image_path, edit_prompt=edit_prompt, **kwargs)
            results[f"image_{i + 1}"] = r
            self._save(r, output_dir, f"image_{i + 1}_")
        return results

    def multi_step_edit(self, image_path: str, edit_steps: List[str], output_dir: str = "multi_step_edit",
                        use_autoregressive: bool = False, **kwargs) -> list:
        """:1366-1435: every step edits the previous step's image."""
        os.makedirs(output_dir, exist_ok=True)
        all_results = []
        current = image_path
        for i, edit_step in enumerate(edit_steps):
            if use_autoregressive:
                r = self.unified_edit_image(image_path=current, edit_prompt=edit_step, use_autoregressive=True, **kwargs)
            else:
                r = self.edit_image(image_path=current, edit_prompt=edit_step, **kwargs)
            all_results.extend(r)
            step_dir = os.path.join(output_dir, f"step_{i + 1}")
            os.makedirs(step_dir, exist_ok=True)
            saved = self._save(r, step_dir, "")
            if saved is not None:
                current = saved
        return all_results
