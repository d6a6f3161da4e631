"""Seeded parity scenarios shared by the golden-vector generator and the tests.

A scenario fixes the model shape, the weight seed, the three CFG contexts (as
token-id prompts that are prefilled through the understanding expert), the image
sizes and the sampler hyper-parameters.  Everything else (weights, noise, index
tensors) is regenerated from these numbers on whichever machine runs the test.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import dataclasses

from bagel_amd.config import BagelDims, ViTDims, tiny_dims, width7b_dims

NEW_TOKEN_IDS = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)


def _prompt(n: int, salt: int, vocab: int) -> List[int]:
    return [5 + (salt * 7919 + i * 104729) % (vocab - 5) for i in range(n)]


@dataclass
class Scenario:
    name: str
    dims: BagelDims
    images: List[Tuple[int, int]]
    # per context: number of prompt tokens per sample (None = empty NaiveCache)
    cond_prompt: Optional[List[int]]
    text_prompt: Optional[List[int]]
    img_prompt: Optional[List[int]]
    num_timesteps: int = 6
    timestep_shift: float = 3.0
    cfg_text_scale: float = 4.0
    cfg_img_scale: float = 1.5
    cfg_interval: Tuple[float, float] = (0.4, 1.0)
    cfg_renorm_min: float = 0.0
    cfg_renorm_type: str = "global"
    weight_seed: int = 0
    noise_seed: int = 42
    enable_taylorseer: bool = False      # TaylorSeer step skipping (modeling/cache_utils/taylorseer.py)
    max_steps: Optional[int] = None      # record/run only the first steps (big shapes)
    slow: bool = False                   # too heavy for the CPU checker inside the default test run

    def prompts(self, which: str) -> Optional[List[List[int]]]:
        lens = getattr(self, which + "_prompt")
        if lens is None:
            return None
        salt = {"cond": 1, "text": 2, "img": 3}[which]
        return [_prompt(n, salt * 31 + b, self.dims.vocab_size) for b, n in enumerate(lens)]


def _small_gqa() -> BagelDims:
    # 14 query heads over 2 kv heads (the 7:1 grouping of the 7B model) at head_dim 128
    return BagelDims(hidden_size=1792, intermediate_size=2304, num_layers=2, num_heads=14,
                     num_kv_heads=2, vocab_size=512)


SCENARIOS = {s.name: s for s in [
    # T2I through the inferencer: text-uncond context is empty, img-uncond == cond (inferencer.py:283-285)
    Scenario("tiny_t2i", tiny_dims(2), [(128, 128)], [5], None, [5]),
    # three distinct context lengths, none a multiple of the attention tile; channel renorm with a floor
    Scenario("tiny_edit3", tiny_dims(3), [(256, 192)], [301], [157], [9],
             cfg_interval=(0.0, 1.0), cfg_renorm_type="text_channel", cfg_img_scale=2.0, cfg_renorm_min=0.3),
    # packed batch of two images of different size; global renorm couples them (bagel.py:1485-1488)
    Scenario("tiny_batch2", tiny_dims(2), [(128, 256), (192, 128)], [12, 40], None, [12, 40]),
    Scenario("tiny_batch2_channel", tiny_dims(2), [(128, 256), (192, 128)], [12, 40], [3, 70], [12, 40],
             cfg_renorm_type="channel"),
    # text CFG only (eval/gen/gen_images_mp.py:61-86)
    Scenario("tiny_textcfg", tiny_dims(2), [(128, 128), (128, 128)], [7, 7], None, None,
             cfg_img_scale=1.0, cfg_interval=(0.0, 1.0)),
    Scenario("gqa7_t2i", _small_gqa(), [(256, 256)], [33], None, [33], num_timesteps=4),
    # BAGEL-7B width, 2 layers, 1024x1024, first two steps
    Scenario("w7b_1024", width7b_dims(2), [(1024, 1024)], [32], None, [32], num_timesteps=50,
             max_steps=2, slow=True),
    # the benchmark's model: BAGEL-7B width AND depth (28 layers, 14.6 B parameters in the two experts), 1024x1024,
    # first step with all three CFG branches
    Scenario("w7b_l28_1024", width7b_dims(28), [(1024, 1024)], [32], None, [32], num_timesteps=50,
             max_steps=1, slow=True),
    # plumbing configuration: Qwen2-0.5B-shaped MoT (head_dim 64), 256x256, 10 timesteps
    Scenario("c1_0p5b_256", BagelDims(hidden_size=896, intermediate_size=4864, num_layers=24, num_heads=14,
                                       num_kv_heads=2, vocab_size=2048),
             [(256, 256)], [16], None, [16], num_timesteps=10, slow=True),
]}


# Scenarios without a golden file (compared between configurations of the HIP path itself)
EXTRA_SCENARIOS = {s.name: s for s in [
    # 12 samples x 3 branches = 36 units in one packed forward: 72 text rows, more than one 64-row launch of the
    # understanding expert's weight-streaming kernel; two ranks hold 18 units (36 text rows) each
    Scenario("tiny_batch12", tiny_dims(2), [(64, 64), (96, 64), (64, 128)] * 4, [5, 9, 3] * 4, [2, 4, 7] * 4, [5, 9, 3] * 4,
             num_timesteps=4, cfg_interval=(0.0, 1.0)),
]}


# TaylorSeer step skipping (kept apart: its steps are not independent, so it is checked through generate_image)
TAYLOR_SCENARIOS = {s.name: s for s in [
    # steps 0-4 full, then one full step in three; the CFG interval ends at step 11
    Scenario("tiny_taylor", tiny_dims(2), [(128, 192)], [9], None, [9], num_timesteps=14, enable_taylorseer=True),
    # the CFG interval opens at step 4 and closes after step 11: the text-free and image-free branches count their own
    # steps (bagel.py:1273-1277), so in steps 5, 6, 8 they run the layers while the conditional branch is extrapolated,
    # in step 10 it is the other way round
    Scenario("tiny_taylor_late", tiny_dims(2), [(128, 192)], [9], [4], [9], num_timesteps=14, cfg_interval=(0.3, 0.9),
             enable_taylorseer=True),
]}


@dataclass
class TextCase:
    """Greedy autoregressive decoding after a prefilled prompt (Bagel.generate_text, bagel.py:1526-1596)."""
    name: str
    dims: BagelDims
    prompt: List[int]              # prompt tokens per sample
    max_length: int
    weight_seed: int = 0

    def prompts(self) -> List[List[int]]:
        return [_prompt(n, 97 + b, self.dims.vocab_size) for b, n in enumerate(self.prompt)]


TEXT_CASES = {c.name: c for c in [
    TextCase("text_tiny_b1", tiny_dims(2), [7], 12),
    TextCase("text_tiny_b2", tiny_dims(3), [12, 40], 10, weight_seed=3),
    TextCase("text_gqa7", _small_gqa(), [33], 8, weight_seed=1),
    # BAGEL-7B width (2 layers): the decode step's one-launch attention and GEMV / weight-streaming projections at their
    # real reduction lengths (3584, 18944), two samples with prompts on both sides of a 64-slot boundary
    TextCase("text_w7b", width7b_dims(2), [34, 70], 10, weight_seed=2),
]}


@dataclass
class ImageCtxCase:
    """Context building with image inputs (bagel.py:876-1143) for a packed batch: optional text prompt, then
    optional VAE-encoded image, then optional ViT-encoded image, per sample in that order; ``edit`` additionally
    appends a second prompt and runs image generation with the three CFG contexts of the editing path
    (inferencer.py:281-292)."""
    name: str
    dims: BagelDims
    first_prompt: Optional[List[int]]
    vae_images: Optional[List[Tuple[int, int]]]
    vit_images: Optional[List[Tuple[int, int]]]
    edit_prompt: Optional[List[int]] = None
    out_image: Tuple[int, int] = (128, 128)
    num_timesteps: int = 4
    weight_seed: int = 0
    image_seed: int = 11
    noise_seed: int = 5

    @property
    def batch(self) -> int:
        for v in (self.first_prompt, self.vae_images, self.vit_images):
            if v is not None:
                return len(v)
        raise ValueError("empty case")

    def prompts(self, which: str) -> List[List[int]]:
        lens = self.first_prompt if which == "first" else self.edit_prompt
        salt = 201 if which == "first" else 307
        return [_prompt(n, salt + b, self.dims.vocab_size) for b, n in enumerate(lens)]

    def images(self, which: str):
        """Seeded stand-ins for transformed images: [3, H, W] in [-1, 1]."""
        import torch
        sizes = self.vae_images if which == "vae" else self.vit_images
        g = torch.Generator().manual_seed(self.image_seed + (0 if which == "vae" else 1000))
        return [torch.rand((3, h, w), generator=g) * 2 - 1 for h, w in sizes]


IMAGE_CTX_CASES = {c.name: c for c in [
    ImageCtxCase("ctx_vit_tiny", tiny_dims(2, vit=True), [5, 9], None, [(56, 84), (112, 70)]),
    ImageCtxCase("ctx_vae_tiny", tiny_dims(2, vit=True), None, [(64, 96), (128, 64)], None),
    ImageCtxCase("edit_tiny", tiny_dims(3, vit=True), None, [(128, 96)], [(112, 84)], edit_prompt=[11],
                 out_image=(128, 96)),
    # the SigLIP so400m tower at its true width (1152 hidden, 16 heads of 72, MLP 4304, 70 x 70 position grid) and
    # reduced depth, two images of different size packed (20 x 28 and 31 x 12 patches), tiny language model behind it
    ImageCtxCase("ctx_vit_so400m", dataclasses.replace(tiny_dims(2), vit=ViTDims(num_layers=3)), [4, 6], None,
                 [(280, 392), (434, 168)]),
    # the editing path at BAGEL-7B width (2 layers) behind the so400m-wide tower (2 layers) and the full FLUX encoder:
    # a 194-token VAE context, a 142-token ViT context, a 22-token instruction, three CFG branches with three
    # different contexts (358 / 336 / 22 tokens) on a 256 x 192 output (sizes kept small for the fixture file)
    # mixed-resolution context building at BAGEL-7B width: two samples, each a prompt, a VAE-encoded image and a
    # ViT-encoded image, all of different sizes, packed into the varlen prefills
    ImageCtxCase("ctx_mixed_w7b", dataclasses.replace(width7b_dims(2), vit=ViTDims(num_layers=2)), [6, 11],
                 [(128, 192), (192, 128)], [(112, 168), (196, 140)]),
    ImageCtxCase("edit_w7b", dataclasses.replace(width7b_dims(2), vit=ViTDims(num_layers=2)), None, [(192, 256)],
                 [(140, 196)], edit_prompt=[20], out_image=(256, 192), num_timesteps=3),
]}
