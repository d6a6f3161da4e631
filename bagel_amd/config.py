"""Static shape description of a BAGEL Mixture-of-Transformers model.

The reference reads these numbers from ``llm_config.json`` at load time
(/root/reference/app.py:39-60) and from ``load_ae``
(/root/reference/modeling/autoencoder.py:339-351); here they are a plain
dataclass so that the same description drives the weight generator, the HIP
host code and the CPU checker.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Optional, Tuple


@dataclass(frozen=True)
class VAEDims:
    """FLUX autoencoder dims (/root/reference/modeling/autoencoder.py:340-351)."""
    resolution: int = 256
    in_channels: int = 3
    downsample: int = 8
    ch: int = 128
    out_ch: int = 3
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 16
    scale_factor: float = 0.3611
    shift_factor: float = 0.1159


@dataclass(frozen=True)
class ViTDims:
    """SigLIP NaViT tower as BAGEL configures it (/root/reference/app.py:45-48: so400m-patch14 at 980 px with
    rope off and the last layer dropped; head_dim 72)."""
    hidden_size: int = 1152
    intermediate_size: int = 4304
    num_layers: int = 26
    num_heads: int = 16
    patch_size: int = 14
    num_channels: int = 3
    max_patches_per_side: int = 70
    layer_norm_eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def patch_dim(self) -> int:
        return self.num_channels * self.patch_size ** 2


@dataclass(frozen=True)
class BagelDims:
    """LLM + generation-head dims.

    ``head_dim`` is always ``hidden_size // num_heads``
    (/root/reference/modeling/qwen2/modeling_qwen2.py:236).
    """
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_layers: int = 28
    num_heads: int = 28
    num_kv_heads: int = 4
    vocab_size: int = 152064
    rope_theta: float = 1000000.0
    rms_norm_eps: float = 1e-6
    latent_patch_size: int = 2
    max_latent_size: int = 64
    timestep_freq_dim: int = 256
    vae: VAEDims = field(default_factory=VAEDims)
    vit: Optional[ViTDims] = None      # None: no understanding tower (image inputs unavailable)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads

    @property
    def latent_channel(self) -> int:
        return self.vae.z_channels

    @property
    def patch_latent_dim(self) -> int:
        # /root/reference/modeling/bagel/bagel.py:75
        return self.latent_patch_size ** 2 * self.vae.z_channels

    @property
    def latent_downsample(self) -> int:
        # /root/reference/modeling/bagel/bagel.py:72
        return self.vae.downsample * self.latent_patch_size

    @property
    def q_dim(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_kv_heads * self.head_dim

    def to_dict(self):
        return asdict(self)


# Public BAGEL-7B-MoT dims (Qwen2.5-7B backbone, SigLIP so400m tower).
BAGEL_7B = BagelDims(vit=ViTDims())

# Qwen2-0.5B-shaped MoT used by the CPU-runnable plumbing configuration.
BAGEL_0P5B = BagelDims(hidden_size=896, intermediate_size=4864, num_layers=24,
                       num_heads=14, num_kv_heads=2, vocab_size=151936)


def tiny_vit() -> ViTDims:
    """Eight heads of the real head_dim (72; hidden 576 = 9 x 64), an intermediate size that is not a
    multiple of 64 (like 4304)."""
    return ViTDims(hidden_size=576, intermediate_size=272, num_layers=2, num_heads=8, max_patches_per_side=8)


def tiny_dims(num_layers: int = 2, vocab_size: int = 512, max_latent_size: int = 64, vit: bool = False) -> BagelDims:
    """Smallest shape that still exercises GQA (2 q heads per kv head) at head_dim 128."""
    return BagelDims(hidden_size=256, intermediate_size=512, num_layers=num_layers,
                     num_heads=2, num_kv_heads=1, vocab_size=vocab_size,
                     max_latent_size=max_latent_size, vit=tiny_vit() if vit else None)


def width7b_dims(num_layers: int = 2, vocab_size: int = 512) -> BagelDims:
    """BAGEL-7B width at reduced depth (CPU-checkable)."""
    return BagelDims(num_layers=num_layers, vocab_size=vocab_size)
