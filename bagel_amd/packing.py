"""Host-side index work of the packed varlen API: everything the reference's ``prepare_*`` methods return.

The packed API describes a batch by flat index tensors.  All of them derive from ONE layout rule
(qwen2_navit.py:574-581): the merged key/value sequence of a forward is, sample after sample,
``[context slots of the sample | slots of the sample's new tokens]``.  ``merged_layout`` states that rule once;
each builder below then only says what a sample's new tokens are:

    prepare_prompts          bos + prompt + eos, consecutive RoPE positions        (bagel.py:806-841)
    prepare_image_tokens     start_of_image + n image tokens + end_of_image at ONE   (bagel.py:883-920, 1017-1058)
                             RoPE position; the context's position advances by 1
    prepare_vae_latent       the same token pattern for the image being generated,   (bagel.py:1145-1201)
                             plus the CPU ``randn`` initial noise, one draw per image
    prepare_vae_latent_cfg   that pattern seen from another CFG context              (bagel.py:1203-1234)
    prepare_start_tokens     no new slots: one bos query per sample                  (bagel.py:1502-1523)

The outputs are compared bit for bit (``torch.equal``, dtype included) with tensors captured from the reference
(tests/test_index_fixtures.py), so dtypes, ordering and the order of RNG draws are part of the contract.
Nothing here touches a GPU.
"""
from __future__ import annotations

from itertools import chain
from typing import Callable, Dict, List, Sequence, Tuple

import torch


def merged_layout(kv_lens: Sequence[int], new_lens: Sequence[int]) -> Tuple[List[range], List[range]]:
    """Slots of the merged key/value sequence: per sample (context slots, new-token slots)."""
    ctx, new = [], []
    cur = 0
    for kv, n in zip(kv_lens, new_lens):
        ctx.append(range(cur, cur + kv))
        new.append(range(cur + kv, cur + kv + n))
        cur += kv + n
    return ctx, new


def _flat(ranges) -> List[int]:
    return list(chain.from_iterable(ranges))


def _long(values, device=None):
    return torch.tensor(list(values), dtype=torch.long, device=device)


def _int(values, device=None):
    return torch.tensor(list(values), dtype=torch.int, device=device)


def flattened_position_ids_extrapolate(img_h: int, img_w: int, patch_size: int, max_num_patches_per_side: int):
    """Row-major patch coordinates on a ``max_num_patches_per_side`` wide grid (data/data_utils.py:53-58)."""
    rows = torch.arange(0, img_h // patch_size)
    cols = torch.arange(0, img_w // patch_size)
    return (rows[:, None] * max_num_patches_per_side + cols).flatten()


def patchify(image: torch.Tensor, patch: int) -> torch.Tensor:
    """[C, H, W] -> [(H/p)(W/p), p*p*C] with (py, px, c) minor order (data/data_utils.py:43-50)."""
    c, h, w = image.shape
    assert h % patch == 0 and w % patch == 0
    return image.reshape(c, h // patch, patch, w // patch, patch).permute(1, 3, 2, 4, 0).reshape(-1, patch * patch * c)


# ---------------------------------------------------------------------------------------------------------
def prepare_prompts(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, device=None):
    token_rows = [[new_token_ids["bos_token_id"], *tokenizer.encode(p), new_token_ids["eos_token_id"]] for p in prompts]
    lens = [len(r) for r in token_rows]
    ctx, new = merged_layout(curr_kvlens, lens)
    generation_input = {
        "text_token_lens": _int(lens, device),
        "packed_text_ids": _long(_flat(token_rows), device),
        "packed_text_position_ids": _long(_flat(range(r, r + n) for r, n in zip(curr_rope, lens)), device),
        "packed_text_indexes": _long(_flat(new), device),
        "packed_key_value_indexes": _long(_flat(ctx), device),
        "key_values_lens": _int(curr_kvlens, device),
    }
    newlens = [kv + n for kv, n in zip(curr_kvlens, lens)]
    new_rope = [r + n for r, n in zip(curr_rope, lens)]
    return generation_input, newlens, new_rope


def _image_pattern(token_counts: Sequence[int]):
    """Query-side rows of ``[start, n tokens, end]`` per sample: (rows of the two special tokens, image-token rows)."""
    special, image = [], []
    q = 0
    for n in token_counts:
        special += [q, q + n + 1]
        image.append(range(q + 1, q + n + 1))
        q += n + 2
    return special, _flat(image)


def prepare_image_tokens(curr_kvlens, curr_rope, token_counts, new_token_ids):
    """Index tensors shared by the ViT-token and VAE-token context updates; returns
    (generation_input, image-token rows, new kv lens, new RoPE positions)."""
    seqlens = [n + 2 for n in token_counts]
    ctx, new = merged_layout(curr_kvlens, seqlens)
    special, image_rows = _image_pattern(token_counts)
    generation_input = {
        "packed_text_ids": _long([new_token_ids["start_of_image"], new_token_ids["end_of_image"]] * len(seqlens)),
        "packed_text_indexes": _long(special),
        "packed_position_ids": _long(_flat([r] * n for r, n in zip(curr_rope, seqlens))),
        "packed_seqlens": _int(seqlens),
        "packed_indexes": _long(_flat(new)),
        "packed_key_value_indexes": _long(_flat(ctx)),
        "key_values_lens": _int(curr_kvlens),
    }
    newlens = [kv + n for kv, n in zip(curr_kvlens, seqlens)]
    new_rope = [r + 1 for r in curr_rope]
    return generation_input, image_rows, newlens, new_rope


def prepare_vit_images(curr_kvlens, curr_rope, images, transforms: Callable, new_token_ids, patch_size: int,
                       max_patches_per_side: int):
    """bagel.py:876-936."""
    tensors = [transforms(im) for im in images]
    tokens = [patchify(t, patch_size) for t in tensors]
    counts = [t.shape[0] for t in tokens]
    gi, image_rows, newlens, new_rope = prepare_image_tokens(curr_kvlens, curr_rope, counts, new_token_ids)
    gi.update({
        "vit_token_seqlens": _int(counts),
        "packed_vit_tokens": torch.cat(tokens, dim=0),
        "packed_vit_position_ids": torch.cat([flattened_position_ids_extrapolate(t.size(1), t.size(2), patch_size,
                                                                                 max_patches_per_side) for t in tensors]),
        "packed_vit_token_indexes": _long(image_rows),
    })
    return gi, newlens, new_rope


def prepare_vae_images(curr_kvlens, curr_rope, images, transforms: Callable, new_token_ids, latent_downsample: int,
                       max_latent_size: int, timestep=0):
    """bagel.py:1010-1081."""
    tensors = [transforms(im) for im in images]
    shapes = [(t.shape[1] // latent_downsample, t.shape[2] // latent_downsample) for t in tensors]
    gi, image_rows, newlens, new_rope = prepare_image_tokens(curr_kvlens, curr_rope, [h * w for h, w in shapes],
                                                             new_token_ids)
    # images of different size share one zero-padded batch tensor, top-left aligned
    extent = [max(dim) for dim in zip(*[t.shape for t in tensors])]
    padded = torch.zeros(size=(len(tensors), *extent))
    for slot, t in zip(padded, tensors):
        slot[:, :t.shape[1], :t.shape[2]] = t
    gi.update({
        "padded_images": padded,
        "patchified_vae_latent_shapes": shapes,
        "packed_vae_position_ids": torch.cat([flattened_position_ids_extrapolate(t.size(1), t.size(2), latent_downsample,
                                                                                 max_latent_size) for t in tensors]),
        "packed_timesteps": torch.tensor([timestep]),
        "packed_vae_token_indexes": _long(image_rows),
    })
    return gi, newlens, new_rope


def prepare_start_tokens(curr_kvlens, curr_rope, new_token_ids, device=None) -> Dict[str, torch.Tensor]:
    ctx, _ = merged_layout(curr_kvlens, [0] * len(curr_kvlens))
    return {
        "packed_start_tokens": _long([new_token_ids["bos_token_id"]] * len(curr_kvlens), device),
        "packed_query_position_ids": _long(curr_rope, device),
        "key_values_lens": _int(curr_kvlens, device),
        "packed_key_value_indexes": _long(_flat(ctx), device),
    }


def prepare_vae_latent(curr_kvlens, curr_rope, image_sizes, new_token_ids, latent_downsample: int, max_latent_size: int,
                       patch_latent_dim: int) -> Dict[str, torch.Tensor]:
    """The initial noise is drawn with the global CPU generator, one ``randn`` per image in batch order: that
    order is what makes a seeded run reproduce the reference's latents (bagel.py:1171)."""
    counts = [(H // latent_downsample) * (W // latent_downsample) for H, W in image_sizes]
    seqlens = [n + 2 for n in counts]
    ctx, new = merged_layout(curr_kvlens, seqlens)
    special, image_rows = _image_pattern(counts)
    noises = [torch.randn(n, patch_latent_dim) for n in counts]
    return {
        "packed_text_ids": _long([new_token_ids["start_of_image"], new_token_ids["end_of_image"]] * len(counts)),
        "packed_text_indexes": _long(special),
        "packed_init_noises": torch.cat(noises, dim=0),
        "packed_vae_position_ids": torch.cat([flattened_position_ids_extrapolate(H, W, latent_downsample, max_latent_size)
                                              for H, W in image_sizes], dim=0),
        "packed_vae_token_indexes": _long(image_rows),
        "packed_seqlens": _int(seqlens),
        "packed_position_ids": _long(_flat([r] * n for r, n in zip(curr_rope, seqlens))),
        "key_values_lens": _int(curr_kvlens),
        "packed_indexes": _long(_flat(new)),
        "packed_key_value_indexes": _long(_flat(ctx)),
    }


def prepare_vae_latent_cfg(curr_kvlens, curr_rope, image_sizes, latent_downsample: int) -> Dict[str, torch.Tensor]:
    seqlens = [(H // latent_downsample) * (W // latent_downsample) + 2 for H, W in image_sizes]
    ctx, new = merged_layout(curr_kvlens, seqlens)
    return {
        "cfg_packed_position_ids": _long(_flat([r] * n for r, n in zip(curr_rope, seqlens))),
        "cfg_key_values_lens": _int(curr_kvlens),
        "cfg_packed_query_indexes": _long(_flat(new)),
        "cfg_packed_key_value_indexes": _long(_flat(ctx)),
    }
