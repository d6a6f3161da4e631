"""Orchestration of interleaved text / image requests over the HIP-backed ``Bagel`` and ``AutoEncoder``.

The public surface is the reference's ``InterleaveInferencer`` (inferencer.py:37-325, 982-1007): same constructor,
method names, keyword defaults and return values, so callers switch by changing an import.  Underneath, a request
is modelled by two small objects of this module:

``GenContext``
    one KV context: per-sample lengths, RoPE positions and the ``NaiveCache``.  It is a ``dict`` with the
    reference's three keys (callers index it, and ``deepcopy`` still works), plus ``fork()``.  The engine never
    writes into a cache tensor - every prefill returns new tensors - so a fork shares tensors with its parent and
    costs nothing, where the reference deep-copies up to 0.5 GB of keys and values per CFG branch
    (inferencer.py:269-292).

``CfgContexts``
    the three contexts classifier-free guidance needs, advanced together term by term: ``full`` (everything),
    ``no_text`` (the state before the most recent text, or right after the most recent image) and ``no_image``
    (text only).  The rules are the reference's (inferencer.py:273-292), stated once in ``add_text`` /
    ``add_image`` instead of inline in the request loop.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch
from PIL import Image

from .modeling.bagel import NaiveCache
from .transforms import pil_img2rgb

# System prompts of the think modes: model inputs, so they are the reference's strings character for character
# (inferencer.py:15-21).
# (written as explicit line pieces: the first line of each prompt ends in a space before its newline)
VLM_THINK_SYSTEM_PROMPT = (
    "You should first think about the reasoning process in the mind and then provide the user with the answer. \n"
    "The reasoning process is enclosed within <think> </think> tags, i.e. <think> reasoning process here </think> answer here")

GEN_THINK_SYSTEM_PROMPT = (
    "You should first think about the planning process in the mind and then generate the image. \n"
    "The planning process is enclosed within <think> </think> tags, i.e. <think> planning process here </think> "
    "<|vision_start|> image will be generated here <|vision_end|>\n"
    "\n"
    "When you want to generate an image, use the special tokens <|vision_start|> and <|vision_end|> to indicate where the "
    "image should be placed.")


class GenContext(dict):
    """``{"kv_lens", "ropes", "past_key_values"}`` with cheap copies."""

    @classmethod
    def empty(cls, num_layers: int, batch: int = 1) -> "GenContext":
        return cls(kv_lens=[0] * batch, ropes=[0] * batch, past_key_values=NaiveCache(num_layers))

    @classmethod
    def of(cls, mapping) -> "GenContext":
        return mapping if isinstance(mapping, cls) else cls(mapping)

    def fork(self) -> "GenContext":
        """An independent context that starts equal to this one.  Lists and the per-layer dicts are copied, the
        key/value tensors are shared: they are never modified in place."""
        src = self["past_key_values"]
        cache = NaiveCache(src.num_layers)
        cache.key_cache = dict(src.key_cache)
        cache.value_cache = dict(src.value_cache)
        return GenContext(kv_lens=list(self["kv_lens"]), ropes=list(self["ropes"]), past_key_values=cache)

    def advance(self, kv_lens, ropes, past_key_values) -> "GenContext":
        self["kv_lens"], self["ropes"], self["past_key_values"] = kv_lens, ropes, past_key_values
        return self


class CfgContexts:
    """The (full, no_text, no_image) context triple of one request."""

    def __init__(self, inferencer: "InterleaveInferencer"):
        self._inf = inferencer
        self.full = inferencer.init_gen_context()
        self.no_text = self.full.fork()
        self.no_image = self.full.fork()
        # Until the request's first image, ``no_image`` has received exactly what ``full`` has: it is then kept as a
        # fork of ``full`` (same tensors) instead of being prefilled a second time with the same tokens.
        self._text_only = True

    def _text_to_both(self, text: str):
        self._inf.update_context_text(text, self.full)
        if self._text_only:
            self.no_image = self.full.fork()
        else:
            self._inf.update_context_text(text, self.no_image)

    def add_system_prompt(self, prompt: str):
        # a system prompt is part of both the conditional and the image-free context, never of the text-free one
        self._text_to_both(prompt)

    def add_text(self, text: str):
        self.no_text = self.full.fork()                      # everything up to, but not including, this text
        self._text_to_both(text)

    def add_image(self, image, with_vae_tokens: bool):
        self._text_only = False
        self._inf.update_context_image(image, self.full, vae=with_vae_tokens)
        self.no_text = self.full.fork()                      # images stay in the text-free context


class InterleaveInferencer:
    def __init__(self, model, vae_model, tokenizer, vae_transform, vit_transform, new_token_ids):
        self.model = model
        self.vae_model = vae_model
        self.tokenizer = tokenizer
        self.vae_transform = vae_transform
        self.vit_transform = vit_transform
        self.new_token_ids = new_token_ids
        # opt-in (not in the reference): skip CFG branches that were given the conditional branch's own context
        self.dedup_identical_branches = False

    # ---- contexts ---------------------------------------------------------------------------------------
    def init_gen_context(self) -> GenContext:
        return GenContext.empty(self.model.config.llm_config.num_hidden_layers)

    @torch.no_grad()
    def update_context_text(self, text, gen_context):
        """Append one prompt (bos + tokens + eos) to the context: prepare_prompts + forward_cache_update_text."""
        ctx = GenContext.of(gen_context)
        inputs, kv_lens, ropes = self.model.prepare_prompts(curr_kvlens=ctx["kv_lens"], curr_rope=ctx["ropes"], prompts=[text],
                                                            tokenizer=self.tokenizer, new_token_ids=self.new_token_ids)
        cache = self.model.forward_cache_update_text(ctx["past_key_values"], **inputs)
        return self._commit(gen_context, kv_lens, ropes, cache)

    @torch.no_grad()
    def update_context_image(self, image, gen_context, vae=True, vit=True):
        """Append an image as VAE tokens (generation expert) and / or ViT tokens (understanding expert)."""
        assert vae or vit
        ctx = GenContext.of(gen_context)
        kv_lens, ropes, cache = ctx["kv_lens"], ctx["ropes"], ctx["past_key_values"]
        stages = []
        if vae:
            stages.append((self.model.prepare_vae_images, self.vae_transform,
                           lambda c, inp: self.model.forward_cache_update_vae(self.vae_model, c, **inp)))
        if vit:
            stages.append((self.model.prepare_vit_images, self.vit_transform,
                           lambda c, inp: self.model.forward_cache_update_vit(c, **inp)))
        for prepare, transform, prefill in stages:
            inputs, kv_lens, ropes = prepare(curr_kvlens=kv_lens, curr_rope=ropes, images=[image], transforms=transform,
                                             new_token_ids=self.new_token_ids)
            cache = prefill(cache, inputs)
        return self._commit(gen_context, kv_lens, ropes, cache)

    @staticmethod
    def _commit(gen_context, kv_lens, ropes, cache):
        gen_context["kv_lens"], gen_context["ropes"], gen_context["past_key_values"] = kv_lens, ropes, cache
        return gen_context

    # ---- generation -------------------------------------------------------------------------------------
    @torch.no_grad()
    def gen_image(self, image_shape, gen_context, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_text_precontext=None,
                  cfg_img_precontext=None, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
                  num_timesteps=50, timestep_shift=3.0, enable_taylorseer=False):
        model = self.model
        sizes = [image_shape]
        kwargs = dict(model.prepare_vae_latent(curr_kvlens=gen_context["kv_lens"], curr_rope=gen_context["ropes"],
                                               image_sizes=sizes, new_token_ids=self.new_token_ids))
        kwargs["past_key_values"] = gen_context["past_key_values"]
        # each guidance branch contributes the same five arguments under its own prefix
        for prefix, ctx in (("cfg_text", cfg_text_precontext), ("cfg_img", cfg_img_precontext)):
            branch = model.prepare_vae_latent_cfg(curr_kvlens=ctx["kv_lens"], curr_rope=ctx["ropes"], image_sizes=sizes)
            kwargs[f"{prefix}_past_key_values"] = ctx["past_key_values"]
            for name in ("packed_position_ids", "packed_query_indexes", "key_values_lens", "packed_key_value_indexes"):
                kwargs[f"{prefix}_{name}"] = branch[f"cfg_{name}"]
        latents = model.generate_image(num_timesteps=num_timesteps, timestep_shift=timestep_shift,
                                       cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval,
                                       cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type,
                                       enable_taylorseer=enable_taylorseer,
                                       dedup_identical_branches=self.dedup_identical_branches, **kwargs)
        return self.decode_image(latents[0], image_shape)

    def decode_image(self, latent, image_shape):
        """Packed latent tokens ``[h*w, p*p*c]`` -> ``[1, c, h*p, w*p]`` -> VAE decode -> uint8 PIL image."""
        m = self.model
        p, c = m.latent_patch_size, m.latent_channel
        h, w = image_shape[0] // m.latent_downsample, image_shape[1] // m.latent_downsample
        grid = latent.reshape(h, w, p, p, c).permute(4, 0, 2, 1, 3).reshape(1, c, h * p, w * p)
        pixels = self.vae_model.decode(grid)[0]
        pixels = (pixels * 0.5 + 0.5).clamp(0, 1).permute(1, 2, 0) * 255
        return Image.fromarray(pixels.to(torch.uint8).cpu().numpy())

    @torch.no_grad()
    def gen_text(self, gen_context, max_length=500, do_sample=True, temperature=1.0, return_tokens=False):
        """Decode up to ``max_length`` tokens after the context (which is left untouched) and return the text between
        the chat markers."""
        ctx = GenContext.of(gen_context)
        start = self.model.prepare_start_tokens(ctx["kv_lens"], ctx["ropes"], self.new_token_ids)
        tokens = self.model.generate_text(past_key_values=ctx["past_key_values"], max_length=max_length,
                                          do_sample=do_sample, temperature=temperature,
                                          end_token_id=self.new_token_ids["eos_token_id"], **start)[:, 0]
        text = self.tokenizer.decode(tokens)
        text = text.split("<|im_end|>")[0].split("<|im_start|>")[1]
        return (text, tokens.tolist()) if return_tokens else text

    # ---- requests ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def interleave_inference(self, input_lists: List[Union[str, Image.Image]], think=False, understanding_output=False,
                             max_think_token_n=1000, do_sample=False, text_temperature=0.3, cfg_text_scale=3.0,
                             cfg_img_scale=1.5, cfg_interval=[0.4, 1.0], timestep_shift=3.0, num_timesteps=50,
                             cfg_renorm_min=0.0, cfg_renorm_type="global", image_shapes=(1024, 1024),
                             enable_taylorseer=False):
        ctxs = CfgContexts(self)
        if think:
            ctxs.add_system_prompt(VLM_THINK_SYSTEM_PROMPT if understanding_output else GEN_THINK_SYSTEM_PROMPT)
        for term in input_lists:
            if isinstance(term, str):
                ctxs.add_text(term)
            elif isinstance(term, Image.Image):
                resized = self.vae_transform.resize_transform(pil_img2rgb(term))
                ctxs.add_image(resized, with_vae_tokens=not understanding_output)
                image_shapes = resized.size[::-1]            # an edited image keeps its (resized) input's shape
            else:
                raise ValueError(f"Unsupported input type: {type(term)}")

        text_kwargs = dict(do_sample=do_sample, temperature=text_temperature, max_length=max_think_token_n)
        if understanding_output:
            return [self.gen_text(ctxs.full, **text_kwargs)]
        outputs = []
        if think:
            thought = self.gen_text(ctxs.full, **text_kwargs)
            self.update_context_text(thought, ctxs.full)     # the plan conditions the image; the CFG contexts do not see it
            outputs.append(thought)
        outputs.append(self.gen_image(image_shapes, ctxs.full, cfg_text_precontext=ctxs.no_text,
                                      cfg_img_precontext=ctxs.no_image, cfg_text_scale=cfg_text_scale,
                                      cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval, timestep_shift=timestep_shift,
                                      num_timesteps=num_timesteps, cfg_renorm_min=cfg_renorm_min,
                                      cfg_renorm_type=cfg_renorm_type, enable_taylorseer=enable_taylorseer))
        return outputs

    def __call__(self, image: Optional[Image.Image] = None, text: Optional[str] = None, **kargs):
        """``{"image": PIL image or None, "text": str or None}`` for an optional input image and / or prompt."""
        result = {"image": None, "text": None}
        terms = [t for t in (image, text) if t is not None]
        if not terms:
            print("Please provide at least one input: either an image or text.")
            return result
        for out in self.interleave_inference(terms, **kargs):
            result["image" if isinstance(out, Image.Image) else "text"] = out
        return result
