"""``InterleaveInferencer``: the reference's orchestration class (inferencer.py:37-325, 982-1007) over the
HIP-backed ``Bagel`` and ``AutoEncoder``.  Same constructor, same method names, same keyword defaults.

Covered: text prompts -> image (``interleave_inference`` / ``__call__`` with ``text=...``), ``gen_image``,
``decode_image``, context bookkeeping with ``deepcopy`` snapshots for the CFG branches.
``gen_text`` and the think modes (autoregressive decoding through the understanding expert) are covered too.
Image inputs (``update_context_image``: VAE encode + VAE-token prefill, SigLIP tower + connector prefill) are
covered as well; the image transforms are host code in ``bagel_amd/transforms.py``.
"""
from __future__ import annotations

from copy import deepcopy
from typing import List, Union

import torch
from PIL import Image

from .transforms import pil_img2rgb
from .modeling.bagel import NaiveCache


# System prompts of the reference's think modes (inferencer.py:15-21)
VLM_THINK_SYSTEM_PROMPT = '''You should first think about the reasoning process in the mind and then provide the user with the answer. 
The reasoning process is enclosed within <think> </think> tags, i.e. <think> reasoning process here </think> answer here'''

GEN_THINK_SYSTEM_PROMPT = '''You should first think about the planning process in the mind and then generate the image. 
The planning process is enclosed within <think> </think> tags, i.e. <think> planning process here </think> <|vision_start|> image will be generated here <|vision_end|>

When you want to generate an image, use the special tokens <|vision_start|> and <|vision_end|> to indicate where the image should be placed.'''


class InterleaveInferencer:
    def __init__(self, model, vae_model, tokenizer, vae_transform, vit_transform, new_token_ids):
        self.model = model
        self.vae_model = vae_model
        self.tokenizer = tokenizer
        self.vae_transform = vae_transform
        self.vit_transform = vit_transform
        self.new_token_ids = new_token_ids

    def init_gen_context(self):
        """inferencer.py:48-54."""
        return {"kv_lens": [0], "ropes": [0],
                "past_key_values": NaiveCache(self.model.config.llm_config.num_hidden_layers)}

    @torch.no_grad()
    def update_context_text(self, text, gen_context):
        """inferencer.py:57-76."""
        generation_input, kv_lens, ropes = self.model.prepare_prompts(
            curr_kvlens=gen_context["kv_lens"], curr_rope=gen_context["ropes"], prompts=[text],
            tokenizer=self.tokenizer, new_token_ids=self.new_token_ids)
        past_key_values = self.model.forward_cache_update_text(gen_context["past_key_values"], **generation_input)
        gen_context["kv_lens"] = kv_lens
        gen_context["ropes"] = ropes
        gen_context["past_key_values"] = past_key_values
        return gen_context

    @torch.no_grad()
    def update_context_image(self, image, gen_context, vae=True, vit=True):
        """inferencer.py:79-123: append the image's VAE tokens and / or ViT tokens to the context."""
        assert vae or vit
        past_key_values = gen_context["past_key_values"]
        kv_lens, ropes = gen_context["kv_lens"], gen_context["ropes"]
        if vae:
            generation_input, kv_lens, ropes = self.model.prepare_vae_images(
                curr_kvlens=kv_lens, curr_rope=ropes, images=[image], transforms=self.vae_transform,
                new_token_ids=self.new_token_ids)
            past_key_values = self.model.forward_cache_update_vae(self.vae_model, past_key_values, **generation_input)
        if vit:
            generation_input, kv_lens, ropes = self.model.prepare_vit_images(
                curr_kvlens=kv_lens, curr_rope=ropes, images=[image], transforms=self.vit_transform,
                new_token_ids=self.new_token_ids)
            past_key_values = self.model.forward_cache_update_vit(past_key_values, **generation_input)
        gen_context["kv_lens"] = kv_lens
        gen_context["ropes"] = ropes
        gen_context["past_key_values"] = past_key_values
        return gen_context

    @torch.no_grad()
    def gen_image(self, image_shape, gen_context, cfg_text_scale=4.0, cfg_img_scale=1.5, cfg_text_precontext=None,
                  cfg_img_precontext=None, cfg_interval=(0.4, 1.0), cfg_renorm_min=0.0, cfg_renorm_type="global",
                  num_timesteps=50, timestep_shift=3.0, enable_taylorseer=False):
        """inferencer.py:126-198."""
        model = self.model
        generation_input = model.prepare_vae_latent(curr_kvlens=gen_context["kv_lens"], curr_rope=gen_context["ropes"],
                                                    image_sizes=[image_shape], new_token_ids=self.new_token_ids)
        gt = model.prepare_vae_latent_cfg(curr_kvlens=cfg_text_precontext["kv_lens"],
                                          curr_rope=cfg_text_precontext["ropes"], image_sizes=[image_shape])
        gm = model.prepare_vae_latent_cfg(curr_kvlens=cfg_img_precontext["kv_lens"],
                                          curr_rope=cfg_img_precontext["ropes"], image_sizes=[image_shape])
        unpacked_latent = model.generate_image(
            past_key_values=gen_context["past_key_values"],
            cfg_text_past_key_values=cfg_text_precontext["past_key_values"],
            cfg_img_past_key_values=cfg_img_precontext["past_key_values"],
            num_timesteps=num_timesteps, cfg_text_scale=cfg_text_scale, cfg_img_scale=cfg_img_scale,
            cfg_interval=cfg_interval, cfg_renorm_min=cfg_renorm_min, cfg_renorm_type=cfg_renorm_type,
            timestep_shift=timestep_shift, **generation_input,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
            cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=gm["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"],
            enable_taylorseer=enable_taylorseer)
        return self.decode_image(unpacked_latent[0], image_shape)

    def decode_image(self, latent, image_shape):
        """inferencer.py:201-212: un-patchify, VAE decode, uint8 PIL image."""
        H, W = image_shape
        m = self.model
        h, w = H // m.latent_downsample, W // m.latent_downsample
        p, c = m.latent_patch_size, m.latent_channel
        latent = latent.reshape(1, h, w, p, p, c)
        latent = torch.einsum("nhwpqc->nchpwq", latent)
        latent = latent.reshape(1, c, h * p, w * p)
        image = self.vae_model.decode(latent)
        image = (image * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255
        return Image.fromarray(image.to(torch.uint8).cpu().numpy())

    @torch.no_grad()
    def gen_text(self, gen_context, max_length=500, do_sample=True, temperature=1.0, return_tokens=False):
        """inferencer.py:215-244."""
        gen_context = deepcopy(gen_context)
        generation_input = self.model.prepare_start_tokens(gen_context["kv_lens"], gen_context["ropes"], self.new_token_ids)
        unpacked_latent = self.model.generate_text(
            past_key_values=gen_context["past_key_values"], max_length=max_length, do_sample=do_sample,
            temperature=temperature, end_token_id=self.new_token_ids["eos_token_id"], **generation_input)
        output = self.tokenizer.decode(unpacked_latent[:, 0])
        output = output.split("<|im_end|>")[0].split("<|im_start|>")[1]
        if return_tokens:
            return output, unpacked_latent[:, 0].tolist()
        return output

    @torch.no_grad()
    def interleave_inference(self, input_lists: List[Union[str, Image.Image]], think=False, understanding_output=False,
                             max_think_token_n=1000, do_sample=False, text_temperature=0.3, cfg_text_scale=3.0,
                             cfg_img_scale=1.5, cfg_interval=[0.4, 1.0], timestep_shift=3.0, num_timesteps=50,
                             cfg_renorm_min=0.0, cfg_renorm_type="global", image_shapes=(1024, 1024),
                             enable_taylorseer=False):
        """inferencer.py:247-325 for text inputs and image output."""
        output_list = []
        gen_context = self.init_gen_context()
        cfg_text_context = deepcopy(gen_context)
        cfg_img_context = deepcopy(gen_context)
        if think:
            system_prompt = VLM_THINK_SYSTEM_PROMPT if understanding_output else GEN_THINK_SYSTEM_PROMPT
            gen_context = self.update_context_text(system_prompt, gen_context)
            cfg_img_context = self.update_context_text(system_prompt, cfg_img_context)
        for input_term in input_lists:
            if isinstance(input_term, str):
                cfg_text_context = deepcopy(gen_context)
                gen_context = self.update_context_text(input_term, gen_context)
                cfg_img_context = self.update_context_text(input_term, cfg_img_context)
            elif isinstance(input_term, Image.Image):
                input_term = self.vae_transform.resize_transform(pil_img2rgb(input_term))
                gen_context = self.update_context_image(input_term, gen_context, vae=not understanding_output)
                image_shapes = input_term.size[::-1]
                cfg_text_context = deepcopy(gen_context)
            else:
                raise ValueError(f"Unsupported input type: {type(input_term)}")
        if understanding_output:
            output_list.append(self.gen_text(gen_context, do_sample=do_sample, temperature=text_temperature,
                                             max_length=max_think_token_n))
            return output_list
        if think:
            gen_text = self.gen_text(gen_context, do_sample=do_sample, temperature=text_temperature,
                                     max_length=max_think_token_n)
            gen_context = self.update_context_text(gen_text, gen_context)
            output_list.append(gen_text)
        img = self.gen_image(image_shapes, gen_context, cfg_text_precontext=cfg_text_context,
                             cfg_img_precontext=cfg_img_context, cfg_text_scale=cfg_text_scale,
                             cfg_img_scale=cfg_img_scale, cfg_interval=cfg_interval, timestep_shift=timestep_shift,
                             num_timesteps=num_timesteps, cfg_renorm_min=cfg_renorm_min,
                             cfg_renorm_type=cfg_renorm_type, enable_taylorseer=enable_taylorseer)
        output_list.append(img)
        return output_list

    def __call__(self, image=None, text=None, **kargs):
        """inferencer.py:982-1007."""
        output_dict = {"image": None, "text": None}
        if image is None and text is None:
            print("Please provide at least one input: either an image or text.")
            return output_dict
        input_list = []
        if image is not None:
            input_list.append(image)
        if text is not None:
            input_list.append(text)
        for i in self.interleave_inference(input_list, **kargs):
            if isinstance(i, Image.Image):
                output_dict["image"] = i
            elif isinstance(i, str):
                output_dict["text"] = i
        return output_dict
