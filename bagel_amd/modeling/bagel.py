"""``Bagel``: the reference's generation API (modeling/bagel/bagel.py:806-1500) over the HIP engine.

Method names, argument names and meanings, return values and error behaviour follow the reference so
that ``InterleaveInferencer`` and the evaluation drivers can call this class unchanged.  What differs is
underneath: all CFG branches of a step run as one packed forward, the KV context lives in a persistent
workspace that is filled once per image, and ``x_t`` stays on the device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import _lib as K
from .. import packing
from ..config import BagelDims
from ..engine import GenUnit, MoTEngine, MoTWeights, TaylorSchedule, TaylorUpdate
from ..parallel import exchange_velocities, plan_exchange, world_info

BF16 = torch.bfloat16


class NaiveCache:
    """Per-layer packed context keys/values, ``[total_ctx_tokens, Hkv, D]`` bf16; ``None`` = empty context
    (reference: qwen2_navit.py:214-228)."""

    def __init__(self, num_layers):
        self.key_cache = {k: None for k in range(num_layers)}
        self.value_cache = {k: None for k in range(num_layers)}

    @property
    def num_layers(self):
        return len(self.key_cache)

    @property
    def seq_lens(self):
        if self.key_cache[0] is not None:
            return self.key_cache[0].shape[0]
        return 0


get_flattened_position_ids_extrapolate = packing.flattened_position_ids_extrapolate


class _LLMConfig:
    def __init__(self, dims: BagelDims):
        self.num_hidden_layers = dims.num_layers
        self.hidden_size = dims.hidden_size
        self.num_attention_heads = dims.num_heads
        self.num_key_value_heads = dims.num_kv_heads
        self.intermediate_size = dims.intermediate_size
        self.layer_module = "Qwen2MoTDecoderLayer"


class _Config:
    def __init__(self, dims: BagelDims):
        self.llm_config = _LLMConfig(dims)
        self.latent_patch_size = dims.latent_patch_size
        self.max_latent_size = dims.max_latent_size


class Bagel:
    def __init__(self, dims: BagelDims, state_dict: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0",
                 weights: Optional[MoTWeights] = None):
        K.lib()  # fail loudly when the HIP extension is missing
        self.dims = dims
        self.device = torch.device(device)
        if weights is None:
            if state_dict is None:
                raise ValueError("Bagel needs a state dict (reference parameter names) or prepared MoTWeights")
            weights = MoTWeights.from_state_dict(dims, state_dict, self.device)
        self.engine = MoTEngine(dims, weights, self.device)
        from .qwen2_navit import Qwen2ForCausalLM
        self.language_model = Qwen2ForCausalLM(self.engine, NaiveCache)
        self.vit_tower = None
        if dims.vit is not None and state_dict is not None and "connector.fc1.weight" in state_dict:
            from .siglip import SiglipTower
            self.vit_tower = SiglipTower(dims.vit, dims.hidden_size, state_dict, self.device)
            self.vit_patch_size = dims.vit.patch_size
            self.vit_max_num_patch_per_side = dims.vit.max_patches_per_side
            self.vit_hidden_size = dims.vit.hidden_size
        self.config = _Config(dims)
        self.hidden_size = dims.hidden_size
        self.use_moe = True
        self.num_heads = dims.num_heads
        self.latent_patch_size = dims.latent_patch_size
        self.latent_downsample = dims.latent_downsample
        self.max_latent_size = dims.max_latent_size
        self.latent_channel = dims.latent_channel
        self.patch_latent_dim = dims.patch_latent_dim
        self.get_flattened_position_ids = get_flattened_position_ids_extrapolate
        # set by generate_image for inspection by tests / callers
        self.last_step_velocities: Optional[dict] = None

    def parameters(self):
        yield self.engine.w.embed_tokens

    def eval(self):
        return self

    # ------------------------------------------------------------------------------------------
    # host index builders: thin delegates, the index work lives in bagel_amd/packing.py
    # ------------------------------------------------------------------------------------------
    def prepare_prompts(self, curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids):
        """bagel.py:806-841 (index work in bagel_amd/packing.py)."""
        return packing.prepare_prompts(curr_kvlens, curr_rope, prompts, tokenizer, new_token_ids, device=self.device)

    @K.on_device
    @torch.no_grad()
    def forward_cache_update_text(self, past_key_values: NaiveCache, packed_text_ids, packed_text_position_ids,
                                  text_token_lens, packed_text_indexes, packed_key_value_indexes, key_values_lens):
        """bagel.py:844-874: causal understanding-mode prefill of packed prompts; returns the updated cache."""
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        lens = [int(x) for x in cpu(text_token_lens).tolist()]
        kvl = [int(x) for x in cpu(key_values_lens).tolist()]
        pos = cpu(packed_text_position_ids).tolist()
        qi, ki = cpu(packed_text_indexes).tolist(), cpu(packed_key_value_indexes).tolist()
        pos0, ctx_k, ctx_v = [], [], []
        q = c = tot = 0
        L = self.dims.num_layers
        for b, n in enumerate(lens):
            if pos[q:q + n] != list(range(pos[q], pos[q] + n)):
                raise NotImplementedError("text position ids must be consecutive within a prompt (bagel.py:823)")
            if ki[c:c + kvl[b]] != list(range(tot, tot + kvl[b])) or qi[q:q + n] != list(range(tot + kvl[b], tot + kvl[b] + n)):
                raise NotImplementedError("packed indexes are not the contiguous per-sample layout of prepare_prompts")
            pos0.append(pos[q])
            if kvl[b] > 0:
                ctx_k.append([past_key_values.key_cache[l][c:c + kvl[b]].to(self.device, BF16) for l in range(L)])
                ctx_v.append([past_key_values.value_cache[l][c:c + kvl[b]].to(self.device, BF16) for l in range(L)])
            else:
                ctx_k.append(None)
                ctx_v.append(None)
            q += n
            c += kvl[b]
            tot += kvl[b] + n
        ids = cpu(packed_text_ids).to(torch.int32).to(self.device)
        emb = torch.empty((sum(lens), self.dims.hidden_size), device=self.device, dtype=BF16)
        K.gather_rows(self.engine.w.embed_tokens, ids, emb, sum(lens))
        keys, values = self.engine.prefill_und(emb, lens, pos0, kvl, ctx_k, ctx_v, is_causal=True)
        # like the reference, the given cache object is updated and returned (qwen2_navit.py:607-609)
        for l in range(L):
            past_key_values.key_cache[l] = keys[l]
            past_key_values.value_cache[l] = values[l]
        return past_key_values

    # ------------------------------------------------------------------------------------------
    # image inputs: ViT tokens (understanding) and VAE tokens (generation expert) into the context
    # ------------------------------------------------------------------------------------------
    def _image_context(self, past_key_values, packed_seqlens, packed_position_ids, packed_indexes, key_values_lens,
                       packed_key_value_indexes, packed_text_indexes, image_token_indexes):
        """Validate the [soi, image tokens, eoi] per-sample layout and slice the context per sample."""
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        seqlens = [int(x) for x in cpu(packed_seqlens).tolist()]
        self._check_packing(seqlens, cpu(packed_text_indexes), cpu(image_token_indexes))
        return seqlens, self._branch_units(0, seqlens, cpu(packed_position_ids), cpu(packed_indexes), past_key_values,
                                           cpu(key_values_lens), cpu(packed_key_value_indexes))

    def prepare_vit_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids):
        """bagel.py:876-936."""
        if self.dims.vit is None:
            raise ValueError("this model was built without a vision tower (dims.vit is None)")
        vit = self.dims.vit
        return packing.prepare_vit_images(curr_kvlens, curr_rope, images, transforms, new_token_ids, vit.patch_size,
                                          vit.max_patches_per_side)

    @K.on_device
    @torch.no_grad()
    def forward_cache_update_vit(self, past_key_values: NaiveCache, packed_text_ids, packed_text_indexes,
                                 packed_vit_tokens, packed_vit_token_indexes, packed_vit_position_ids, vit_token_seqlens,
                                 packed_position_ids, packed_seqlens, packed_indexes, packed_key_value_indexes,
                                 key_values_lens):
        """bagel.py:938-1008: SigLIP tower + connector, then a non-causal understanding-mode prefill of
        [start_of_image, ViT tokens, end_of_image] per sample.  Updates and returns the cache."""
        if self.vit_tower is None:
            raise ValueError("this model was built without vision-tower weights")
        if vit_token_seqlens is None or len(vit_token_seqlens) == 0:
            raise ValueError("vit_token_seqlens must hold one length per image")
        if int(torch.min(vit_token_seqlens)) < 0:
            raise ValueError(f"vit_token_seqlens holds a negative length: {vit_token_seqlens}")
        seqlens, units = self._image_context(past_key_values, packed_seqlens, packed_position_ids, packed_indexes,
                                             key_values_lens, packed_key_value_indexes, packed_text_indexes,
                                             packed_vit_token_indexes)
        dev, H = self.device, self.dims.hidden_size
        ve = self.vit_tower.embed(packed_vit_tokens, packed_vit_position_ids, vit_token_seqlens.tolist())
        ids = packed_text_ids.detach().cpu().to(torch.int32).to(dev)
        te = torch.empty((ids.numel(), H), device=dev, dtype=BF16)
        K.gather_rows(self.engine.w.embed_tokens, ids, te, ids.numel())
        M = sum(seqlens)
        seq = torch.empty((M, H), device=dev, dtype=BF16)
        seq[packed_text_indexes.to(dev)] = te
        seq[packed_vit_token_indexes.to(dev)] = ve
        keys, values = self.engine.prefill_und(seq, seqlens, [u.rope_pos for u in units], [u.ctx_len for u in units],
                                               [u.ctx_k for u in units], [u.ctx_v for u in units], is_causal=False,
                                               same_position=True)
        for l in range(self.dims.num_layers):
            past_key_values.key_cache[l] = keys[l]
            past_key_values.value_cache[l] = values[l]
        return past_key_values

    def prepare_vae_images(self, curr_kvlens, curr_rope, images, transforms, new_token_ids, timestep=0):
        """bagel.py:1010-1081."""
        return packing.prepare_vae_images(curr_kvlens, curr_rope, images, transforms, new_token_ids,
                                          self.latent_downsample, self.max_latent_size, timestep=timestep)

    @K.on_device
    @torch.no_grad()
    def forward_cache_update_vae(self, vae_model, past_key_values: NaiveCache, padded_images, patchified_vae_latent_shapes,
                                 packed_vae_position_ids, packed_timesteps, packed_vae_token_indexes, packed_text_ids,
                                 packed_text_indexes, packed_position_ids, packed_seqlens, packed_indexes, key_values_lens,
                                 packed_key_value_indexes, latent_noise: Optional[torch.Tensor] = None):
        """bagel.py:1083-1143: VAE-encode the images, embed the latent patches (vae2llm + t_emb + pos) and
        prefill them through the generation expert (non-causal; start/end tokens through the understanding
        expert).  ``latent_noise`` (not in the reference) fixes the Gaussian draw of the VAE posterior, which
        the reference takes from the device RNG.  Updates and returns the cache."""
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        eng, d, dev = self.engine, self.dims, self.device
        seqlens, units = self._image_context(past_key_values, packed_seqlens, packed_position_ids, packed_indexes,
                                             key_values_lens, packed_key_value_indexes, packed_text_indexes,
                                             packed_vae_token_indexes)
        tt = cpu(packed_timesteps)
        if tt.numel() != 1:
            raise NotImplementedError("one timestep for all VAE tokens is supported (prepare_vae_images passes one)")
        if not tt.dtype.is_floating_point and int(tt[0]) != 0:
            # modeling_utils.py:98-103 casts the frequency table to the timestep's dtype: an integer timestep
            # truncates the table to [1, 0, 0, ...]; only 0 (what every caller passes) is meaningful
            raise NotImplementedError("integer timesteps other than 0 are not supported")
        padded_latent = vae_model.encode(padded_images, noise=latent_noise)
        q, c = self.latent_patch_size, self.latent_channel
        packed = []
        for latent, (h, w) in zip(padded_latent, patchified_vae_latent_shapes):
            latent = latent[:, :h * q, :w * q].reshape(c, h, q, w, q)
            packed.append(torch.einsum("chpwq->hwpqc", latent).reshape(-1, q * q * c))
        packed = torch.cat(packed, dim=0).to(BF16).contiguous()
        sample_nv = [n - 2 for n in seqlens]
        if packed.shape[0] != sum(sample_nv):
            raise ValueError("patchified_vae_latent_shapes do not match packed_seqlens")
        sample_row0 = [sum(sample_nv[:b]) for b in range(len(seqlens))]
        plan = eng.build_plan(units, sample_nv, sample_row0)
        eng.load_context(plan)
        for u in units:
            u.ctx_k = u.ctx_v = None
        eng.set_latent_position_ids(cpu(packed_vae_position_ids))
        ids = cpu(packed_text_ids).to(torch.int32).to(dev)
        text_emb = torch.empty((plan.Mt, d.hidden_size), device=dev, dtype=BF16)
        K.gather_rows(eng.w.embed_tokens, ids, text_emb, plan.Mt)
        temb = eng.timestep_table(tt.float())
        eng.forward_gen(plan, packed, temb[0], text_emb)
        keys, values = eng.extract_gen_kv(plan)
        for l in range(d.num_layers):
            past_key_values.key_cache[l] = keys[l]
            past_key_values.value_cache[l] = values[l]
        return past_key_values

    def prepare_start_tokens(self, curr_kvlens, curr_rope, new_token_ids):
        """bagel.py:1502-1523."""
        return packing.prepare_start_tokens(curr_kvlens, curr_rope, new_token_ids, device=self.device)

    @K.on_device
    @torch.no_grad()
    def generate_text(self, past_key_values: NaiveCache, packed_key_value_indexes, key_values_lens, packed_start_tokens,
                      packed_query_position_ids, max_length: int, do_sample: bool = False, temperature: float = 1.0,
                      end_token_id: int = None, logits_trace: Optional[list] = None):
        """bagel.py:1526-1596.  Returns the fed tokens ``[steps, B]`` (the start token first).  Greedy decoding
        (``do_sample=False``) is fully on HIP kernels; with ``do_sample=True`` the token is drawn with
        torch.multinomial from softmax(logits / temperature), as the reference does (its draws depend on the
        device RNG and are not reproducible across implementations either).  The given cache is not modified."""
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        kvl = [int(x) for x in cpu(key_values_lens).tolist()]
        ki = cpu(packed_key_value_indexes).tolist()
        if ki != list(range(sum(kvl))):
            raise NotImplementedError("packed_key_value_indexes is not the contiguous layout of prepare_start_tokens")
        L = self.dims.num_layers
        ctx_k, ctx_v = [], []
        c = 0
        for n in kvl:
            if n > 0:
                ctx_k.append([past_key_values.key_cache[l][c:c + n].to(self.device, BF16) for l in range(L)])
                ctx_v.append([past_key_values.value_cache[l][c:c + n].to(self.device, BF16) for l in range(L)])
            else:
                ctx_k.append(None)
                ctx_v.append(None)
            c += n
        sample_fn = None
        if do_sample:
            def sample_fn(logits):
                probs = torch.softmax(logits.float() / temperature, dim=-1)
                return torch.multinomial(probs, num_samples=1).squeeze(1)
        return self.engine.decode_tokens(cpu(packed_start_tokens).tolist(), cpu(packed_query_position_ids).tolist(), kvl,
                                         ctx_k, ctx_v, max_length, end_token_id=end_token_id, sample_fn=sample_fn,
                                         logits_trace=logits_trace)

    def prepare_vae_latent(self, curr_kvlens, curr_rope, image_sizes, new_token_ids):
        """bagel.py:1145-1201."""
        return packing.prepare_vae_latent(curr_kvlens, curr_rope, image_sizes, new_token_ids, self.latent_downsample,
                                          self.max_latent_size, self.patch_latent_dim)

    def prepare_vae_latent_cfg(self, curr_kvlens, curr_rope, image_sizes):
        """bagel.py:1203-1234."""
        return packing.prepare_vae_latent_cfg(curr_kvlens, curr_rope, image_sizes, self.latent_downsample)

    # ------------------------------------------------------------------------------------------
    # translation of the packed-index API into engine units
    # ------------------------------------------------------------------------------------------
    def _check_packing(self, seqlens: List[int], text_indexes, vae_indexes):
        """The engine relies on the canonical per-sample layout [soi, latent tokens, eoi] that
        prepare_vae_latent produces (bagel.py:1156-1186)."""
        t, v = [], []
        q = 0
        for n in seqlens:
            t += [q, q + n - 1]
            v += list(range(q + 1, q + n - 1))
            q += n
        if text_indexes.tolist() != t or vae_indexes.tolist() != v:
            raise NotImplementedError("packed_text_indexes / packed_vae_token_indexes do not follow the "
                                      "[start_of_image, latent tokens, end_of_image] layout of prepare_vae_latent")

    @staticmethod
    def _same_unit(a: GenUnit, b: GenUnit) -> bool:
        """Two units compute the same thing when they serve the same sample at the same RoPE position over the SAME
        context tensors (same storage and extent in every layer: no data is compared)."""
        if (a.sample, a.ctx_len, a.rope_pos) != (b.sample, b.ctx_len, b.rope_pos):
            return False
        if a.ctx_len == 0:
            return True
        same = lambda x, y: x.data_ptr() == y.data_ptr() and x.shape == y.shape and x.stride() == y.stride()  # noqa: E731
        return all(same(x, y) for x, y in zip(a.ctx_k, b.ctx_k)) and all(same(x, y) for x, y in zip(a.ctx_v, b.ctx_v))

    def _branch_units(self, branch: int, seqlens: List[int], position_ids, query_indexes, past: Optional[NaiveCache],
                      kv_lens, kv_indexes) -> List[GenUnit]:
        """Per-sample units of one CFG branch from the reference's packed description of that branch
        (position ids, query slots, context lengths, context slots; bagel.py:1413-1464)."""
        kvl = [int(x) for x in kv_lens.tolist()]
        pos = position_ids.tolist()
        qi = query_indexes.tolist()
        ki = kv_indexes.tolist()
        units = []
        q = c = tot = 0
        for b, n in enumerate(seqlens):
            sample_pos = pos[q:q + n]
            if any(p != sample_pos[0] for p in sample_pos):
                raise NotImplementedError("query tokens of one image must share one RoPE position (bagel.py:1185)")
            # merged layout must be [context of sample b | queries of sample b] (qwen2_navit.py:574-581)
            if ki[c:c + kvl[b]] != list(range(tot, tot + kvl[b])) or qi[q:q + n] != list(range(tot + kvl[b], tot + kvl[b] + n)):
                raise NotImplementedError("packed_indexes / packed_key_value_indexes are not the contiguous "
                                          "per-sample layout produced by prepare_vae_latent")
            u = GenUnit(sample=b, branch=branch, ctx_len=kvl[b], rope_pos=int(sample_pos[0]))
            if kvl[b] > 0:
                if past is None or past.key_cache[0] is None:
                    raise ValueError("key_values_lens > 0 but the NaiveCache is empty")
                u.ctx_k = [past.key_cache[l][c:c + kvl[b]].to(self.device, BF16) for l in range(self.dims.num_layers)]
                u.ctx_v = [past.value_cache[l][c:c + kvl[b]].to(self.device, BF16) for l in range(self.dims.num_layers)]
            units.append(u)
            q += n
            c += kvl[b]
            tot += kvl[b] + n
        if past is not None and past.key_cache[0] is not None and past.key_cache[0].shape[0] != c:
            raise ValueError("NaiveCache length does not match key_values_lens")
        return units

    # ------------------------------------------------------------------------------------------
    # sampler
    # ------------------------------------------------------------------------------------------
    @K.on_device
    @torch.no_grad()
    def generate_image(
        self,
        packed_text_ids: torch.LongTensor,
        packed_text_indexes: torch.LongTensor,
        packed_init_noises: torch.Tensor,
        packed_vae_position_ids: torch.LongTensor,
        packed_vae_token_indexes: torch.LongTensor,
        packed_seqlens: torch.IntTensor,
        packed_position_ids: torch.LongTensor,
        packed_indexes: torch.LongTensor,
        past_key_values: NaiveCache,
        key_values_lens: torch.IntTensor,
        packed_key_value_indexes: torch.LongTensor,
        num_timesteps: int = 24,
        timestep_shift: float = 1.0,
        cfg_renorm_min: float = 0.0,
        cfg_renorm_type: str = "global",
        cfg_interval: Optional[Tuple[float, float]] = [0, 1],
        # cfg_text
        cfg_text_scale: float = 1.0,
        cfg_text_packed_query_indexes: Optional[torch.LongTensor] = None,
        cfg_text_packed_position_ids: Optional[torch.LongTensor] = None,
        cfg_text_past_key_values: Optional[NaiveCache] = None,
        cfg_text_key_values_lens: Optional[torch.IntTensor] = None,
        cfg_text_packed_key_value_indexes: Optional[torch.LongTensor] = None,
        # cfg_img
        cfg_img_scale: float = 1.0,
        cfg_img_packed_query_indexes: Optional[torch.LongTensor] = None,
        cfg_img_packed_position_ids: Optional[torch.LongTensor] = None,
        cfg_img_past_key_values: Optional[NaiveCache] = None,
        cfg_img_key_values_lens: Optional[torch.IntTensor] = None,
        cfg_img_packed_key_value_indexes: Optional[torch.LongTensor] = None,
        cfg_type: str = "parallel",
        # cache_args
        enable_taylorseer=False,
        # additions (not in the reference): stop after this many Euler steps; collect per-step tensors
        max_steps: Optional[int] = None,
        trace: Optional[list] = None,
        teacher_x: Optional[Sequence[torch.Tensor]] = None,
        group=None,
        dedup_identical_branches: bool = False,
    ):
        """Bagel.generate_image (bagel.py:1237-1347).  Returns a tuple of per-image ``[Nv, 64]`` fp32 latents
        (on the device).  ``enable_taylorseer`` skips two of three forwards after a branch's first five and
        extrapolates the last layer's output instead (modeling/cache_utils/taylorseer.py).  Every CFG branch keeps its
        own schedule and factor cache as in the reference (bagel.py:1273-1277): a branch that first runs after step 0
        (a CFG interval that opens late) counts its own steps, so in one Euler step some branches may run the layers
        while others are extrapolated; the branches that run are packed into one forward.
        When torch.distributed is initialised with more than one rank, the (sample, branch) units of each
        step are split over the ranks of ``group`` (bagel_amd/parallel.py); every rank must make the same call
        and every rank returns the same latents.
        ``dedup_identical_branches`` (not in the reference, off by default): when the image-free CFG branch is given
        exactly the conditional branch's context for every sample - what text-to-image requests do
        (inferencer.py:283-285) - its forward is bit-identical to the conditional one, so it is not run and the
        conditional velocity is used for it; results do not change, a third of every CFG step's work goes away."""
        if cfg_renorm_type not in K.RENORM and cfg_text_scale > 1.0:
            raise NotImplementedError(f"{cfg_renorm_type} is not suppoprted")
        eng, d, dev = self.engine, self.dims, self.device
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        seqlens = [int(x) for x in cpu(packed_seqlens).tolist()]
        self._check_packing(seqlens, cpu(packed_text_indexes), cpu(packed_vae_token_indexes))
        sample_nv = [n - 2 for n in seqlens]
        sample_row0 = [sum(sample_nv[:b]) for b in range(len(seqlens))]
        Nv = sum(sample_nv)

        use_text = cfg_text_scale > 1.0
        use_img = use_text and cfg_img_scale > 1.0   # image CFG only acts inside the text-CFG branch (bagel.py:1466-1483)
        units = self._branch_units(0, seqlens, cpu(packed_position_ids), cpu(packed_indexes), past_key_values,
                                   cpu(key_values_lens), cpu(packed_key_value_indexes))
        n_cond = len(units)
        if use_text:
            units += self._branch_units(1, seqlens, cpu(cfg_text_packed_position_ids), cpu(cfg_text_packed_query_indexes),
                                        cfg_text_past_key_values, cpu(cfg_text_key_values_lens),
                                        cpu(cfg_text_packed_key_value_indexes))
        img_is_cond = False
        if use_img:
            img_units = self._branch_units(2, seqlens, cpu(cfg_img_packed_position_ids), cpu(cfg_img_packed_query_indexes),
                                           cfg_img_past_key_values, cpu(cfg_img_key_values_lens),
                                           cpu(cfg_img_packed_key_value_indexes))
            img_is_cond = dedup_identical_branches and all(self._same_unit(a, b) for a, b in zip(units[:n_cond], img_units))
            if img_is_cond and enable_taylorseer:
                # the image-free branch keeps its own TaylorSeer schedule, which only coincides with the conditional
                # one while both have run in every step so far: the CFG interval must open at the first step
                t0 = float(timestep_shift / (1 + (timestep_shift - 1)))      # the first timestep (1 after the shift)
                img_is_cond = cfg_interval[0] < t0 <= cfg_interval[1]
            if not img_is_cond:
                units += img_units

        # split the units over the ranks (one process per GPU); world == 1 keeps everything local
        rank, world = world_info(group)
        ex_full = plan_exchange([u.sample for u in units], [u.branch for u in units], sample_nv, rank, world)
        ex_cond = plan_exchange([u.sample for u in units[:n_cond]], [0] * n_cond, sample_nv, rank, world)
        local_full = [units[i] for i in ex_full.local_units]
        local_cond = [units[i] for i in ex_cond.local_units]   # a prefix of local_full: same key slots
        plan_full = eng.build_plan(local_full, sample_nv, sample_row0) if local_full else None
        plan_cond = eng.build_plan(local_cond, sample_nv, sample_row0) if local_cond else None
        if plan_full is not None:
            eng.load_context(plan_full)
            if plan_cond is not None:
                plan_cond.S = plan_full.S  # same workspace, same strides; the cond units own its first slots
        for u in units:
            u.ctx_k = u.ctx_v = None  # context now lives in the workspace
        eng.set_latent_position_ids(cpu(packed_vae_position_ids))

        # text rows: embed_tokens([soi, eoi]) per unit, constant over steps (bagel.py:1389-1391)
        ids = cpu(packed_text_ids).to(torch.int32)
        text_emb = {}

        # timesteps exactly as the reference builds them, in fp32 on the host (bagel.py:1286-1289)
        timesteps = torch.linspace(1, 0, num_timesteps)
        timesteps = timestep_shift * timesteps / (1 + (timestep_shift - 1) * timesteps)
        dts = timesteps[:-1] - timesteps[1:]
        timesteps = timesteps[:-1]
        temb_table = eng.timestep_table(timesteps)

        x_t = packed_init_noises.to(device=dev, dtype=torch.float32).clone().contiguous()
        x_bf16 = torch.empty((Nv, d.patch_latent_dim), device=dev, dtype=BF16)
        K.cast_f32_bf16(x_t, x_bf16)
        v_t = torch.empty((Nv, d.patch_latent_dim), device=dev, dtype=BF16)
        cfg_ws = torch.empty(K.cfg_workspace_bytes(Nv, d.patch_latent_dim), device=dev, dtype=torch.uint8)

        # text rows of a plan over a subset of the full plan's units
        def text_rows(plan):
            if id(plan) not in text_emb:
                idx = torch.cat([ids[2 * u.sample:2 * u.sample + 2] for u in plan.units]).to(dev)
                te = torch.empty((plan.Mt, d.hidden_size), device=dev, dtype=BF16)
                K.gather_rows(eng.w.embed_tokens, idx, te, plan.Mt)
                text_emb[id(plan)] = te
            return text_emb[id(plan)]

        # TaylorSeer: one schedule and one factor cache per CFG branch as in the reference (bagel.py:1273-1277).  The
        # caches are the row ranges of the branches' latent rows inside the local full plan.
        schedules = factors = None
        branch_rows, sub_plans = {}, {}
        if enable_taylorseer:
            schedules = {b: TaylorSchedule() for b in (0, 1, 2)}
            base = plan_full if plan_full is not None else plan_cond
            if base is not None:
                for u, r0, n in zip(base.units, base.vae_row0, base.nv):
                    a, cnt = branch_rows.get(u.branch, (r0, 0))
                    branch_rows[u.branch] = (a, cnt + n)           # a branch's units are adjacent (branch-major order)
                factors = torch.empty((7, base.Mv * d.hidden_size), device=dev, dtype=BF16)

        def rows_of(plan, branch):
            """(first latent row, rows) of a branch inside ``plan``."""
            r0 = cnt = None
            for u, a, n in zip(plan.units, plan.vae_row0, plan.nv):
                if u.branch == branch:
                    r0, cnt = (a, n) if r0 is None else (r0, cnt + n)
            return r0, cnt

        def plan_for(branches, step_plan):
            """The step's plan restricted to the given branches: same key slots (the workspace is laid out once, by the
            full plan), latent rows compacted."""
            if set(branches) == {u.branch for u in step_plan.units}:
                return step_plan
            key = tuple(branches)
            if key not in sub_plans:
                slot = {(u.sample, u.branch): ks for u, ks in zip(plan_full.units, plan_full.kv_start)}
                us = [u for u in plan_full.units if u.branch in branches]
                sub_plans[key] = eng.build_plan(us, sample_nv, sample_row0, kv_start=[slot[(u.sample, u.branch)] for u in us],
                                                S=plan_full.S)
            return sub_plans[key]

        for i, t in enumerate(timesteps.tolist()):
            if max_steps is not None and i >= max_steps:
                break
            if teacher_x is not None:
                x_t.copy_(teacher_x[i].to(dev, torch.float32))
                K.cast_f32_bf16(x_t, x_bf16)
            active = t > cfg_interval[0] and t <= cfg_interval[1]
            ts_, is_ = (cfg_text_scale, cfg_img_scale) if active else (1.0, 1.0)
            full = active and use_text
            plan, ex = (plan_full, ex_full) if full else (plan_cond, ex_cond)
            if schedules is None:
                vloc = eng.forward_gen(plan, x_bf16, temb_table[i], text_rows(plan)) if plan is not None else None
            else:
                active_br = [0] + ([1] if full else []) + ([2] if (full and use_img and not img_is_cond) else [])
                for b in active_br:
                    schedules[b].advance()
                vloc = None
                if plan is not None:
                    mine = [b for b in active_br if rows_of(plan, b)[0] is not None]      # branches with a unit on this rank
                    run = [b for b in mine if schedules[b].kind == "full"]
                    vloc = eng._get("v_step", (plan.Mv, d.patch_latent_dim))
                    if run:
                        sub = plan_for(run, plan)
                        ups = []
                        for b in run:
                            r0, n = rows_of(sub, b)
                            f0 = branch_rows[b][0]
                            ups.append(TaylorUpdate(r0, n, factors[:, f0 * d.hidden_size:(f0 + n) * d.hidden_size],
                                                    schedules[b].n_factors, schedules[b].distance))
                        vsub = eng.forward_gen(sub, x_bf16, temb_table[i], text_rows(sub), taylor_updates=ups)
                        for b in run:
                            (r0, n), (q0, _) = rows_of(sub, b), rows_of(plan, b)
                            vloc[q0:q0 + n].copy_(vsub[r0:r0 + n])
                    for b in mine:
                        if schedules[b].kind == "Taylor":
                            q0, n = rows_of(plan, b)
                            f0 = branch_rows[b][0]
                            v = eng.taylor_velocity(factors[:, f0 * d.hidden_size:(f0 + n) * d.hidden_size], n,
                                                    schedules[b].n_factors, schedules[b].x)
                            vloc[q0:q0 + n].copy_(v)
                for b in active_br:
                    schedules[b].done()
            vout = exchange_velocities(vloc, ex, d.patch_latent_dim, dev, BF16, group)
            v_cond = vout[:Nv]
            v_text = vout[Nv:2 * Nv] if full else None
            v_img = (v_cond if img_is_cond else vout[2 * Nv:3 * Nv]) if (full and use_img) else None
            if trace is not None:
                rec = {"x_t": x_t.clone(), "v_cond": v_cond.clone()}
                if v_text is not None:
                    rec["v_text"] = v_text.clone()
                if v_img is not None:
                    rec["v_img"] = v_img.clone()
            K.cfg_euler_update(v_cond, v_text, v_img, ts_, is_, cfg_renorm_type, cfg_renorm_min, float(dts[i]),
                               x_t, x_bf16, v_t, cfg_ws)
            if trace is not None:
                rec["v_t"] = v_t.clone()
                trace.append(rec)

        return x_t.split(sample_nv)

    @K.on_device
    @torch.no_grad()
    def _forward_flow(self, x_t, timestep, packed_vae_token_indexes, packed_vae_position_ids, packed_text_ids,
                      packed_text_indexes, packed_indexes, packed_position_ids, packed_seqlens, key_values_lens,
                      past_key_values, packed_key_value_indexes, cfg_renorm_min=0.0, cfg_renorm_type="global",
                      cfg_text_scale=1.0, cfg_text_packed_position_ids=None, cfg_text_packed_query_indexes=None,
                      cfg_text_key_values_lens=None, cfg_text_past_key_values=None,
                      cfg_text_packed_key_value_indexes=None, cfg_img_scale=1.0, cfg_img_packed_position_ids=None,
                      cfg_img_packed_query_indexes=None, cfg_img_key_values_lens=None, cfg_img_past_key_values=None,
                      cfg_img_packed_key_value_indexes=None, cfg_type="parallel", **unused_cache_args):
        """Bagel._forward_flow (bagel.py:1350-1500): the CFG-combined velocity for one timestep.
        Standalone use re-stages the KV context on every call; generate_image does it once per image."""
        tt = timestep.detach().float().cpu()
        assert tt.unique().shape[0] == 1
        eng, d, dev = self.engine, self.dims, self.device
        cpu = lambda t: t.detach().cpu()  # noqa: E731
        seqlens = [int(x) for x in cpu(packed_seqlens).tolist()]
        self._check_packing(seqlens, cpu(packed_text_indexes), cpu(packed_vae_token_indexes))
        sample_nv = [n - 2 for n in seqlens]
        sample_row0 = [sum(sample_nv[:b]) for b in range(len(seqlens))]
        Nv = sum(sample_nv)
        use_text = cfg_text_scale > 1.0
        use_img = use_text and cfg_img_scale > 1.0
        if use_text and cfg_renorm_type not in K.RENORM:
            raise NotImplementedError(f"{cfg_renorm_type} is not suppoprted")
        units = self._branch_units(0, seqlens, cpu(packed_position_ids), cpu(packed_indexes), past_key_values,
                                   cpu(key_values_lens), cpu(packed_key_value_indexes))
        if use_text:
            units += self._branch_units(1, seqlens, cpu(cfg_text_packed_position_ids), cpu(cfg_text_packed_query_indexes),
                                        cfg_text_past_key_values, cpu(cfg_text_key_values_lens),
                                        cpu(cfg_text_packed_key_value_indexes))
        if use_img:
            units += self._branch_units(2, seqlens, cpu(cfg_img_packed_position_ids), cpu(cfg_img_packed_query_indexes),
                                        cfg_img_past_key_values, cpu(cfg_img_key_values_lens),
                                        cpu(cfg_img_packed_key_value_indexes))
        plan = eng.build_plan(units, sample_nv, sample_row0)
        eng.load_context(plan)
        eng.set_latent_position_ids(cpu(packed_vae_position_ids))
        ids = cpu(packed_text_ids).to(torch.int32)
        idx = torch.cat([ids[2 * u.sample:2 * u.sample + 2] for u in plan.units]).to(dev)
        text_emb = torch.empty((plan.Mt, d.hidden_size), device=dev, dtype=BF16)
        K.gather_rows(eng.w.embed_tokens, idx, text_emb, plan.Mt)
        temb = eng.timestep_table(tt[:1])
        x = x_t.to(device=dev, dtype=torch.float32).clone().contiguous()
        x_bf16 = torch.empty((Nv, d.patch_latent_dim), device=dev, dtype=BF16)
        K.cast_f32_bf16(x, x_bf16)
        vout = eng.forward_gen(plan, x_bf16, temb[0], text_emb)
        v_t = torch.empty((Nv, d.patch_latent_dim), device=dev, dtype=BF16)
        cfg_ws = torch.empty(K.cfg_workspace_bytes(Nv, d.patch_latent_dim), device=dev, dtype=torch.uint8)
        K.cfg_euler_update(vout[:Nv], vout[Nv:2 * Nv] if use_text else None, vout[2 * Nv:3 * Nv] if use_img else None,
                           cfg_text_scale, cfg_img_scale, cfg_renorm_type, cfg_renorm_min, 0.0, x, x_bf16, v_t, cfg_ws)
        self.last_step_velocities = {"v_cond": vout[:Nv].clone(),
                                     "v_text": vout[Nv:2 * Nv].clone() if use_text else None,
                                     "v_img": vout[2 * Nv:3 * Nv].clone() if use_img else None}
        return v_t
