"""The CPU checker's image-context paths (SigLIP tower + connector prefill, VAE encode + VAE-token prefill, the
editing path's generation) against golden vectors captured from the reference."""
import copy
import math

import pytest
import torch

from tests.helpers import O, load_golden, metrics
from tests.scenarios import IMAGE_CTX_CASES, NEW_TOKEN_IDS
from bagel_amd.weights import (make_llm_state_dict, make_vit_state_dict, make_vae_encoder_state_dict,
                               make_vae_decoder_state_dict)


def run_oracle_case(ic, g, exact=False):
    """Mirror of tests/golden/make_golden.py:run_image_ctx_case on the checker.  ``exact``: the same graph in fp32 with
    no bf16 rounding point (the exact-arithmetic answer the parity bounds are relative to)."""
    if exact:
        with O.exact_arithmetic():
            return _run_oracle_case(ic, g, True)
    return _run_oracle_case(ic, g, False)


def _run_oracle_case(ic, g, exact):
    dims = ic.dims
    sd = make_llm_state_dict(dims, seed=ic.weight_seed)
    sd.update(make_vit_state_dict(dims, seed=ic.weight_seed))
    vsd = make_vae_encoder_state_dict(dims.vae, seed=0)
    vsd.update(make_vae_decoder_state_dict(dims.vae, seed=0))
    if exact:
        sd = {k: v.float() for k, v in sd.items()}
    n, L = ic.batch, dims.num_layers
    out = {}

    def snap(tag, past):
        out[f"{tag}.k0"], out[f"{tag}.v0"] = past.key_cache[0], past.value_cache[0]
        out[f"{tag}.k_last"], out[f"{tag}.v_last"] = past.key_cache[L - 1], past.value_cache[L - 1]

    with torch.no_grad():
        past = O.Cache(L)
        kvlens, ropes = [0] * n, [0] * n
        if ic.first_prompt is not None:
            gi, kvlens, ropes = O.prepare_prompts(kvlens, ropes, ic.prompts("first"), NEW_TOKEN_IDS)
            past = O.forward_cache_update_text(sd, dims, past, **gi)
        if ic.vae_images is not None:
            gi, kvlens, ropes = O.prepare_vae_images(dims, kvlens, ropes, ic.images("vae"), NEW_TOKEN_IDS)
            out["vae.moments"] = O.vae_encode_moments(vsd, dims.vae, gi["padded_images"]).float()
            past = O.forward_cache_update_vae(sd, vsd, dims, past, noise=g["vae.noise"], **gi)
            snap("after_vae", past)
        if ic.vit_images is not None:
            gi, kvlens, ropes = O.prepare_vit_images(dims, kvlens, ropes, ic.images("vit"), NEW_TOKEN_IDS)
            out["vit.tower"] = O.siglip_forward(sd, dims.vit, gi["packed_vit_tokens"], gi["packed_vit_position_ids"],
                                                gi["vit_token_seqlens"]).float()
            past = O.forward_cache_update_vit(sd, dims, past, **gi)
            snap("after_vit", past)
        out["kv_lens"], out["ropes"] = torch.tensor(kvlens), torch.tensor(ropes)
        if ic.edit_prompt is not None:
            text_ctx = (copy.deepcopy(past), list(kvlens), list(ropes))
            gi, kvlens, ropes = O.prepare_prompts(kvlens, ropes, ic.prompts("edit"), NEW_TOKEN_IDS)
            past = O.forward_cache_update_text(sd, dims, past, **gi)
            gi2, ikv, irope = O.prepare_prompts([0] * n, [0] * n, ic.prompts("edit"), NEW_TOKEN_IDS)
            img_past = O.forward_cache_update_text(sd, dims, O.Cache(L), **gi2)
            torch.manual_seed(ic.noise_seed + 1)
            sizes = [ic.out_image] * n
            gg = O.prepare_vae_latent(dims, kvlens, ropes, sizes, NEW_TOKEN_IDS)
            gt = O.prepare_vae_latent_cfg(dims, text_ctx[1], text_ctx[2], sizes)
            gm = O.prepare_vae_latent_cfg(dims, ikv, irope, sizes)
            out["edit.init_noise"] = gg["packed_init_noises"]
            out["edit.contexts"] = dict(cond=(past, kvlens, ropes), text=text_ctx, img=(img_past, ikv, irope))
            out["edit.inputs"] = (gg, gt, gm)
    return sd, vsd, out


def exact_twins(ic, g):
    """``exact.<name>`` for every float tensor of an image-context golden that the checker also produces, and the
    exact-arithmetic branch velocities of the editing steps (teacher-forced from the golden's x_t)."""
    sd, vsd, out = run_oracle_case(ic, g, exact=True)
    twins = {f"exact.{k}": v.float().clone() for k, v in out.items()
             if isinstance(v, torch.Tensor) and v.is_floating_point() and k in g and k != "edit.init_noise"}
    if ic.edit_prompt is not None:
        ts, _ = O.flow_timesteps(ic.num_timesteps, 3.0)
        for i in range(ic.num_timesteps - 1):
            with torch.no_grad(), O.exact_arithmetic():
                vt, (vc, vx, vm) = edit_flow(ic, sd, out, g[f"edit.step{i}.x_t"].float(), ts[i])
            for k, v in (("v_cond", vc), ("v_text", vx), ("v_img", vm), ("v_t", vt)):
                twins[f"exact.edit.step{i}.{k}"] = v.float().clone()
    return twins


def edit_flow(ic, sd, out, x_t, t):
    gg, gt, gm = out["edit.inputs"]
    c = out["edit.contexts"]
    return O.forward_flow(
        sd, ic.dims, x_t, torch.tensor([t] * x_t.shape[0]), gg["packed_vae_token_indexes"], gg["packed_vae_position_ids"],
        gg["packed_text_ids"], gg["packed_text_indexes"], gg["packed_indexes"], gg["packed_position_ids"],
        gg["packed_seqlens"], gg["key_values_lens"], c["cond"][0], gg["packed_key_value_indexes"],
        cfg_renorm_min=0.0, cfg_renorm_type="text_channel", cfg_text_scale=4.0, cfg_img_scale=2.0, return_branches=True,
        cfg_text_past_key_values=c["text"][0], cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
        cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
        cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_past_key_values=c["img"][0], cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
        cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
        cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"])


def close(got, ref, what, rel=0.03, cos_min=0.999):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    ma, cos = metrics(got, ref)
    assert ma <= rel * float(ref.float().abs().max()) and cos >= cos_min, (what, ma, cos)


@pytest.mark.parametrize("name", list(IMAGE_CTX_CASES))
def test_oracle_image_context_matches_reference(name):
    ic = IMAGE_CTX_CASES[name]
    g = load_golden(name)
    sd, vsd, out = run_oracle_case(ic, g)
    assert torch.equal(out["kv_lens"], g["kv_lens"]) and torch.equal(out["ropes"], g["ropes"])
    for key in g:
        if key in out and isinstance(out[key], torch.Tensor) and key not in ("kv_lens", "ropes"):
            close(out[key], g[key], key)
    if ic.edit_prompt is not None:
        assert torch.equal(out["edit.init_noise"], g["edit.init_noise"])
        ts, _ = O.flow_timesteps(ic.num_timesteps, 3.0)
        for i in range(ic.num_timesteps - 1):
            with torch.no_grad():
                vt, (vc, vx, vm) = edit_flow(ic, sd, out, g[f"edit.step{i}.x_t"], ts[i])
            for k, v in (("v_cond", vc), ("v_text", vx), ("v_img", vm), ("v_t", vt)):
                ref = g[f"edit.step{i}.{k}"]
                ma, cos = metrics(v, ref)
                # 2e-2, or two bf16 ulps at the tensor's top magnitude where that is more (7B-width velocities reach
                # |v| > 4, where one ulp is 0.031); CFG amplifies branch differences by up to
                # (2 s_text - 1)(2 s_img - 1) = 21
                ulp = 2.0 ** (math.floor(math.log2(max(float(ref.float().abs().max()), 1e-30))) - 7)
                bound = max(2e-2, 2 * ulp)
                assert cos >= 0.999 and ma <= (bound if k != "v_t" else 10 * bound), (i, k, ma, cos)
