"""Capture golden vectors from the reference's own PyTorch path (authoring container only).

    python tests/golden/make_golden.py [scenario ...]      # default: all LLM scenarios + VAE

For every scenario of tests/scenarios.py this imports the reference from
/root/reference (see ref_harness.py for the shims), loads the repo's seeded
weights into the reference ``Bagel``, builds the three CFG contexts with the
reference's ``prepare_prompts`` / ``forward_cache_update_text``, runs the
reference's ``generate_image`` and records, per Euler step: the input ``x_t``,
each branch's velocity before CFG, and the combined velocity; plus the final
latent.  Text cases record ``generate_text`` (greedy) tokens and per-step logits.  VAE scenarios record
``AutoEncoder.decode`` outputs.  The files written
here (tests/golden/*.safetensors) are the only thing that travels to the GPU box.
"""
from __future__ import annotations

import json
import os
import sys
import time

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402
from scenarios import SCENARIOS, TAYLOR_SCENARIOS, TEXT_CASES, IMAGE_CTX_CASES, NEW_TOKEN_IDS  # noqa: E402
from bagel_amd.config import VAEDims  # noqa: E402
from bagel_amd.weights import (make_llm_state_dict, make_vae_decoder_state_dict, make_vae_encoder_state_dict,  # noqa: E402
                               make_vit_state_dict)


def build_context(model, NaiveCache, prompts, n_samples, tok):
    """Prefill one CFG context the way InterleaveInferencer.update_context_text does (inferencer.py:57-76)."""
    past = NaiveCache(model.config.llm_config.num_hidden_layers)
    kvlens, ropes = [0] * n_samples, [0] * n_samples
    gi = None
    if prompts is not None:
        gi, kvlens, ropes = model.prepare_prompts(kvlens, ropes, [" ".join(map(str, p)) for p in prompts],
                                                  tok, NEW_TOKEN_IDS)
        gi = {k: v.clone() for k, v in gi.items()}
        past = model.forward_cache_update_text(past, **gi)
    return past, kvlens, ropes, gi


def save_index_tensors(out, tag, gi, extra=None):
    """Record every integer tensor (and list of ints) a prepare_* call of the reference returned, under
    ``idx.<tag>.<name>``: the product's host-side builders are compared with these bit for bit."""
    for k, v in (gi or {}).items():
        if isinstance(v, torch.Tensor):
            if not v.dtype.is_floating_point:
                out[f"idx.{tag}.{k}"] = v.detach().cpu().clone()          # dtype kept: int32 lens, int64 indexes
        elif isinstance(v, (list, tuple)) and len(v) and all(isinstance(e, (list, tuple)) for e in v):
            out[f"idx.{tag}.{k}"] = torch.tensor([list(e) for e in v], dtype=torch.int64)
    for k, v in (extra or {}).items():
        out[f"idx.{tag}.{k}"] = torch.tensor(list(v), dtype=torch.int64)


def run_llm_scenario(sc):
    t0 = time.time()
    sd = make_llm_state_dict(sc.dims, seed=sc.weight_seed, threads=8)
    big = sum(v.numel() for v in sd.values()) > 2e9     # the full-depth model: one copy of the weights only
    model = rh.build_reference_model(sc.dims, sd, lean=big)
    print(f"  [{sc.name}] model built at {time.time() - t0:.1f}s", flush=True)
    from modeling.bagel.qwen2_navit import NaiveCache
    tok = rh.StandInTokenizer()
    n = len(sc.images)
    out = {}
    with torch.no_grad(), rh.autocast():
        ctx = {w: build_context(model, NaiveCache, sc.prompts(w), n, tok) for w in ("cond", "text", "img")}
        torch.manual_seed(sc.noise_seed)
        gi = model.prepare_vae_latent(ctx["cond"][1], ctx["cond"][2], sc.images, NEW_TOKEN_IDS)
        gt = model.prepare_vae_latent_cfg(ctx["text"][1], ctx["text"][2], sc.images)
        gm = model.prepare_vae_latent_cfg(ctx["img"][1], ctx["img"][2], sc.images)
        out["init_noise"] = gi["packed_init_noises"].clone()
        for w in ("cond", "text", "img"):
            save_index_tensors(out, f"prompts.{w}", ctx[w][3], {"newlens": ctx[w][1], "new_rope": ctx[w][2]})
        save_index_tensors(out, "vae_latent", gi)
        save_index_tensors(out, "vae_latent_cfg.text", gt)
        save_index_tensors(out, "vae_latent_cfg.img", gm)

        steps = []
        branch_v = []
        orig_flow = model._forward_flow
        vae_idx = gi["packed_vae_token_indexes"]
        hook = model.llm2vae.register_forward_hook(lambda m, i, o: branch_v.append(o[vae_idx].clone()))

        class Stop(Exception):
            pass

        def flow(**kw):
            if sc.max_steps is not None and len(steps) >= sc.max_steps:
                raise Stop()
            branch_v.clear()
            v = orig_flow(**kw)
            rec = {"x_t": kw["x_t"].clone(), "v_t": v.clone(), "v_cond": branch_v[0]}
            k = 1
            if kw["cfg_text_scale"] > 1.0:
                rec["v_text"] = branch_v[k]
                k += 1
            if kw["cfg_img_scale"] > 1.0:
                rec["v_img"] = branch_v[k]
            steps.append(rec)
            print(f"  [{sc.name}] step {len(steps)} done at {time.time() - t0:.1f}s", flush=True)
            return v

        model._forward_flow = flow
        try:
            latents = model.generate_image(
                past_key_values=ctx["cond"][0], num_timesteps=sc.num_timesteps, timestep_shift=sc.timestep_shift,
                cfg_renorm_min=sc.cfg_renorm_min, cfg_renorm_type=sc.cfg_renorm_type,
                cfg_interval=list(sc.cfg_interval), cfg_text_scale=sc.cfg_text_scale,
                cfg_text_past_key_values=ctx["text"][0],
                cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
                cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
                cfg_text_key_values_lens=gt["cfg_key_values_lens"],
                cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
                cfg_img_scale=sc.cfg_img_scale, cfg_img_past_key_values=ctx["img"][0],
                cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
                cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"],
                cfg_img_key_values_lens=gm["cfg_key_values_lens"],
                cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"],
                enable_taylorseer=sc.enable_taylorseer, **gi)
            out["final_latent"] = torch.cat(list(latents), dim=0).clone()
        except Stop:
            pass
        hook.remove()
        for i, rec in enumerate(steps):
            for k, v in rec.items():
                out[f"step{i}.{k}"] = v
        # context KV of the cond branch, layer 0 and last layer (pins the prefill path)
        pk = ctx["cond"][0]
        if pk.key_cache[0] is not None:
            last = sc.dims.num_layers - 1
            out["cond_ctx.k0"] = pk.key_cache[0].clone()
            out["cond_ctx.v0"] = pk.value_cache[0].clone()
            out["cond_ctx.k_last"] = pk.key_cache[last].clone()
            out["cond_ctx.v_last"] = pk.value_cache[last].clone()
    # Rounding-noise floor of the scenario: distance between the reference's bf16 result and the
    # exact-arithmetic (fp32, no rounding points) evaluation of the same step by the CPU checker.
    from helpers import OracleRun
    noise = {}
    # (skipped TaylorSeer steps have no per-step exact answer)
    exact = None if sc.enable_taylorseer else OracleRun(sc, sd=sd if big else None, exact=True, lazy=big)
    for i, rec in enumerate(steps if exact is not None else []):
        vt, (c, x, m) = exact.step(i, rec["x_t"])
        print(f"  [{sc.name}] exact step {i + 1} done at {time.time() - t0:.1f}s", flush=True)
        for key, t in (("v_cond", c), ("v_text", x), ("v_img", m), ("v_t", vt)):
            if key in rec:
                out[f"step{i}.exact.{key}"] = t.float().clone()
                a, b = rec[key].float().flatten(), t.float().flatten()
                d = (a - b).abs()
                noise[f"step{i}.{key}"] = [float(d.max()), float(d.mean()),
                                           float(torch.dot(a, b) / (a.norm() * b.norm()))]
    if exact is not None and "cond_ctx.k0" in out:      # the prefill's exact-arithmetic keys/values, same layers
        xk = exact.ctx["cond"][0]
        last = sc.dims.num_layers - 1
        out["cond_ctx.exact.k0"] = xk.key_cache[0].float().clone()
        out["cond_ctx.exact.v0"] = xk.value_cache[0].float().clone()
        out["cond_ctx.exact.k_last"] = xk.key_cache[last].float().clone()
        out["cond_ctx.exact.v_last"] = xk.value_cache[last].float().clone()
    if noise:
        print(f"  [{sc.name}] reference-vs-exact noise floor: max {max(v[0] for v in noise.values()):.4f} "
              f"mean {max(v[1] for v in noise.values()):.5f}")
    meta = {"scenario": sc.name, "n_steps": str(len(steps)), "torch": torch.__version__,
            "noise_floor": json.dumps(noise),
            "source": "reference Bagel.generate_image on CPU, bf16 weights + bf16 autocast, SDPA stand-in for flash_attn"}
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, f"{sc.name}.safetensors"), metadata=meta)
    print(f"[{sc.name}] wrote {len(out)} tensors in {time.time() - t0:.1f}s")


def run_text_case(tc):
    """Reference Bagel.generate_text, greedy: the fed tokens and the lm_head output of every step."""
    t0 = time.time()
    sd = make_llm_state_dict(tc.dims, seed=tc.weight_seed, include_lm_head=True)
    model = rh.build_reference_model(tc.dims, sd)
    from modeling.bagel.qwen2_navit import NaiveCache
    tok = rh.StandInTokenizer()
    logits = []
    hook = model.language_model.lm_head.register_forward_hook(lambda m, i, o: logits.append(o.clone()))
    with torch.no_grad(), rh.autocast():
        past, kvlens, ropes, pgi = build_context(model, NaiveCache, tc.prompts(), len(tc.prompt), tok)
        gi = model.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
        start = {k: v.clone() for k, v in gi.items()}   # generate_text edits packed_key_value_indexes in place
        tokens = model.generate_text(past_key_values=past, max_length=tc.max_length, do_sample=False, **gi)
    hook.remove()
    out = {"tokens": tokens.to(torch.int64).contiguous(), "logits": torch.stack(logits).float().contiguous()}
    for k, v in start.items():
        out[f"start.{k}"] = v.to(torch.int64).contiguous()
    save_index_tensors(out, "prompts", pgi, {"newlens": kvlens, "new_rope": ropes})
    save_file(out, os.path.join(HERE, f"{tc.name}.safetensors"),
              metadata={"source": "reference Bagel.generate_text (greedy) on CPU, bf16 weights + bf16 autocast, "
                                  "SDPA stand-in for flash_attn"})
    print(f"[{tc.name}] tokens {tokens.T.tolist()} in {time.time() - t0:.1f}s")


def run_image_ctx_case(ic):
    """Reference prepare_vae_images / forward_cache_update_vae / prepare_vit_images / forward_cache_update_vit on a
    packed batch (identity stand-in for the image transforms), then, for an edit case, generate_image with the
    editing path's three CFG contexts."""
    import copy
    t0 = time.time()
    sd = make_llm_state_dict(ic.dims, seed=ic.weight_seed)
    sd.update(make_vit_state_dict(ic.dims, seed=ic.weight_seed))
    model = rh.build_reference_model(ic.dims, sd)
    vd = ic.dims.vae
    vsd = make_vae_encoder_state_dict(vd, seed=0)
    vsd.update(make_vae_decoder_state_dict(vd, seed=0))
    vae = rh.build_reference_vae(vd, vsd)
    from modeling.bagel.qwen2_navit import NaiveCache
    tok = rh.StandInTokenizer()
    ident = lambda x: x  # noqa: E731
    n = ic.batch
    out = {}
    L = ic.dims.num_layers

    def snap(tag, past):
        out[f"{tag}.k0"] = past.key_cache[0].clone()
        out[f"{tag}.v0"] = past.value_cache[0].clone()
        out[f"{tag}.k_last"] = past.key_cache[L - 1].clone()
        out[f"{tag}.v_last"] = past.value_cache[L - 1].clone()

    with torch.no_grad(), rh.autocast():
        past = NaiveCache(L)
        kvlens, ropes = [0] * n, [0] * n
        if ic.first_prompt is not None:
            gi, kvlens, ropes = model.prepare_prompts(kvlens, ropes, [" ".join(map(str, p)) for p in ic.prompts("first")],
                                                      tok, NEW_TOKEN_IDS)
            save_index_tensors(out, "prompts.first", gi, {"newlens": kvlens, "new_rope": ropes})
            past = model.forward_cache_update_text(past, **gi)
        if ic.vae_images is not None:
            gi, kvlens, ropes = model.prepare_vae_images(kvlens, ropes, ic.images("vae"), ident, NEW_TOKEN_IDS)
            save_index_tensors(out, "vae_images", gi, {"newlens": kvlens, "new_rope": ropes})
            hooks = [vae.encoder.register_forward_hook(lambda m, i, o: out.__setitem__("vae.moments", o.float().clone()))]
            real_randn_like = torch.randn_like

            def randn_like(x, *a, **k):
                r = real_randn_like(x, *a, **k)
                out["vae.noise"] = r.float().clone()
                return r

            torch.manual_seed(ic.noise_seed)
            torch.randn_like = randn_like
            try:
                past = model.forward_cache_update_vae(vae, past, **gi)
            finally:
                torch.randn_like = real_randn_like
                for h in hooks:
                    h.remove()
            snap("after_vae", past)
        ctx_before_vit = None
        if ic.vit_images is not None:
            gi, kvlens, ropes = model.prepare_vit_images(kvlens, ropes, ic.images("vit"), ident, NEW_TOKEN_IDS)
            save_index_tensors(out, "vit_images", gi, {"newlens": kvlens, "new_rope": ropes})
            hooks = [model.vit_model.register_forward_hook(lambda m, i, o: out.__setitem__("vit.tower", o.float().clone())),
                     model.connector.register_forward_hook(lambda m, i, o: out.__setitem__("vit.connector", o.float().clone()))]
            past = model.forward_cache_update_vit(past, **gi)
            for h in hooks:
                h.remove()
            snap("after_vit", past)
        out["kv_lens"] = torch.tensor(kvlens)
        out["ropes"] = torch.tensor(ropes)
        if ic.edit_prompt is not None:
            # editing path of interleave_inference (inferencer.py:281-292): the text-free context is the one
            # before the prompt; the image-free context holds the prompt alone
            text_ctx = (copy.deepcopy(past), list(kvlens), list(ropes))
            prompts = [" ".join(map(str, p)) for p in ic.prompts("edit")]
            gi, kvlens, ropes = model.prepare_prompts(kvlens, ropes, prompts, tok, NEW_TOKEN_IDS)
            past = model.forward_cache_update_text(past, **gi)
            gi2, ikv, irope = model.prepare_prompts([0] * n, [0] * n, prompts, tok, NEW_TOKEN_IDS)
            img_past = model.forward_cache_update_text(NaiveCache(L), **gi2)
            torch.manual_seed(ic.noise_seed + 1)
            sizes = [ic.out_image] * n
            g = model.prepare_vae_latent(kvlens, ropes, sizes, NEW_TOKEN_IDS)
            gt = model.prepare_vae_latent_cfg(text_ctx[1], text_ctx[2], sizes)
            gm = model.prepare_vae_latent_cfg(ikv, irope, sizes)
            save_index_tensors(out, "edit.vae_latent", g)
            save_index_tensors(out, "edit.vae_latent_cfg.text", gt)
            save_index_tensors(out, "edit.vae_latent_cfg.img", gm)
            out["edit.init_noise"] = g["packed_init_noises"].clone()
            steps, branch_v = [], []
            vae_idx = g["packed_vae_token_indexes"]
            hook = model.llm2vae.register_forward_hook(lambda m, i, o: branch_v.append(o[vae_idx].clone()))
            orig_flow = model._forward_flow

            def flow(**kw):
                branch_v.clear()
                v = orig_flow(**kw)
                steps.append({"x_t": kw["x_t"].clone(), "v_t": v.clone(), "v_cond": branch_v[0], "v_text": branch_v[1],
                              "v_img": branch_v[2]})
                return v

            model._forward_flow = flow
            latents = model.generate_image(
                past_key_values=past, num_timesteps=ic.num_timesteps, timestep_shift=3.0, cfg_renorm_min=0.0,
                cfg_renorm_type="text_channel", cfg_interval=[0.0, 1.0], cfg_text_scale=4.0,
                cfg_text_past_key_values=text_ctx[0], cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
                cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
                cfg_text_key_values_lens=gt["cfg_key_values_lens"],
                cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
                cfg_img_scale=2.0, cfg_img_past_key_values=img_past,
                cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
                cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"],
                cfg_img_key_values_lens=gm["cfg_key_values_lens"],
                cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], **g)
            hook.remove()
            out["edit.final_latent"] = torch.cat(list(latents), dim=0).clone()
            for i, rec in enumerate(steps):
                for k, v in rec.items():
                    out[f"edit.step{i}.{k}"] = v
    # exact-arithmetic twins (fp32, no rounding points) of every float tensor, from the CPU checker
    from test_oracle_image_ctx import exact_twins
    out.update(exact_twins(ic, out))
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, f"{ic.name}.safetensors"),
              metadata={"source": "reference Bagel image-context prefill (+ edit generation) on CPU, bf16 weights + bf16 "
                                  "autocast, fp32 VAE, SDPA stand-in for flash_attn, identity image transforms"})
    print(f"[{ic.name}] wrote {len(out)} tensors in {time.time() - t0:.1f}s")


VAE_CASES = {"vae_128": (16, 16), "vae_256": (32, 32), "vae_512x384": (64, 48), "vae_1024": (128, 128)}


def run_vae(name, latent_hw):
    t0 = time.time()
    vd = VAEDims()
    sd = make_vae_decoder_state_dict(vd, seed=0)
    ae = rh.build_reference_vae(vd, sd)
    g = torch.Generator().manual_seed(7)
    z = torch.randn(1, vd.z_channels, *latent_hw, generator=g)
    with torch.no_grad():
        img32 = ae.decode(z)
        with rh.autocast():
            img16 = ae.decode(z)
    big = latent_hw[0] * latent_hw[1] > 64 * 64        # 1024 x 1024: images stored in fp16 (values in [-1.3, 1.3])
    cast = (lambda t: t.half().contiguous()) if big else (lambda t: t.float().contiguous())
    d = (img32.double() - img16.double())
    meta = {"source": "reference AutoEncoder.decode on CPU, fp32 and under bf16 autocast",
            "sum_fp32": repr(float(img32.double().sum())), "sum_autocast": repr(float(img16.double().sum())),
            "psnr_autocast_vs_fp32": repr(float(10 * torch.log10(4.0 / (d * d).mean())))}
    save_file({"z": z, "image_fp32": cast(img32), "image_autocast": cast(img16)},
              os.path.join(HERE, f"{name}.safetensors"), metadata=meta)
    print(f"[{name}] wrote in {time.time() - t0:.1f}s; |fp32-autocast|max={float((img32 - img16.float()).abs().max()):.4f}")


if __name__ == "__main__":
    names = sys.argv[1:] or (list(SCENARIOS) + list(TAYLOR_SCENARIOS) + list(TEXT_CASES) + list(IMAGE_CTX_CASES) + list(VAE_CASES))
    for nme in names:
        if nme in SCENARIOS or nme in TAYLOR_SCENARIOS:
            run_llm_scenario(SCENARIOS.get(nme) or TAYLOR_SCENARIOS[nme])
        elif nme in TEXT_CASES:
            run_text_case(TEXT_CASES[nme])
        elif nme in IMAGE_CTX_CASES:
            run_image_ctx_case(IMAGE_CTX_CASES[nme])
        else:
            run_vae(nme, VAE_CASES[nme])
