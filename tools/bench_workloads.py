"""Timings of the other BASELINE workloads at BAGEL-7B dims on one MI355X (random-init weights, synthetic
contexts and images): image editing with long contexts, mixed-resolution context building (VAE encode + SigLIP
tower + prefill), greedy text decoding.  Prints one JSON line per workload.

    python tools/bench_workloads.py edit [--batch 8]
    python tools/bench_workloads.py prefill
    python tools/bench_workloads.py think [--decode-batch 4]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synthetic_context  # noqa: E402
from bagel_amd.config import BAGEL_7B, VAEDims  # noqa: E402
from bagel_amd.engine import MoTWeights  # noqa: E402
from bagel_amd.modeling import AutoEncoder, Bagel, NaiveCache  # noqa: E402

IDS = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)
BF16 = torch.bfloat16


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def edit(model, dims, dev, batch, num_timesteps):
    """BASELINE config 3: editing contexts (VAE 64x64 + 2, ViT 70x70 + 2, 34 text tokens), every step runs three
    branches (cfg_interval (0, 1)), text_channel renorm (app.py:224-228)."""
    res, B = 1024, batch
    c_cond, c_text, c_img = 9034, 9000, 34
    cond, text, img = (synthetic_context(dims, B, c, dev, s) for c, s in ((c_cond, 1), (c_text, 2), (c_img, 3)))
    torch.manual_seed(42)
    gi = model.prepare_vae_latent([c_cond] * B, [3] * B, [(res, res)] * B, IDS)
    gt = model.prepare_vae_latent_cfg([c_text] * B, [2] * B, [(res, res)] * B)
    gm = model.prepare_vae_latent_cfg([c_img] * B, [c_img] * B, [(res, res)] * B)

    def run(max_steps=None):
        return model.generate_image(
            past_key_values=cond, num_timesteps=num_timesteps, timestep_shift=3.0, cfg_renorm_min=0.0,
            cfg_renorm_type="text_channel", cfg_interval=[0.0, 1.0], cfg_text_scale=4.0, cfg_text_past_key_values=text,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"], cfg_img_scale=2.0,
            cfg_img_past_key_values=img, cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], max_steps=max_steps, **gi)

    run(max_steps=1)
    _, dt = timed(run)
    n = num_timesteps - 1
    return {"workload": f"edit: batch {B}, 1024x1024, contexts {c_cond}/{c_text}/{c_img}, {n} steps x 3 branches",
            "seconds": dt, "images_per_sec": B / dt, "ms_per_flow_step_per_image": dt / n / B * 1e3}


def prefill(model, dims, dev):
    """BASELINE config 4: three images of different size through VAE encode + VAE-token prefill and the SigLIP tower
    + connector prefill, then a 32-token prompt (the context an edit request builds)."""
    from bagel_amd.weights import make_vae_encoder_state_dict
    vae = AutoEncoder(VAEDims(), make_vae_encoder_state_dict(VAEDims(), seed=0), dev)
    g = torch.Generator().manual_seed(0)
    vae_imgs = [torch.rand((3, h, w), generator=g) * 2 - 1 for h, w in ((512, 768), (1024, 1024), (768, 576))]
    vit_imgs = [torch.rand((3, h, w), generator=g) * 2 - 1 for h, w in ((490, 742), (980, 980), (742, 560))]
    ident = lambda x: x  # noqa: E731
    B = 3

    class Tok:
        def encode(self, s):
            return [int(t) for t in s.split()]

    def run():
        past = NaiveCache(dims.num_layers)
        kv, rope = [0] * B, [0] * B
        gi, kv, rope = model.prepare_vae_images(kv, rope, vae_imgs, ident, IDS)
        past = model.forward_cache_update_vae(vae, past, **gi)
        gi, kv, rope = model.prepare_vit_images(kv, rope, vit_imgs, ident, IDS)
        past = model.forward_cache_update_vit(past, **gi)
        gi, kv, rope = model.prepare_prompts(kv, rope, [" ".join(str(10 + i) for i in range(32))] * B, Tok(), IDS)
        past = model.forward_cache_update_text(past, **gi)
        return kv

    run()
    kv, dt = timed(run)
    return {"workload": "prefill: 3 mixed-resolution images (VAE 512x768 / 1024x1024 / 768x576, ViT 490x742 / 980x980 / "
                        "742x560) + 32-token prompts", "seconds": dt, "context_tokens": kv,
            "context_tokens_per_sec": sum(kv) / dt}


def prompt(model, dims, dev, tokens, batch=1):
    """A text prompt's prefill alone (the context a text-to-image or a chat request builds): `tokens` ids per sample
    through forward_cache_update_text, weights read once per layer."""
    class Tok:
        def encode(self, s):
            return [int(t) for t in s.split()]

    text = " ".join(str(10 + i % 1000) for i in range(tokens - 2))   # + bos and eos

    def run():
        past = NaiveCache(dims.num_layers)
        gi, kv, rope = model.prepare_prompts([0] * batch, [0] * batch, [text] * batch, Tok(), IDS)
        model.forward_cache_update_text(past, **gi)
        return kv

    run()
    run()
    kv, dt = timed(run)
    w = model.engine.w
    layer_bytes = sum(t.numel() * 2 for lw in w.layers for t in (lw.wqkv[1], lw.wo[1], lw.wgu[1], lw.wd[1]))
    return {"workload": f"prompt: text prefill, {kv[0]} tokens x {batch}", "seconds": dt, "ms": dt * 1e3,
            "tokens_per_sec": sum(kv) / dt, "weight_bytes": layer_bytes, "achieved_GBps_of_weights": layer_bytes / dt / 1e9}


def think(model, dims, dev, max_length, batch=1, ctx=400):
    """BASELINE config 5, text half: greedy decode of max_length tokens over a ctx-token context (per sample)."""
    past = synthetic_context(dims, batch, ctx, dev, 5)
    gi = model.prepare_start_tokens([ctx] * batch, [ctx] * batch, IDS)
    run = lambda: model.generate_text(past_key_values=past, max_length=max_length, do_sample=False, **gi)  # noqa: E731
    model.generate_text(past_key_values=past, max_length=8, do_sample=False, **gi)
    toks, dt = timed(run)
    w = model.engine.w
    bytes_per_token = sum(t.numel() * 2 for lw in w.layers for t in (lw.wqkv[1], lw.wo[1], lw.wgu[1], lw.wd[1])) + w.lm_head.numel() * 2
    return {"workload": f"think: greedy decode, {max_length} tokens, batch {batch}, context {ctx}", "seconds": dt,
            "tokens_per_sec": toks.shape[0] * batch / dt, "ms_per_token": dt / toks.shape[0] * 1e3,
            "weight_bytes_per_token": bytes_per_token, "achieved_GBps": bytes_per_token * toks.shape[0] / dt / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["edit", "prefill", "think", "prompt"])
    ap.add_argument("--prompt-tokens", type=int, default=200, help="tokens per sample (prompt)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--num-timesteps", type=int, default=50)
    ap.add_argument("--max-length", type=int, default=300)
    ap.add_argument("--decode-batch", type=int, default=1, help="samples decoded together (think)")
    ap.add_argument("--context", type=int, default=400, help="context tokens per sample (think)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dims = BAGEL_7B
    weights = MoTWeights.random(dims, dev, seed=0, vocab_rows=1024)
    sd = None
    if args.workload == "prefill":
        from bagel_amd.weights import make_vit_state_dict
        sd = make_vit_state_dict(dims, seed=0)
    if args.workload == "think":
        weights.lm_head = torch.empty((dims.vocab_size, dims.hidden_size), device=dev, dtype=BF16).normal_(0, 0.02)
        weights.embed_tokens = torch.empty((dims.vocab_size, dims.hidden_size), device=dev, dtype=BF16).normal_(0, 0.02)
    model = Bagel(dims, state_dict=sd, device=dev, weights=weights)
    if args.workload == "edit":
        out = edit(model, dims, dev, args.batch, args.num_timesteps)
    elif args.workload == "prefill":
        out = prefill(model, dims, dev)
    elif args.workload == "prompt":
        out = prompt(model, dims, dev, args.prompt_tokens, args.decode_batch)
    else:
        if os.environ.get("DECODE_MIN_CHUNK"):   # development: keys per workgroup of the one-launch decode attention
            from bagel_amd import _lib as K
            K.lib().bagel_set_decode_min_chunk(int(os.environ["DECODE_MIN_CHUNK"]))
        out = think(model, dims, dev, args.max_length, args.decode_batch, args.context)
    out.update({"dims": "BAGEL-7B-MoT (28 layers) + SigLIP so400m (26 layers) + FLUX VAE", "data": "synthetic, random-init weights",
                "n_gpus": 1, "dtype": "bf16"})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
