"""Size-independent properties of the other workloads at their full size (BAGEL-7B dims, 28 layers, SigLIP so400m
tower with 26 layers, FLUX VAE) - shapes no CPU checker finishes in seconds, so they are pinned by determinism,
packing / batch invariance and finiteness rather than by golden vectors:

  * image editing: batch 8, three CFG branches with contexts of 9034 / 9000 / 34 tokens (24 units, 98 k rows in
    one packed forward), text_channel renorm (app.py:224-228, eval/gen/gen_images_mp_imgedit.py:278-407);
  * context building from mixed-resolution images: VAE encode + VAE-token prefill, SigLIP tower + connector +
    ViT-token prefill, prompt prefill (bagel.py:876-1143, siglip_navit.py:198-371);
  * think-then-generate: 300 greedily decoded tokens over a 400-token context, pushed back into the context, then
    a 1024 x 1024 generation over it (bagel.py:1526-1596, inferencer.py:273-305).

Random-init weights generated on the device.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16
IDS = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)


class _Tok:
    def encode(self, s):
        return [int(t) for t in s.split()]


@pytest.fixture(scope="module")
def full():
    from bagel_amd.config import BAGEL_7B, VAEDims
    from bagel_amd.engine import MoTWeights
    from bagel_amd.modeling import AutoEncoder, Bagel
    from bagel_amd.weights import make_vae_encoder_state_dict, make_vit_state_dict
    dims = BAGEL_7B
    dev = torch.device(DEV)
    weights = MoTWeights.random(dims, dev, seed=0, vocab_rows=1024)
    model = Bagel(dims, state_dict=make_vit_state_dict(dims, seed=0), device=dev, weights=weights)
    vae = AutoEncoder(VAEDims(), make_vae_encoder_state_dict(VAEDims(), seed=0), dev)
    yield model, vae, dims
    del model, vae
    torch.cuda.empty_cache()


def _tile_cache(cache, times, dims):
    from bagel_amd.modeling import NaiveCache
    out = NaiveCache(dims.num_layers)
    for l in range(dims.num_layers):
        out.key_cache[l] = cache.key_cache[l].repeat(times, 1, 1)
        out.value_cache[l] = cache.value_cache[l].repeat(times, 1, 1)
    return out


def _edit_step(model, caches, lens, x, B, trace):
    res = (1024, 1024)
    c_cond, c_text, c_img = lens
    torch.manual_seed(0)
    gi = model.prepare_vae_latent([c_cond] * B, [3] * B, [res] * B, IDS)
    gt = model.prepare_vae_latent_cfg([c_text] * B, [2] * B, [res] * B)
    gm = model.prepare_vae_latent_cfg([c_img] * B, [c_img] * B, [res] * B)
    gi["packed_init_noises"] = x
    return model.generate_image(
        past_key_values=caches[0], num_timesteps=50, timestep_shift=3.0, cfg_renorm_min=0.0,
        cfg_renorm_type="text_channel", cfg_interval=[0.0, 1.0], cfg_text_scale=4.0, cfg_text_past_key_values=caches[1],
        cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
        cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
        cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"], cfg_img_scale=2.0,
        cfg_img_past_key_values=caches[2], cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
        cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
        cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], max_steps=1, trace=trace, **gi)


def test_edit_batch8_three_long_contexts(full):
    """24 units in one packed forward.  Eight copies of one sample (same contexts, same noise) must each reproduce
    the single-sample result bit for bit (samples only meet in the packed GEMM rows and the tile tables; the
    text_channel renorm is per token), twice in a row (determinism)."""
    from bench import synthetic_context
    model, _, dims = full
    dev = torch.device(DEV)
    lens = (9034, 9000, 34)
    one = [synthetic_context(dims, 1, c, dev, s) for c, s in zip(lens, (1, 2, 3))]
    Nv, P = 4096, dims.patch_latent_dim
    x1 = torch.randn((Nv, P), generator=torch.Generator().manual_seed(8))
    t1 = []
    lat1 = _edit_step(model, one, lens, x1, 1, t1)
    B = 8
    many = [_tile_cache(c, B, dims) for c in one]
    t8, t8b = [], []
    lat8 = _edit_step(model, many, lens, x1.repeat(B, 1), B, t8)
    _edit_step(model, many, lens, x1.repeat(B, 1), B, t8b)
    assert len(lat8) == B and len(t8) == 1
    for k in ("v_cond", "v_text", "v_img", "v_t"):
        v8 = t8[0][k]
        assert v8.shape == (B * Nv, P) and torch.isfinite(v8.float()).all(), k
        assert torch.equal(v8, t8b[0][k]), f"{k}: two identical calls differ"
        for b in range(B):
            assert torch.equal(v8[b * Nv:(b + 1) * Nv], t1[0][k]), f"{k}: sample {b} of the batch differs from the single-sample run"
    for b in range(B):
        assert torch.equal(lat8[b], lat1[0])
    assert 0.05 < float(t1[0]["v_cond"].float().std()) < 50


def test_mixed_resolution_context_building(full):
    """Three images of different size through both image paths, then prompts."""
    from bagel_amd.modeling import NaiveCache
    model, vae, dims = full
    L = dims.num_layers
    g = torch.Generator().manual_seed(0)
    vae_sizes = ((512, 768), (1024, 1024), (768, 576))
    vit_sizes = ((490, 742), (980, 980), (742, 560))
    vae_imgs = [torch.rand((3, h, w), generator=g) * 2 - 1 for h, w in vae_sizes]
    vit_imgs = [torch.rand((3, h, w), generator=g) * 2 - 1 for h, w in vit_sizes]
    ident = lambda x: x  # noqa: E731
    B = 3
    noise = torch.randn((B, 16, 128, 128), generator=g).to(BF16)      # posterior draw of the padded batch

    def snap(past):
        return [past.key_cache[l].clone() for l in range(L)], [past.value_cache[l].clone() for l in range(L)]

    def build(vimgs, timgs, nz, prompt_len=32):
        n = len(vimgs)
        past = NaiveCache(L)
        kv, rope = [0] * n, [0] * n
        stages = {}
        gi, kv, rope = model.prepare_vae_images(kv, rope, vimgs, ident, IDS)
        past = model.forward_cache_update_vae(vae, past, latent_noise=nz, **gi)
        stages["vae"] = (list(kv), *snap(past))
        gi, kv, rope = model.prepare_vit_images(kv, rope, timgs, ident, IDS)
        past = model.forward_cache_update_vit(past, **gi)
        stages["vit"] = (list(kv), *snap(past))
        gi, kv, rope = model.prepare_prompts(kv, rope, [" ".join(str(10 + i) for i in range(prompt_len))] * n, _Tok(), IDS)
        past = model.forward_cache_update_text(past, **gi)
        stages["text"] = (list(kv), *snap(past))
        return kv, rope, past, stages

    kv, rope, past, stages = build(vae_imgs, vit_imgs, noise)
    want = [(h // 16) * (w // 16) + 2 + (hh // 14) * (ww // 14) + 2 + 34 for (h, w), (hh, ww) in zip(vae_sizes, vit_sizes)]
    assert kv == want and rope == [2 + 34] * B
    for l in (0, L - 1):
        assert past.key_cache[l].shape == (sum(kv), dims.num_kv_heads, dims.head_dim)
        assert torch.isfinite(past.key_cache[l].float()).all() and torch.isfinite(past.value_cache[l].float()).all()
        assert 0.05 < float(past.key_cache[l].float().std()) < 50
    # determinism
    _, _, _, stages2 = build(vae_imgs, vit_imgs, noise)
    for st in stages:
        for l in range(L):
            assert torch.equal(stages[st][1][l], stages2[st][1][l]) and torch.equal(stages[st][2][l], stages2[st][2][l]), (st, l)
    del stages2
    # Packing invariance.  The 1024 x 1024 image needs no padding, so alone it must give exactly the keys/values it
    # gives inside the batch, after each stage and in every layer (the smaller images are encoded zero-padded to the
    # batch's largest size, as the reference does, so they legitimately differ from their stand-alone encodes).
    # The prompt rows are the exception: 34 rows alone take the weight-streaming GEMM (split-K partial sums), 3 x 34
    # packed rows the MFMA tile GEMM, so their keys agree to rounding (cosine >= 0.999 in every layer), not bit for bit.
    kv1, _, _, alone = build(vae_imgs[1:2], vit_imgs[1:2], noise[1:2])
    assert kv1 == [kv[1]]
    for st in ("vae", "vit", "text"):
        lens = stages[st][0]
        off, n = lens[0], lens[1]
        assert alone[st][0] == [n]
        n_exact = n - 34 if st == "text" else n
        for l in range(L):
            for which in (1, 2):
                got, ref = alone[st][which][l], stages[st][which][l][off:off + n]
                assert torch.equal(got[:n_exact], ref[:n_exact]), f"after the {st} stage, layer {l}, {'keys' if which == 1 else 'values'}"
                if n_exact < n:
                    a, b_ = got[n_exact:].float().flatten(), ref[n_exact:].float().flatten()
                    assert float(torch.dot(a, b_) / (a.norm() * b_.norm())) >= 0.999, (st, l)


def test_vit_tower_so400m_packing_invariance(full):
    """The vision tower alone at its full size (26 layers, 16 heads of 72 zero-padded to 128): an image's tokens do
    not depend on what else is packed with it, and two runs agree bit for bit."""
    model, _, dims = full
    g = torch.Generator().manual_seed(1)
    imgs = [torch.rand((3, h, w), generator=g) * 2 - 1 for h, w in ((490, 742), (980, 980), (154, 70))]
    ident = lambda x: x  # noqa: E731
    gi, _, _ = model.prepare_vit_images([0] * 3, [0] * 3, imgs, ident, IDS)
    lens = gi["vit_token_seqlens"].tolist()
    assert lens == [35 * 53, 70 * 70, 11 * 5]
    a = model.vit_tower.tower(gi["packed_vit_tokens"], gi["packed_vit_position_ids"], lens).clone()
    b = model.vit_tower.tower(gi["packed_vit_tokens"], gi["packed_vit_position_ids"], lens)
    assert a.shape == (sum(lens), dims.vit.hidden_size) and torch.isfinite(a.float()).all() and torch.equal(a, b)
    r = 0
    for i, n in enumerate(lens):
        gi1, _, _ = model.prepare_vit_images([0], [0], imgs[i:i + 1], ident, IDS)
        alone = model.vit_tower.tower(gi1["packed_vit_tokens"], gi1["packed_vit_position_ids"], [n])
        assert torch.equal(alone, a[r:r + n]), f"image {i} ({n} tokens) depends on its batch"
        r += n


def test_think_then_generate(full):
    """300 greedy tokens over a 400-token context (twice: identical; a shorter run is a prefix of the longer one),
    the text pushed back into the context, then two flow steps of a 1024 x 1024 generation over it."""
    from bench import synthetic_context
    model, _, dims = full
    dev = torch.device(DEV)
    w = model.engine.w
    if w.lm_head is None or w.lm_head.shape[0] != 4096:
        # a 4096-row slice of the vocabulary keeps the test's memory modest; the GEMV path is the same
        gen = torch.Generator(device=dev).manual_seed(5)
        w.lm_head = torch.empty((4096, dims.hidden_size), device=dev, dtype=BF16).normal_(0, 0.02, generator=gen)
        w.embed_tokens = torch.empty((4096, dims.hidden_size), device=dev, dtype=BF16).normal_(0, 0.02, generator=gen)
    ctx = 400
    past = synthetic_context(dims, 1, ctx, dev, 5)
    gi = model.prepare_start_tokens([ctx], [ctx], IDS)
    logits = []
    toks = model.generate_text(past_key_values=past, max_length=300, do_sample=False, logits_trace=logits, **gi)
    toks2 = model.generate_text(past_key_values=past, max_length=300, do_sample=False, **gi)
    short = model.generate_text(past_key_values=past, max_length=100, do_sample=False, **gi)
    assert toks.shape == (300, 1) and int(toks[0, 0]) == IDS["bos_token_id"]
    assert torch.equal(toks, toks2) and torch.equal(short, toks[:100])
    assert all(torch.isfinite(l.float()).all() for l in logits[::50])
    assert len(set(toks[:, 0].tolist())) > 3            # not stuck on one token from the first step
    # push the text back (inferencer.py:299-301) and generate over the 702-token context
    text = " ".join(str(int(t)) for t in toks[1:, 0].tolist())
    gp, kv, rope = model.prepare_prompts([ctx], [ctx], [text], _Tok(), IDS)
    past = model.forward_cache_update_text(past, **gp)
    assert kv == [ctx + 299 + 2] and past.key_cache[0].shape[0] == kv[0]
    res = (1024, 1024)
    torch.manual_seed(3)
    g0 = model.prepare_vae_latent(kv, rope, [res], IDS)
    gt = model.prepare_vae_latent_cfg([0], [0], [res])
    gm = model.prepare_vae_latent_cfg(kv, rope, [res])
    from bagel_amd.modeling import NaiveCache

    def run(trace):
        return model.generate_image(
            past_key_values=past, num_timesteps=50, timestep_shift=3.0, cfg_renorm_type="global", cfg_interval=[0.4, 1.0],
            cfg_text_scale=4.0, cfg_text_past_key_values=NaiveCache(dims.num_layers),
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"], cfg_img_scale=1.5,
            cfg_img_past_key_values=past, cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], max_steps=2, trace=trace, **g0)

    ta, tb = [], []
    la, lb = run(ta), run(tb)
    assert torch.equal(la[0], lb[0]) and torch.isfinite(la[0]).all()
    for ra, rb in zip(ta, tb):
        for k in ra:
            assert torch.equal(ra[k], rb[k]), k
    assert torch.equal(ta[0]["v_img"], ta[0]["v_cond"])      # same context, same RoPE position -> same velocity

@pytest.mark.parametrize("batch", [6, 20])
def test_batched_decode_matches_single_sample_logits(full, batch):
    """Greedy decode of several samples at once (6: weight streaming; 20: the tile kernels and the tile-GEMM vocabulary
    projection): twice identical, and every sample's first-step logits equal - within bf16 accumulation-order noise -
    the logits the sample gets when decoded alone; identical contexts give identical rows."""
    from bench import synthetic_context
    from bagel_amd.modeling import NaiveCache
    model, _, dims = full
    dev = torch.device(DEV)
    w = model.engine.w
    if w.lm_head is None or w.lm_head.shape[0] != 4096:
        gen = torch.Generator(device=dev).manual_seed(5)
        w.lm_head = torch.empty((4096, dims.hidden_size), device=dev, dtype=BF16).normal_(0, 0.02, generator=gen)
        w.embed_tokens = torch.empty((4096, dims.hidden_size), device=dev, dtype=BF16).normal_(0, 0.02, generator=gen)
    ctx = 120
    one = synthetic_context(dims, 1, ctx, dev, 9)
    gi1 = model.prepare_start_tokens([ctx], [ctx], IDS)
    l1 = []
    model.generate_text(past_key_values=one, max_length=3, do_sample=False, logits_trace=l1, **gi1)
    many = NaiveCache(dims.num_layers)
    for l in range(dims.num_layers):
        many.key_cache[l] = torch.cat([one.key_cache[l]] * batch)
        many.value_cache[l] = torch.cat([one.value_cache[l]] * batch)
    gib = model.prepare_start_tokens([ctx] * batch, [ctx] * batch, IDS)
    la, lb = [], []
    ta = model.generate_text(past_key_values=many, max_length=12, do_sample=False, logits_trace=la, **gib)
    tb = model.generate_text(past_key_values=many, max_length=12, do_sample=False, logits_trace=lb, **gib)
    assert ta.shape == (12, batch) and torch.equal(ta, tb)
    assert all(torch.equal(x, y) and torch.isfinite(x.float()).all() for x, y in zip(la, lb))
    first = la[0].float()
    assert all(torch.equal(la[0][0], la[0][b]) for b in range(1, batch))      # the same context in every slot
    ref = l1[0].float()[0]
    # 28 layers of bf16 activations through a different kernel form (other summation orders): a few ulps of the largest logit
    err = float((first[0] - ref).abs().max())
    cos = float(torch.dot(first[0], ref) / (first[0].norm() * ref.norm()))
    print(f"batch {batch}: first-step logits vs the sample alone: max-abs {err:.4f}, cosine {cos:.6f}")
    assert err <= 2.0 ** -4 * float(ref.abs().max()) and cos >= 0.999, (err, cos)
