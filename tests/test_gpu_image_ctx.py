"""Image-context paths on the HIP kernels against the reference's golden vectors: SigLIP tower + connector
prefill, VAE encode + VAE-token prefill, the editing path's generation; plus the kernels they add."""
import copy

import pytest
import torch

from tests.helpers import load_golden, metrics, check_tensor_vs_reference
from tests.scenarios import IMAGE_CTX_CASES, NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16


class _Tok:
    def encode(self, text):
        return [int(t) for t in text.split()]


def _build(ic):
    from bagel_amd.modeling import Bagel, AutoEncoder
    from bagel_amd.weights import (make_llm_state_dict, make_vit_state_dict, make_vae_encoder_state_dict,
                                   make_vae_decoder_state_dict)
    sd = make_llm_state_dict(ic.dims, seed=ic.weight_seed)
    sd.update(make_vit_state_dict(ic.dims, seed=ic.weight_seed))
    vsd = make_vae_encoder_state_dict(ic.dims.vae, seed=0)
    vsd.update(make_vae_decoder_state_dict(ic.dims.vae, seed=0))
    return Bagel(ic.dims, state_dict=sd, device=DEV), AutoEncoder(ic.dims.vae, vsd, device=DEV)


def close(got, ref, what, g, key, case=""):
    """Against the reference's tensor, with every bound relative to the reference's own distance from the
    exact-arithmetic answer ``exact.<key>`` (tests/helpers.py:check_tensor_vs_reference)."""
    assert tuple(got.shape) == tuple(ref.shape), (what, got.shape, ref.shape)
    exact = g.get("exact." + key)
    assert exact is not None, f"no exact-arithmetic twin for {key}: run tests/golden/add_exact_context.py"
    if exact.dim() == ref.dim() + 1:          # a per-sample slice of a batched tensor
        exact = exact[int(what[what.index("[") + 1:what.index("]")])]
    check_tensor_vs_reference(got, ref, exact, case, what)


def _snap_check(tag, past, g, L, case):
    for key, got in (("k0", past.key_cache[0]), ("v0", past.value_cache[0]),
                     ("k_last", past.key_cache[L - 1]), ("v_last", past.value_cache[L - 1])):
        close(got, g[f"{tag}.{key}"], f"{tag}.{key}", g, f"{tag}.{key}", case)


@pytest.mark.parametrize("name", list(IMAGE_CTX_CASES))
def test_image_context_matches_reference(name):
    _check_case(IMAGE_CTX_CASES[name], load_golden(name), name)


def test_random_image_contexts_match_cpu_checker():
    """Random batches of prompts, VAE-encoded and ViT-encoded images (sizes, counts and order of stages vary; mixed
    resolutions inside a batch) through the HIP path and through oracle/ on the same seeded inputs: encoder moments,
    tower output and the context keys / values after every stage.  BAGEL_FUZZ_IMAGE_CTX=n runs n cases (default 3)."""
    import os
    import random
    from bagel_amd.config import tiny_dims
    from tests.scenarios import ImageCtxCase
    from tests.test_oracle_image_ctx import run_oracle_case, exact_twins
    rnd = random.Random(31)
    n_cases = int(os.environ.get("BAGEL_FUZZ_IMAGE_CTX", 3))
    for case in range(n_cases):
        B = rnd.randint(1, 3)
        first = rnd.choice([None, [rnd.randint(1, 40) for _ in range(B)]])
        vae_images = rnd.choice([None, [(16 * rnd.randint(2, 8), 16 * rnd.randint(2, 8)) for _ in range(B)]])
        vit_images = rnd.choice([None, [(14 * rnd.randint(1, 8), 14 * rnd.randint(1, 8)) for _ in range(B)]])
        if vae_images is None and vit_images is None:
            vit_images = [(14 * rnd.randint(1, 8), 14 * rnd.randint(1, 8)) for _ in range(B)]
        ic = ImageCtxCase(f"random{case}", tiny_dims(rnd.randint(1, 3), vit=True), first, vae_images, vit_images,
                          weight_seed=case % 3, image_seed=100 + case)
        g = {}
        if vae_images is not None:   # the posterior draw, one latent grid per padded image
            hm, wm = max(h for h, _ in vae_images), max(w for _, w in vae_images)
            g["vae.noise"] = torch.randn((B, ic.dims.vae.z_channels, hm // 8, wm // 8), generator=torch.Generator().manual_seed(case))
        _, _, out = run_oracle_case(ic, g)
        g.update({k: v for k, v in out.items() if isinstance(v, torch.Tensor)})
        g.update(exact_twins(ic, g))
        try:
            _check_case(ic, g, ic.name)
        except AssertionError as e:
            raise AssertionError(f"case {case}: prompts {first} vae {vae_images} vit {vit_images} L={ic.dims.num_layers}: {e}")
        if case % 10 == 9:
            print(f"  image-context case {case + 1} of {n_cases} ok", flush=True)     # a long soak stays audible under -s
    print(f"{n_cases} random image contexts agree with the CPU checker")


def _check_case(ic, g, case=""):
    from bagel_amd.modeling import NaiveCache
    model, vae = _build(ic)
    ident = lambda x: x  # noqa: E731
    n, L = ic.batch, ic.dims.num_layers
    tok = _Tok()
    past = NaiveCache(L)
    kvlens, ropes = [0] * n, [0] * n
    if ic.first_prompt is not None:
        gi, kvlens, ropes = model.prepare_prompts(kvlens, ropes, [" ".join(map(str, p)) for p in ic.prompts("first")], tok,
                                                  NEW_TOKEN_IDS)
        past = model.forward_cache_update_text(past, **gi)
    if ic.vae_images is not None:
        gi, kvlens, ropes = model.prepare_vae_images(kvlens, ropes, ic.images("vae"), ident, NEW_TOKEN_IDS)
        # encoder moments per padded image, NHWC -> NCHW to compare with the reference's hook capture
        ref_m = g["vae.moments"]
        for b in range(n):
            mom = vae.encode_moments(gi["padded_images"][b:b + 1])
            h, w = ref_m.shape[2], ref_m.shape[3]
            close(mom.float().view(h, w, -1).permute(2, 0, 1), ref_m[b], f"vae.moments[{b}]", g, "vae.moments", case)
        past = model.forward_cache_update_vae(vae, past, latent_noise=g["vae.noise"].to(BF16), **gi)
        _snap_check("after_vae", past, g, L, case)
    if ic.vit_images is not None:
        gi, kvlens, ropes = model.prepare_vit_images(kvlens, ropes, ic.images("vit"), ident, NEW_TOKEN_IDS)
        tower = model.vit_tower.tower(gi["packed_vit_tokens"], gi["packed_vit_position_ids"], gi["vit_token_seqlens"].tolist())
        close(tower, g["vit.tower"], "vit.tower", g, "vit.tower", case)
        past = model.forward_cache_update_vit(past, **gi)
        _snap_check("after_vit", past, g, L, case)
    assert kvlens == g["kv_lens"].tolist() and ropes == g["ropes"].tolist()
    if ic.edit_prompt is None:
        return
    # the editing path: three CFG contexts as interleave_inference builds them (inferencer.py:281-292)
    text_ctx = (copy.deepcopy(past), list(kvlens), list(ropes))
    prompts = [" ".join(map(str, p)) for p in ic.prompts("edit")]
    gi, kvlens, ropes = model.prepare_prompts(kvlens, ropes, prompts, tok, NEW_TOKEN_IDS)
    past = model.forward_cache_update_text(past, **gi)
    gi2, ikv, irope = model.prepare_prompts([0] * n, [0] * n, prompts, tok, NEW_TOKEN_IDS)
    img_past = model.forward_cache_update_text(NaiveCache(L), **gi2)
    torch.manual_seed(ic.noise_seed + 1)
    sizes = [ic.out_image] * n
    gg = model.prepare_vae_latent(kvlens, ropes, sizes, NEW_TOKEN_IDS)
    gt = model.prepare_vae_latent_cfg(text_ctx[1], text_ctx[2], sizes)
    gm = model.prepare_vae_latent_cfg(ikv, irope, sizes)
    assert torch.equal(gg["packed_init_noises"], g["edit.init_noise"])
    nsteps = ic.num_timesteps - 1
    trace = []
    model.generate_image(
        past_key_values=past, num_timesteps=ic.num_timesteps, timestep_shift=3.0, cfg_renorm_min=0.0,
        cfg_renorm_type="text_channel", cfg_interval=[0.0, 1.0], cfg_text_scale=4.0,
        cfg_text_past_key_values=text_ctx[0], cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
        cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
        cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_scale=2.0, cfg_img_past_key_values=img_past, cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
        cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
        cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], trace=trace,
        teacher_x=[g[f"edit.step{i}.x_t"] for i in range(nsteps)], **gg)
    for i, rec in enumerate(trace):
        for k in ("v_cond", "v_text", "v_img", "v_t"):
            key = f"edit.step{i}.{k}"
            # the combined velocity amplifies branch differences by up to (2 s_text - 1)(2 s_img - 1) = 21
            check_tensor_vs_reference(rec[k], g[key], g["exact." + key], case, key, cosine=0.999 if k != "v_t" else 0.995)


def test_layernorm_kernel():
    from bagel_amd import _lib as k
    for rows, H in ((37, 576), (130, 1152), (5, 3584)):
        g_ = torch.Generator().manual_seed(rows)
        x = (torch.randn((rows, H), generator=g_) * 2 + 0.3).to(BF16)
        w = (torch.randn(H, generator=g_) * 0.1 + 1).to(BF16)
        b = (torch.randn(H, generator=g_) * 0.1).to(BF16)
        out = torch.empty((rows, H), dtype=BF16, device=DEV)
        k.layernorm(x.to(DEV), out, w.to(DEV), b.to(DEV), 1e-6)
        ref = torch.nn.functional.layer_norm(x.float(), (H,), w.float(), b.float(), 1e-6).to(BF16)
        d = (out.cpu().float() - ref.float()).abs()
        assert float(d.max()) <= 2 * 2.0 ** -8 * float(ref.float().abs().max()), (rows, H, float(d.max()))
        assert float((out.cpu() != ref).float().mean()) < 0.02


def test_gemm_gelu_epilogue_and_ragged_n():
    from bagel_amd import _lib as k
    g_ = torch.Generator().manual_seed(3)
    M, N, Kd = 300, 272, 576
    A = torch.randn((M, Kd), generator=g_).to(BF16)
    W = (torch.randn((N, Kd), generator=g_) * Kd ** -0.5).to(BF16)
    b = (torch.randn(N, generator=g_) * 0.1).to(BF16)
    C = torch.zeros((M, 320), dtype=BF16, device=DEV)
    k.gemm(A.to(DEV), W.to(DEV), C, bias=b.to(DEV), epilogue=k.EPI_BIAS_GELU)
    pre = (A.float() @ W.float().t() + b.float()).to(BF16)
    ref = torch.nn.functional.gelu(pre.float(), approximate="tanh").to(BF16)
    got = C.cpu()
    assert float(got[:, N:].abs().max()) == 0.0          # columns past N are left alone
    d = (got[:, :N].float() - ref.float()).abs()
    assert float(d.max()) <= 3 * 2.0 ** -8 * max(1.0, float(ref.float().abs().max())), float(d.max())


def test_split_heads_pad_and_scaled_attention():
    """72-wide heads, zero-padded to 128 by the split kernel, through the attention kernel with scale 1/sqrt(72)."""
    from bagel_amd import _lib as k
    from oracle import bagel_oracle as O
    g_ = torch.Generator().manual_seed(9)
    heads, Din, D = 4, 72, 128
    lens = [70, 200]
    M = sum(lens)
    qkv = torch.randn((M, 3 * heads * Din), generator=g_).to(BF16)
    starts, slots, tiles = [], [], []
    s = r = 0
    for n in lens:
        starts.append(s)
        slots += list(range(s, s + n))
        for q0 in range(0, n, 128):
            tiles.append((r + q0, min(128, n - q0), s, n, -1, 0, 0, 0))
        s += (n + 63) // 64 * 64
        r += n
    S = s
    Q = torch.empty((M, heads * D), dtype=BF16, device=DEV)
    Kw = torch.zeros((heads, S, D), dtype=BF16, device=DEV)
    Vt = torch.zeros((heads, D, S), dtype=BF16, device=DEV)
    Oo = torch.empty((M, heads * D), dtype=BF16, device=DEV)
    k.split_heads_pad(qkv.to(DEV), M, heads, Din, D, torch.tensor(slots, dtype=torch.int32, device=DEV), Q, Kw, Vt, S)
    q3 = qkv.view(M, 3, heads, Din)
    assert torch.equal(Q.cpu().view(M, heads, D)[:, :, :Din], q3[:, 0]) and float(Q.cpu().view(M, heads, D)[:, :, Din:].abs().max()) == 0
    k.varlen_attention(Q, Kw, Vt, Oo, torch.tensor(tiles, dtype=torch.int32, device=DEV), len(tiles), heads, heads, D, S,
                       softmax_scale=Din ** -0.5)
    ref = O.varlen_attention(q3[:, 0], q3[:, 1], q3[:, 2], lens, lens, causal=False)
    got = Oo.cpu().view(M, heads, D)
    assert float(got[:, :, Din:].abs().max()) == 0.0
    ma, cos = metrics(got[:, :, :Din], ref)
    assert ma <= 2e-2 and cos >= 0.9999, (ma, cos)


def test_conv3x3_stride2_matches_torch():
    from bagel_amd import _lib as k
    g_ = torch.Generator().manual_seed(4)
    Cin, Cout, H, W = 128, 128, 12, 20
    x = torch.randn((1, Cin, H, W), generator=g_).to(BF16)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g_) * (9 * Cin) ** -0.5).to(BF16)
    b = (torch.randn(Cout, generator=g_) * 0.1).to(BF16)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
    xn = x[0].permute(1, 2, 0).reshape(H * W, Cin).contiguous().to(DEV)
    wn = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    out = torch.empty((H * W // 4, Cout), dtype=BF16, device=DEV)
    k.conv3x3(xn, H, W, Cin, wn, b.to(DEV), out, torch.zeros(256, dtype=torch.uint8, device=DEV), downsample=True)
    got = out.cpu().float().view(H // 2, W // 2, Cout).permute(2, 0, 1)
    d = (got - ref[0]).abs()
    assert float(d.max()) <= 3 * 2.0 ** -8 * float(ref.abs().max()), float(d.max())


def test_vae_encode_sampling_is_seeded_by_noise():
    """encode = moments + the caller's noise; two calls with the same noise agree bit for bit, the default draws
    its own."""
    ic = IMAGE_CTX_CASES["ctx_vae_tiny"]
    _, vae = _build(ic)
    x = ic.images("vae")[1][None]
    h, w = x.shape[2] // 8, x.shape[3] // 8
    noise = torch.randn((1, 16, h, w), generator=torch.Generator().manual_seed(1)).to(BF16)
    z1, z2 = vae.encode(x, noise=noise), vae.encode(x, noise=noise)
    assert z1.shape == (1, 16, h, w) and z1.dtype == BF16 and torch.equal(z1, z2)
    z3 = vae.encode(x)
    assert not torch.equal(z1, z3)
    # against the same arithmetic in torch on the kernel's own moments
    mom = vae.encode_moments(x).float().cpu().view(h, w, 32).permute(2, 0, 1)
    mean, logvar = mom[:16].to(BF16), mom[16:].to(BF16)
    ref = 0.3611 * ((mean + torch.exp(0.5 * logvar) * noise[0]) - 0.1159)
    assert float((z1[0].cpu().float() - ref.float()).abs().max()) <= 2.0 ** -7 * float(ref.float().abs().max())
