"""Host-side index work (bagel_amd/packing.py) against the tensors the reference's own prepare_* methods
returned, captured by tests/golden/make_golden.py: bit-for-bit (values and dtypes), on the CPU.

Reference: bagel.py:806-841 (prepare_prompts), 876-936 (prepare_vit_images), 1010-1081 (prepare_vae_images),
1145-1201 (prepare_vae_latent), 1203-1234 (prepare_vae_latent_cfg), 1502-1523 (prepare_start_tokens).
"""
import pytest
import torch

from bagel_amd import packing
from tests.helpers import have_golden, load_golden
from tests.scenarios import IMAGE_CTX_CASES, NEW_TOKEN_IDS, SCENARIOS, TAYLOR_SCENARIOS, TEXT_CASES


class _Tok:
    def encode(self, text):
        return [int(t) for t in text.split()]


def _strs(prompts):
    return [" ".join(map(str, p)) for p in prompts]


def _check(golden, tag, got, lists=None, checked=None):
    """Every integer tensor the reference returned under ``idx.<tag>.`` equals the builder's, dtype included."""
    keys = [k for k in golden if k.startswith(f"idx.{tag}.")]
    assert keys, f"golden holds no idx.{tag}.* tensors: regenerate it with tests/golden/make_golden.py"
    for k in keys:
        name = k[len(f"idx.{tag}."):]
        ref = golden[k]
        if lists is not None and name in lists:
            assert ref.tolist() == list(lists[name]), (tag, name)
        else:
            val = got[name]
            if not isinstance(val, torch.Tensor):      # patchified_vae_latent_shapes: list of (h, w)
                assert ref.tolist() == [list(e) for e in val], (tag, name)
            else:
                assert val.dtype == ref.dtype, (tag, name, val.dtype, ref.dtype)
                assert torch.equal(val.cpu(), ref), (tag, name)
        if checked is not None:
            checked.add(k)


@pytest.mark.parametrize("name", list(SCENARIOS) + list(TAYLOR_SCENARIOS))
def test_generation_index_tensors_match_reference(name):
    if not have_golden(name):
        pytest.skip("no golden file")
    sc = (SCENARIOS.get(name) or TAYLOR_SCENARIOS[name])
    g = load_golden(name)
    d = sc.dims
    n = len(sc.images)
    seen = set()
    ctx = {}
    for w in ("cond", "text", "img"):
        kvlens, ropes = [0] * n, [0] * n
        if sc.prompts(w) is not None:
            gi, kvlens, ropes = packing.prepare_prompts(kvlens, ropes, _strs(sc.prompts(w)), _Tok(), NEW_TOKEN_IDS)
            _check(g, f"prompts.{w}", gi, {"newlens": kvlens, "new_rope": ropes}, seen)
        else:
            _check(g, f"prompts.{w}", {}, {"newlens": kvlens, "new_rope": ropes}, seen)
        ctx[w] = (kvlens, ropes)
    torch.manual_seed(sc.noise_seed)
    gi = packing.prepare_vae_latent(*ctx["cond"], sc.images, NEW_TOKEN_IDS, d.latent_downsample, d.max_latent_size,
                                    d.patch_latent_dim)
    _check(g, "vae_latent", gi, checked=seen)
    # the initial noise is the global CPU generator's stream, one draw per image in order
    assert torch.equal(gi["packed_init_noises"], g["init_noise"])
    _check(g, "vae_latent_cfg.text", packing.prepare_vae_latent_cfg(*ctx["text"], sc.images, d.latent_downsample), checked=seen)
    _check(g, "vae_latent_cfg.img", packing.prepare_vae_latent_cfg(*ctx["img"], sc.images, d.latent_downsample), checked=seen)
    assert seen == {k for k in g if k.startswith("idx.")}, "an index tensor of the golden file was not compared"


@pytest.mark.parametrize("name", list(TEXT_CASES))
def test_text_decode_index_tensors_match_reference(name):
    tc = TEXT_CASES[name]
    g = load_golden(name)
    n = len(tc.prompt)
    gi, kvlens, ropes = packing.prepare_prompts([0] * n, [0] * n, _strs(tc.prompts()), _Tok(), NEW_TOKEN_IDS)
    _check(g, "prompts", gi, {"newlens": kvlens, "new_rope": ropes})
    st = packing.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
    for k, v in st.items():
        assert torch.equal(v.to(torch.int64), g[f"start.{k}"]), k


@pytest.mark.parametrize("name", list(IMAGE_CTX_CASES))
def test_image_context_index_tensors_match_reference(name):
    ic = IMAGE_CTX_CASES[name]
    g = load_golden(name)
    d = ic.dims
    n = ic.batch
    ident = lambda x: x  # noqa: E731
    seen = set()
    kvlens, ropes = [0] * n, [0] * n
    if ic.first_prompt is not None:
        gi, kvlens, ropes = packing.prepare_prompts(kvlens, ropes, _strs(ic.prompts("first")), _Tok(), NEW_TOKEN_IDS)
        _check(g, "prompts.first", gi, {"newlens": kvlens, "new_rope": ropes}, seen)
    if ic.vae_images is not None:
        gi, kvlens, ropes = packing.prepare_vae_images(kvlens, ropes, ic.images("vae"), ident, NEW_TOKEN_IDS,
                                                       d.latent_downsample, d.max_latent_size)
        _check(g, "vae_images", gi, {"newlens": kvlens, "new_rope": ropes}, seen)
    if ic.vit_images is not None:
        gi, kvlens, ropes = packing.prepare_vit_images(kvlens, ropes, ic.images("vit"), ident, NEW_TOKEN_IDS,
                                                       d.vit.patch_size, d.vit.max_patches_per_side)
        _check(g, "vit_images", gi, {"newlens": kvlens, "new_rope": ropes}, seen)
    assert g["kv_lens"].tolist() == kvlens and g["ropes"].tolist() == ropes
    if ic.edit_prompt is not None:
        text_ctx = (list(kvlens), list(ropes))
        prompts = _strs(ic.prompts("edit"))
        _, kvlens, ropes = packing.prepare_prompts(kvlens, ropes, prompts, _Tok(), NEW_TOKEN_IDS)
        _, ikv, irope = packing.prepare_prompts([0] * n, [0] * n, prompts, _Tok(), NEW_TOKEN_IDS)
        torch.manual_seed(ic.noise_seed + 1)
        sizes = [ic.out_image] * n
        gi = packing.prepare_vae_latent(kvlens, ropes, sizes, NEW_TOKEN_IDS, d.latent_downsample, d.max_latent_size,
                                        d.patch_latent_dim)
        _check(g, "edit.vae_latent", gi, checked=seen)
        assert torch.equal(gi["packed_init_noises"], g["edit.init_noise"])
        _check(g, "edit.vae_latent_cfg.text", packing.prepare_vae_latent_cfg(*text_ctx, sizes, d.latent_downsample), checked=seen)
        _check(g, "edit.vae_latent_cfg.img", packing.prepare_vae_latent_cfg(ikv, irope, sizes, d.latent_downsample), checked=seen)
    assert seen == {k for k in g if k.startswith("idx.")}, "an index tensor of the golden file was not compared"


def test_merged_layout_properties():
    """Slots of a packed batch tile [0, total) exactly once, context before new tokens per sample."""
    ctx, new = packing.merged_layout([3, 0, 5], [2, 4, 1])
    flat = [s for c, n in zip(ctx, new) for s in list(c) + list(n)]
    assert flat == list(range(15))
    assert [len(c) for c in ctx] == [3, 0, 5] and [len(n) for n in new] == [2, 4, 1]
    assert packing.merged_layout([], []) == ([], [])
