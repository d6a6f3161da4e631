"""Size-independent properties at the benchmark's full size (BAGEL-7B dims, 28 layers, 1024x1024, three CFG
branches) - the shapes the CPU checker cannot finish in seconds.  Random-init weights generated on the device."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
IDS = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)


@pytest.fixture(scope="module")
def full():
    from bench import synthetic_context
    from bagel_amd.config import BAGEL_7B
    from bagel_amd.engine import MoTWeights
    from bagel_amd.modeling import Bagel
    dims = BAGEL_7B
    model = Bagel(dims, device=DEV, weights=MoTWeights.random(dims, torch.device(DEV), seed=0, vocab_rows=64))
    ctx = 34
    cond = synthetic_context(dims, 1, ctx, torch.device(DEV), 1)
    empty = synthetic_context(dims, 1, 0, torch.device(DEV), 2)
    yield model, dims, ctx, cond, empty
    del model
    torch.cuda.empty_cache()


def _flow(model, ctx, cond, empty, x_t, B=1, ts=4.0, is_=1.5, renorm="channel", res=(1024, 1024)):
    torch.manual_seed(0)
    gi = model.prepare_vae_latent([ctx] * B, [ctx] * B, [res] * B, IDS)
    gt = model.prepare_vae_latent_cfg([0] * B, [0] * B, [res] * B)
    gm = model.prepare_vae_latent_cfg([ctx] * B, [ctx] * B, [res] * B)
    gi.pop("packed_init_noises")
    v = model._forward_flow(
        x_t=x_t, timestep=torch.full((x_t.shape[0],), 0.7), past_key_values=cond, cfg_renorm_min=0.0, cfg_renorm_type=renorm,
        cfg_text_scale=ts, cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
        cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
        cfg_text_past_key_values=empty, cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_scale=is_, cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
        cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
        cfg_img_past_key_values=cond, cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], **gi)
    return v.clone(), {k: (None if t is None else t.clone()) for k, t in model.last_step_velocities.items()}


def test_fullsize_step_properties(full):
    model, dims, ctx, cond, empty = full
    Nv = 4096
    x = torch.randn((Nv, dims.patch_latent_dim), generator=torch.Generator().manual_seed(3))
    v3, br3 = _flow(model, ctx, cond, empty, x)
    assert torch.isfinite(v3.float()).all() and 0.05 < float(v3.float().std()) < 50
    # determinism: the same call twice is bit-identical
    v3b, br3b = _flow(model, ctx, cond, empty, x)
    assert torch.equal(v3, v3b) and all(torch.equal(br3[k], br3b[k]) for k in br3)
    # packing: each branch of the packed three-branch forward equals that branch run alone (a branch's rows never
    # read another branch's activations, and a row's GEMM result does not depend on how many rows share the launch)
    v1, br1 = _flow(model, ctx, cond, empty, x, ts=1.0, is_=1.0)
    assert torch.equal(br1["v_cond"], br3["v_cond"])
    assert torch.equal(v1, br1["v_cond"])            # CFG scales 1.0: the combined velocity IS the cond velocity
    # the image-uncond branch was given the cond context: identical inputs, identical outputs
    assert torch.equal(br3["v_img"], br3["v_cond"])
    # CFG algebra on the kernel's own branch outputs (channel renorm, fp32 reference of bagel.py:1466-1500)
    vc, vt_, vi = (br3[k].float() for k in ("v_cond", "v_text", "v_img"))
    vtext = vt_ + 4.0 * (vc - vt_)
    vfull = vi + 1.5 * (vtext - vi)
    scale = (vc.norm(dim=-1, keepdim=True) / (vfull.norm(dim=-1, keepdim=True) + 1e-8)).clamp(0.0, 1.0)
    ref = vfull * scale
    err = (v3.float().cpu() - ref.cpu()).abs().max()
    assert float(err) <= 0.05 * float(ref.abs().max()), float(err)


def test_fullsize_batch_invariance(full):
    """Two copies of one sample in a packed batch give, per sample, what the single sample gives (per-token renorm)."""
    model, dims, ctx, cond, empty = full
    from bagel_amd.modeling import NaiveCache
    Nv = 4096
    x = torch.randn((Nv, dims.patch_latent_dim), generator=torch.Generator().manual_seed(4))
    v1, _ = _flow(model, ctx, cond, empty, x)
    cond2 = NaiveCache(dims.num_layers)
    for l in range(dims.num_layers):
        cond2.key_cache[l] = torch.cat([cond.key_cache[l], cond.key_cache[l]])
        cond2.value_cache[l] = torch.cat([cond.value_cache[l], cond.value_cache[l]])
    v2, _ = _flow(model, ctx, cond2, empty, torch.cat([x, x]), B=2)
    assert torch.equal(v2[:Nv], v1) and torch.equal(v2[Nv:], v1)


def test_fullsize_ragged_resolution(full):
    """704 x 1264 -> 44 x 79 = 3476 latent tokens: a row count that is a multiple of neither the GEMM tile nor the
    attention tile.  Same packing and determinism properties, plus global renorm."""
    model, dims, ctx, cond, empty = full
    res = (704, 1264)
    Nv = (res[0] // 16) * (res[1] // 16)
    x = torch.randn((Nv, dims.patch_latent_dim), generator=torch.Generator().manual_seed(5))
    v3, br3 = _flow(model, ctx, cond, empty, x, renorm="global", res=res)
    v3b, _ = _flow(model, ctx, cond, empty, x, renorm="global", res=res)
    assert v3.shape == (Nv, dims.patch_latent_dim) and torch.isfinite(v3.float()).all() and torch.equal(v3, v3b)
    v1, br1 = _flow(model, ctx, cond, empty, x, ts=1.0, is_=1.0, res=res)
    assert torch.equal(br1["v_cond"], br3["v_cond"]) and torch.equal(br3["v_img"], br3["v_cond"])
    vc, vt_, vi = (br3[k].float() for k in ("v_cond", "v_text", "v_img"))
    vtext = vt_ + 4.0 * (vc - vt_)
    vfull = vi + 1.5 * (vtext - vi)
    ref = vfull * (vc.norm() / (vfull.norm() + 1e-8)).clamp(0.0, 1.0)
    assert float((v3.float() - ref).abs().max()) <= 0.05 * float(ref.abs().max())


def _flow3(model, x_t, caches, lens, ts=4.0, is_=1.5, renorm="global", res=(1024, 1024)):
    """One flow step with a different context (cache, length) for each of cond / text-uncond / img-uncond."""
    torch.manual_seed(0)
    (c_cond, c_text, c_img), (n_cond, n_text, n_img) = caches, lens
    gi = model.prepare_vae_latent([n_cond], [n_cond], [res], IDS)
    gt = model.prepare_vae_latent_cfg([n_text], [n_text], [res])
    gm = model.prepare_vae_latent_cfg([n_img], [n_img], [res])
    gi.pop("packed_init_noises")
    v = model._forward_flow(
        x_t=x_t, timestep=torch.full((x_t.shape[0],), 0.7), past_key_values=c_cond, cfg_renorm_min=0.0,
        cfg_renorm_type=renorm, cfg_text_scale=ts, cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
        cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
        cfg_text_past_key_values=c_text, cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_scale=is_, cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
        cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
        cfg_img_past_key_values=c_img, cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], **gi)
    return v.clone(), {k: (None if t is None else t.clone()) for k, t in model.last_step_velocities.items()}


def test_fullsize_edit_contexts(full):
    """The editing shape: three branches with contexts of 9034 / 9000 / 34 tokens (13 k keys per query tile).
    Determinism; every branch equals itself run with the other two contexts swapped for short ones (packing);
    and non-causal attention does not care about the ORDER of the context tokens: the same context with its
    tokens permuted gives the same velocity up to accumulation order (cosine >= 0.999, the parity bar)."""
    from bench import synthetic_context
    from bagel_amd.modeling import NaiveCache
    model, dims, ctx, cond34, empty = full
    dev = torch.device(DEV)
    long_a = synthetic_context(dims, 1, 9034, dev, 11)
    long_b = synthetic_context(dims, 1, 9000, dev, 12)
    Nv = 4096
    x = torch.randn((Nv, dims.patch_latent_dim), generator=torch.Generator().manual_seed(6))
    v, br = _flow3(model, x, (long_a, long_b, cond34), (9034, 9000, 34))
    assert torch.isfinite(v.float()).all() and 0.05 < float(v.float().std()) < 50
    v_again, br_again = _flow3(model, x, (long_a, long_b, cond34), (9034, 9000, 34))
    assert torch.equal(v, v_again) and all(torch.equal(br[k], br_again[k]) for k in br)
    # packing: the cond branch (9034 keys) does not depend on what the other two branches attend to
    _, br_short = _flow3(model, x, (long_a, empty, cond34), (9034, 0, 34))
    assert torch.equal(br_short["v_cond"], br["v_cond"]) and torch.equal(br_short["v_img"], br["v_img"])
    # key order: permute the context tokens of the cond branch (K and V rows together)
    perm = torch.randperm(9034, generator=torch.Generator().manual_seed(7)).to(dev)
    shuffled = NaiveCache(dims.num_layers)
    for l in range(dims.num_layers):
        shuffled.key_cache[l] = long_a.key_cache[l][perm].contiguous()
        shuffled.value_cache[l] = long_a.value_cache[l][perm].contiguous()
    _, br_perm = _flow3(model, x, (shuffled, long_b, cond34), (9034, 9000, 34))
    a, b = br["v_cond"].float().flatten(), br_perm["v_cond"].float().flatten()
    cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
    assert cos >= 0.999, cos
    assert torch.equal(br_perm["v_text"], br["v_text"])   # the other branches never saw the permutation
