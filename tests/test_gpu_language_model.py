"""``model.language_model.forward_inference`` (the reference's Qwen2ForCausalLM surface) on HIP against the CPU
checker's restatement of Qwen2Model.forward_inference."""
import pytest
import torch

from tests.helpers import O, metrics
from tests.scenarios import NEW_TOKEN_IDS
from bagel_amd.config import tiny_dims
from bagel_amd.weights import make_llm_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def setup():
    from bagel_amd.modeling import Bagel
    dims = tiny_dims(3)
    sd = make_llm_state_dict(dims, seed=5, include_lm_head=True)
    return dims, sd, Bagel(dims, state_dict=sd, device=DEV)


def _context(dims, sd, lens):
    with torch.no_grad():
        gi, kvl, ropes = O.prepare_prompts([0] * len(lens), [0] * len(lens), [list(range(10 + b, 10 + b + n)) for b, n in enumerate(lens)],
                                           NEW_TOKEN_IDS)
        past = O.forward_cache_update_text(sd, dims, O.Cache(dims.num_layers), **gi)
    return past, kvl, ropes


def _layout(kvl, lens):
    """Contiguous per-sample layout [context | new tokens]."""
    qi, ki, tot = [], [], 0
    for c, n in zip(kvl, lens):
        ki += list(range(tot, tot + c))
        qi += list(range(tot + c, tot + c + n))
        tot += c + n
    return torch.tensor(qi), torch.tensor(ki)


def _to_gpu(past, L):
    from tests.gpu_common import to_gpu_cache
    return to_gpu_cache(past, L)


def _close(got, ref, what):
    ma, cos = metrics(got, ref)
    assert tuple(got.shape) == tuple(ref.shape), what
    assert ma <= 0.03 * float(ref.float().abs().max()) and cos >= 0.999, (what, ma, cos)


@pytest.mark.parametrize("is_causal", [True, False])
def test_forward_inference_und(setup, is_causal):
    dims, sd, model = setup
    L = dims.num_layers
    past, kvl, ropes = _context(dims, sd, [9, 150])
    lens = [70, 5]
    g = torch.Generator().manual_seed(1)
    seq = (torch.randn((sum(lens), dims.hidden_size), generator=g) * 0.05).to(BF16)
    pos = torch.tensor([ropes[0] + i for i in range(lens[0])] + [ropes[1] + i for i in range(lens[1])])
    qi, ki = _layout(kvl, lens)
    import copy
    with torch.no_grad():
        ref_h, ref_past = O.llm_forward_inference(sd, dims, seq, torch.tensor(lens), pos, qi, copy.deepcopy(past), torch.tensor(kvl), ki,
                                                  update_past_key_values=True, is_causal=is_causal, mode="und")
    gp = _to_gpu(past, L)
    out = model.language_model.forward_inference(
        packed_query_sequence=seq.to(DEV), query_lens=torch.tensor(lens), packed_query_position_ids=pos, packed_query_indexes=qi,
        past_key_values=gp, key_values_lens=torch.tensor(kvl), packed_key_value_indexes=ki, update_past_key_values=True,
        is_causal=is_causal, mode="und")
    _close(out.packed_query_sequence, ref_h, "hidden")
    assert out.past_key_values is gp
    for l in (0, L - 1):
        _close(gp.key_cache[l], ref_past.key_cache[l], f"k{l}")
        _close(gp.value_cache[l], ref_past.value_cache[l], f"v{l}")
    # update_past_key_values=False leaves the cache alone
    gp2 = _to_gpu(past, L)
    k0 = gp2.key_cache[0].clone()
    out2 = model.language_model.forward_inference(seq.to(DEV), torch.tensor(lens), pos, qi, gp2, torch.tensor(kvl), ki,
                                                  update_past_key_values=False, is_causal=is_causal, mode="und")
    assert torch.equal(gp2.key_cache[0], k0)
    _close(out2.packed_query_sequence, ref_h, "hidden (no update)")


def test_forward_inference_gen_mixed_rows(setup):
    """mode="gen": rows split between the experts by index lists, [text, vae..., text] per sample, with and
    without context, non-causal."""
    dims, sd, model = setup
    L = dims.num_layers
    past, kvl, ropes = _context(dims, sd, [21, 4])
    lens = [2 + 130, 2 + 40]
    g = torch.Generator().manual_seed(2)
    seq = (torch.randn((sum(lens), dims.hidden_size), generator=g) * 0.05).to(BF16)
    text_idx = torch.tensor([0, lens[0] - 1, lens[0], sum(lens) - 1])
    vae_idx = torch.tensor([i for i in range(sum(lens)) if i not in set(text_idx.tolist())])
    pos = torch.tensor([ropes[0]] * lens[0] + [ropes[1]] * lens[1])
    qi, ki = _layout(kvl, lens)
    import copy
    with torch.no_grad():
        ref_h, ref_past = O.llm_forward_inference(sd, dims, seq, torch.tensor(lens), pos, qi, copy.deepcopy(past), torch.tensor(kvl), ki,
                                                  update_past_key_values=True, is_causal=False, mode="gen",
                                                  packed_vae_token_indexes=vae_idx, packed_text_indexes=text_idx)
    gp = _to_gpu(past, L)
    out = model.language_model.forward_inference(seq.to(DEV), torch.tensor(lens), pos, qi, gp, torch.tensor(kvl), ki,
                                                 update_past_key_values=True, is_causal=False, mode="gen",
                                                 packed_vae_token_indexes=vae_idx, packed_text_indexes=text_idx)
    _close(out.packed_query_sequence, ref_h, "hidden")
    for l in (0, L - 1):
        _close(gp.key_cache[l], ref_past.key_cache[l], f"k{l}")
        _close(gp.value_cache[l], ref_past.value_cache[l], f"v{l}")
    with pytest.raises(AssertionError):
        model.language_model.forward_inference(seq.to(DEV), torch.tensor(lens), pos, qi, gp, torch.tensor(kvl), ki, mode="gen")
    with pytest.raises(NotImplementedError):
        model.language_model.forward_inference(seq.to(DEV), torch.tensor(lens), pos, qi, _to_gpu(past, L), torch.tensor(kvl), ki,
                                               is_causal=True, mode="gen", packed_vae_token_indexes=vae_idx,
                                               packed_text_indexes=text_idx)


def test_embed_tokens_and_lm_head(setup):
    dims, sd, model = setup
    ids = torch.tensor([[1, 5, 300], [2, 2, 511]])
    emb = model.language_model.model.embed_tokens(ids.to(DEV))
    assert emb.shape == (2, 3, dims.hidden_size)
    assert torch.equal(emb.cpu(), sd["language_model.model.embed_tokens.weight"][ids])
    for rows in (3, 100):
        x = (torch.randn((rows, dims.hidden_size), generator=torch.Generator().manual_seed(rows)) * 0.5).to(BF16)
        logits = model.language_model.lm_head(x.to(DEV))
        ref = torch.nn.functional.linear(x.float(), sd["language_model.lm_head.weight"].float()).to(BF16)
        assert logits.shape == (rows, dims.vocab_size)
        ma, cos = metrics(logits, ref)
        assert ma <= 2.0 ** -7 * float(ref.float().abs().max()) and cos >= 0.9999, (rows, ma, cos)
