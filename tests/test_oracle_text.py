"""The CPU checker's greedy text decoding against the reference's golden tokens and logits."""
import pytest
import torch

from tests.helpers import O, load_golden, metrics, oracle_context
from tests.scenarios import TEXT_CASES, NEW_TOKEN_IDS
from bagel_amd.weights import make_llm_state_dict


@pytest.mark.parametrize("name", list(TEXT_CASES))
def test_oracle_generate_text_matches_reference(name):
    tc = TEXT_CASES[name]
    g = load_golden(name)
    sd = make_llm_state_dict(tc.dims, seed=tc.weight_seed, include_lm_head=True)
    with torch.no_grad():
        past, kvlens, ropes = oracle_context(sd, tc.dims, tc.prompts(), len(tc.prompt))
        gi = O.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
        for k, v in gi.items():
            assert torch.equal(v.to(torch.int64), g[f"start.{k}"]), k
        trace = []
        tokens = O.generate_text(sd, tc.dims, past, max_length=tc.max_length, logits_trace=trace, **gi)
    assert torch.equal(tokens.to(torch.int64), g["tokens"])
    logits = torch.stack(trace).float()
    # same rounding points; the attention inner loop is flash-attention arithmetic here and SDPA in the
    # capture, so logits may differ by a bf16 ulp or two
    ma, cos = metrics(logits, g["logits"])
    assert ma <= max(2e-2, 2.0 ** -6 * float(g["logits"].abs().max())) and cos >= 0.9999, (ma, cos)


def test_oracle_generate_text_stops_at_end_token():
    tc = TEXT_CASES["text_tiny_b1"]
    g = load_golden(tc.name)
    sd = make_llm_state_dict(tc.dims, seed=tc.weight_seed, include_lm_head=True)
    end = int(g["tokens"][4, 0])           # the token produced by step 3 and fed at step 4
    first = int((g["tokens"][1:, 0] == end).nonzero()[0]) + 1
    with torch.no_grad():
        past, kvlens, ropes = oracle_context(sd, tc.dims, tc.prompts(), 1)
        gi = O.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
        tokens = O.generate_text(sd, tc.dims, past, max_length=tc.max_length, end_token_id=end, **gi)
    # bagel.py:1590-1594: the loop ends after the step that produced the end token, which is not appended
    assert tokens.shape[0] == first
    assert torch.equal(tokens.to(torch.int64), g["tokens"][:first])
