"""Greedy text decoding on the HIP path (Bagel.generate_text) against the reference's golden tokens/logits."""
import pytest
import torch

from tests.helpers import load_golden, metrics
from tests.scenarios import TEXT_CASES, NEW_TOKEN_IDS

pytestmark = pytest.mark.gpu


class _Tok:
    def encode(self, text):
        return [int(t) for t in text.split()]

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def _model(tc):
    from bagel_amd.modeling import Bagel
    from bagel_amd.weights import make_llm_state_dict
    sd = make_llm_state_dict(tc.dims, seed=tc.weight_seed, include_lm_head=True)
    return Bagel(tc.dims, state_dict=sd, device="cuda:0")


def _context(model, tc):
    from bagel_amd.modeling import NaiveCache
    n = len(tc.prompt)
    gi, kvlens, ropes = model.prepare_prompts([0] * n, [0] * n, [" ".join(map(str, p)) for p in tc.prompts()], _Tok(),
                                              NEW_TOKEN_IDS)
    past = model.forward_cache_update_text(NaiveCache(tc.dims.num_layers), **gi)
    return past, kvlens, ropes


def _logit_bound(ref):
    return max(2e-2, 2.0 ** -6 * float(ref.abs().max()))


@pytest.mark.parametrize("name", list(TEXT_CASES))
def test_generate_text_teacher_forced_logits(name):
    """Feed the reference's tokens; every step's logits must match the reference's."""
    tc = TEXT_CASES[name]
    g = load_golden(name)
    model = _model(tc)
    past, kvlens, ropes = _context(model, tc)
    gi = model.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
    for k, v in gi.items():
        assert torch.equal(v.cpu().to(torch.int64), g[f"start.{k}"]), k
    ref_tokens = g["tokens"].to("cuda:0")
    step = [0]

    def forced(logits):
        step[0] += 1
        return ref_tokens[min(step[0], ref_tokens.shape[0] - 1)]

    trace = []
    model.engine.decode_tokens(gi["packed_start_tokens"].tolist(), gi["packed_query_position_ids"].tolist(),
                               kvlens, *_split_ctx(model, past, kvlens), tc.max_length, sample_fn=forced,
                               logits_trace=trace)
    logits = torch.stack(trace).float().cpu()
    ref = g["logits"]
    assert logits.shape == ref.shape
    ma, cos = metrics(logits, ref)
    assert ma <= _logit_bound(ref) and cos >= 0.9999, (ma, cos)
    # greedy choice on these logits: identical wherever the reference's top-2 margin exceeds the bound
    top2 = ref.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * _logit_bound(ref)
    assert torch.equal(logits.argmax(-1)[clear], ref.argmax(-1)[clear])


def test_random_decode_batches_match_cpu_checker():
    """Random sample counts (1 ... 24: the GEMV, weight-streaming and tile forms of a decode step), prompt lengths and
    depths: the CPU checker decodes greedily, the HIP path is fed the checker's tokens and must give its logits.
    BAGEL_FUZZ_DECODE=n runs n cases (default 5)."""
    import os
    import random
    from bagel_amd.config import tiny_dims
    from bagel_amd.weights import make_llm_state_dict
    from tests.helpers import O, oracle_context
    from tests.scenarios import TextCase
    rnd = random.Random(77)
    n_cases = int(os.environ.get("BAGEL_FUZZ_DECODE", 5))
    for case in range(n_cases):
        if case and case % 25 == 0:
            print(f"  random decode batches: {case} / {n_cases}", flush=True)
        n = rnd.choice([1, 2, 3, 4, 5, 7, 12, 16, 17, 24])
        tc = TextCase(f"random{case}", tiny_dims(rnd.randint(1, 3)), [rnd.choice([1, 3, 30, 62, 63, 64, 65, 140]) for _ in range(n)],
                      rnd.randint(3, 9), weight_seed=case % 4)
        what = f"case {case}: prompts {tc.prompt} L={tc.dims.num_layers} max_length {tc.max_length}"
        model = _model(tc)
        sd = make_llm_state_dict(tc.dims, seed=tc.weight_seed, include_lm_head=True)
        with torch.no_grad():
            opast, okv, oropes = oracle_context(sd, tc.dims, tc.prompts(), n)
            ogi = O.prepare_start_tokens(okv, oropes, NEW_TOKEN_IDS)
            otrace = []
            otokens = O.generate_text(sd, tc.dims, opast, max_length=tc.max_length, logits_trace=otrace, **ogi)
        past, kvlens, ropes = _context(model, tc)
        gi = model.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
        ref_tokens = otokens.to("cuda:0")
        step = [0]

        def forced(logits):
            step[0] += 1
            return ref_tokens[min(step[0], ref_tokens.shape[0] - 1)]

        trace = []
        model.engine.decode_tokens(gi["packed_start_tokens"].tolist(), gi["packed_query_position_ids"].tolist(), kvlens,
                                   *_split_ctx(model, past, kvlens), tc.max_length, sample_fn=forced, logits_trace=trace)
        logits, ref = torch.stack(trace).float().cpu(), torch.stack(otrace).float()
        assert logits.shape == ref.shape, what
        ma, cos = metrics(logits, ref)
        assert ma <= _logit_bound(ref) and cos >= 0.9999, (what, ma, cos)
    print(f"{n_cases} random decode batches agree with the CPU checker")


def _split_ctx(model, past, kvlens):
    L = model.dims.num_layers
    ks, vs, c = [], [], 0
    for n in kvlens:
        ks.append([past.key_cache[l][c:c + n] for l in range(L)])
        vs.append([past.value_cache[l][c:c + n] for l in range(L)])
        c += n
    return ks, vs


@pytest.mark.parametrize("name", list(TEXT_CASES))
def test_generate_text_free_running(name):
    """Bagel.generate_text end to end (device argmax feeds the next step).  Tokens must equal the reference's
    up to the first step whose reference top-2 margin is within rounding noise."""
    tc = TEXT_CASES[name]
    g = load_golden(name)
    model = _model(tc)
    past, kvlens, ropes = _context(model, tc)
    gi = model.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
    k0 = past.key_cache[0].clone()
    tokens = model.generate_text(past_key_values=past, max_length=tc.max_length, do_sample=False, **gi).cpu()
    assert torch.equal(past.key_cache[0], k0)          # the caller's cache is left alone
    ref = g["tokens"]
    assert tokens.shape == ref.shape and tokens.dtype == torch.long
    # identical tokens, except that a step whose two best reference logits are within rounding noise of
    # each other may legitimately pick the other one; everything after such a step is a different sequence
    for t in range(ref.shape[0] - 1):
        if torch.equal(tokens[t + 1], ref[t + 1]):
            continue
        for b in range(ref.shape[1]):
            lg = g["logits"][t, b]
            assert float(lg[ref[t + 1, b]] - lg[tokens[t + 1, b]]) <= 2 * _logit_bound(g["logits"]), (t, b, tokens.T.tolist())
        break


def test_generate_text_end_token_and_sampling():
    tc = TEXT_CASES["text_tiny_b1"]
    g = load_golden(tc.name)
    model = _model(tc)
    past, kvlens, ropes = _context(model, tc)
    gi = model.prepare_start_tokens(kvlens, ropes, NEW_TOKEN_IDS)
    end = int(g["tokens"][4, 0])
    first = int((g["tokens"][1:, 0] == end).nonzero()[0]) + 1
    tokens = model.generate_text(past_key_values=past, max_length=tc.max_length, end_token_id=end, **gi).cpu()
    assert torch.equal(tokens, g["tokens"][:first])
    # sampling: valid ids, right shape, temperature -> 0 reproduces greedy
    torch.manual_seed(0)
    s = model.generate_text(past_key_values=past, max_length=6, do_sample=True, temperature=1.0, **gi).cpu()
    assert s.shape == (6, 1) and int(s.min()) >= 0 and int(s.max()) < tc.dims.vocab_size
    s0 = model.generate_text(past_key_values=past, max_length=6, do_sample=True, temperature=1e-4, **gi).cpu()
    assert torch.equal(s0, g["tokens"][:6])


def test_inferencer_gen_text_and_think():
    """InterleaveInferencer.gen_text (inferencer.py:215-244) and the think path of interleave_inference."""
    from bagel_amd.inferencer import InterleaveInferencer
    tc = TEXT_CASES["text_tiny_b1"]
    g = load_golden(tc.name)
    model = _model(tc)

    class Tok(_Tok):
        def decode(self, ids):
            return "<|im_start|>" + " ".join(str(int(i)) for i in ids[1:]) + "<|im_end|>"

    inf = InterleaveInferencer(model, None, Tok(), None, None, NEW_TOKEN_IDS)
    ctx = inf.init_gen_context()
    ctx = inf.update_context_text(" ".join(map(str, tc.prompts()[0])), ctx)
    text = inf.gen_text(ctx, max_length=tc.max_length, do_sample=False)
    assert text == " ".join(str(int(t)) for t in g["tokens"][1:, 0])
    out = inf.interleave_inference([" ".join(map(str, tc.prompts()[0]))], understanding_output=True, do_sample=False,
                                   max_think_token_n=tc.max_length)
    assert out == [text]


def test_argmax_rows_kernel():
    """First maximum per row, like torch.argmax: ties, ragged widths, a vocabulary-sized row, a strided view."""
    from bagel_amd import _lib as k
    g_ = torch.Generator().manual_seed(0)
    for rows, n in ((3, 512), (2, 1001), (1, 152064), (5, 7)):
        x = torch.randn((rows, n), generator=g_).to(torch.bfloat16)     # bf16 rounding makes ties common
        x[0, n // 2] = x[0].max()                                        # an exact tie with an earlier or later maximum
        xd = x.to("cuda:0")
        out = torch.empty(rows, dtype=torch.int32, device="cuda:0")
        k.argmax_rows(xd, out)
        assert out.cpu().tolist() == torch.argmax(x.float(), dim=-1).tolist(), (rows, n)
    wide = torch.randn((4, 1024), generator=g_).to(torch.bfloat16).to("cuda:0")
    view = wide[:, 3:900]                                                # unaligned start, row stride 1024
    out = torch.empty(4, dtype=torch.int32, device="cuda:0")
    k.argmax_rows(view, out)
    assert out.cpu().tolist() == torch.argmax(view.float().cpu(), dim=-1).tolist()


def test_skinny_gemm_residual_rmsnorm_matches_two_step():
    """The fused projection + residual + RMSNorm launch equals skinny GEMM (residual epilogue) followed by the
    RMSNorm kernel, bit for bit (same partial sums, same rounding points)."""
    from bagel_amd import _lib as k
    g_ = torch.Generator().manual_seed(2)
    for M, N, Kd in ((1, 3584, 3584), (5, 256, 512), (3, 3584, 18944)):
        X = torch.randn((M, Kd), generator=g_).to(torch.bfloat16).to("cuda:0")
        W = (torch.randn((N, Kd), generator=g_) * Kd ** -0.5).to(torch.bfloat16).to("cuda:0")
        R = torch.randn((M, N), generator=g_).to(torch.bfloat16).to("cuda:0")
        wn = (torch.randn(N, generator=g_) * 0.1 + 1).to(torch.bfloat16).to("cuda:0")
        ws = torch.empty((16 * M * N,), dtype=torch.float32, device="cuda:0")
        C1, Xn1 = R.clone(), torch.empty_like(R)
        k.skinny_gemm(X, W, C1, ws, residual=C1, epilogue=k.SK_RESIDUAL)
        k.rmsnorm(C1, Xn1, wn, None, M, 1e-6)
        C2, Xn2 = R.clone(), torch.empty_like(R)
        k.skinny_gemm_residual_rmsnorm(X, W, C2, C2, wn, Xn2, 1e-6, ws)
        assert torch.equal(C1, C2), (M, N, Kd)
        d = (Xn1.float() - Xn2.float()).abs().max()
        assert float(d) <= 2.0 ** -7 * float(Xn1.float().abs().max()), (M, N, Kd, float(d))   # sum order of the row norm differs
