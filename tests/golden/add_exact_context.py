"""Add the exact-arithmetic context keys/values (``cond_ctx.exact.*``) to LLM golden files that lack them.

    python tests/golden/add_exact_context.py [scenario ...]

make_golden.py writes them for new captures; this script brings older files up to date without re-running the
reference: the reference's tensors in the file are left as they are, the exact-arithmetic evaluation (fp32, no
rounding points) of the same prefill by the CPU checker is added beside them."""
import os
import sys

import torch
from safetensors import safe_open
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.helpers import OracleRun, scenario_state_dict  # noqa: E402
from tests.scenarios import SCENARIOS, IMAGE_CTX_CASES  # noqa: E402


def exact_context_tensors(sc, sd=None, lazy=False):
    run = OracleRun(sc, sd=sd, exact=True, lazy=lazy)
    pk = run.ctx["cond"][0]
    if pk.key_cache[0] is None:
        return {}
    last = sc.dims.num_layers - 1
    return {"cond_ctx.exact.k0": pk.key_cache[0].float().clone(), "cond_ctx.exact.v0": pk.value_cache[0].float().clone(),
            "cond_ctx.exact.k_last": pk.key_cache[last].float().clone(),
            "cond_ctx.exact.v_last": pk.value_cache[last].float().clone()}


def upgrade_image_ctx(name):
    from tests.test_oracle_image_ctx import exact_twins
    path = os.path.join(HERE, name + ".safetensors")
    with safe_open(path, framework="pt") as f:
        tensors = {k: f.get_tensor(k) for k in f.keys()}
        meta = f.metadata()
    if any(k.startswith("exact.") for k in tensors):
        return
    with torch.no_grad():
        twins = exact_twins(IMAGE_CTX_CASES[name], tensors)
    for k, v in twins.items():
        assert v.shape == tensors[k[len("exact."):]].shape, k
    tensors.update(twins)
    save_file({k: v.contiguous() for k, v in tensors.items()}, path, metadata=meta)
    worst = max(float((tensors[k[len("exact."):]].float() - v).abs().max()) for k, v in twins.items())
    print(f"[{name}] added {len(twins)} exact tensors; largest reference-vs-exact distance {worst:.4f}", flush=True)


if __name__ == "__main__":
    for name in sys.argv[1:] or list(SCENARIOS) + list(IMAGE_CTX_CASES):
        path = os.path.join(HERE, name + ".safetensors")
        if not os.path.exists(path):
            continue
        if name in IMAGE_CTX_CASES:
            upgrade_image_ctx(name)
            continue
        with safe_open(path, framework="pt") as f:
            tensors = {k: f.get_tensor(k) for k in f.keys()}
            meta = f.metadata()
        if "cond_ctx.k0" not in tensors or "cond_ctx.exact.k0" in tensors:
            continue
        sc = SCENARIOS[name]
        big = sc.dims.num_layers * sc.dims.hidden_size * sc.dims.intermediate_size > 1e9
        with torch.no_grad():
            extra = exact_context_tensors(sc, sd=scenario_state_dict(sc.dims, sc.weight_seed) if big else None, lazy=big)
        for k, v in extra.items():
            assert v.shape == tensors[k.replace(".exact", "")].shape
        tensors.update(extra)
        save_file({k: v.contiguous() for k, v in tensors.items()}, path, metadata=meta)
        d = {k: float((tensors[k.replace('.exact', '')].float() - v).abs().max()) for k, v in extra.items()}
        print(f"[{name}] added {sorted(extra)}; reference-vs-exact max {d}", flush=True)
