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
from tests.scenarios import SCENARIOS  # noqa: E402


def exact_context_tensors(sc, sd=None, lazy=False):
    run = OracleRun(sc, sd=sd, exact=True, lazy=lazy)
    pk = run.ctx["cond"][0]
    if pk.key_cache[0] is None:
        return {}
    last = sc.dims.num_layers - 1
    return {"cond_ctx.exact.k0": pk.key_cache[0].float().clone(), "cond_ctx.exact.v0": pk.value_cache[0].float().clone(),
            "cond_ctx.exact.k_last": pk.key_cache[last].float().clone(),
            "cond_ctx.exact.v_last": pk.value_cache[last].float().clone()}


if __name__ == "__main__":
    for name in sys.argv[1:] or list(SCENARIOS):
        path = os.path.join(HERE, name + ".safetensors")
        if not os.path.exists(path):
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
