"""A/B timing of one 3-branch flow step (BAGEL-7B width, reduced depth) under two settings of a development
knob, interleaved in one process (timings taken in different processes or on different boxes differ by more
than the effects being measured).

    python tools/step_ab.py persistent 0 1        # bagel_set_gemm_persistent(0) vs (1)
    python tools/step_ab.py variant -1 3          # bagel_set_gemm_variant
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bagel_amd import _lib as K  # noqa: E402
from bagel_amd.config import width7b_dims  # noqa: E402
from bagel_amd.engine import GenUnit, MoTEngine, MoTWeights  # noqa: E402


def main():
    knob, settings = sys.argv[1], [int(v) for v in sys.argv[2:]]
    setter = getattr(K.lib(), {"persistent": "bagel_set_gemm_persistent", "variant": "bagel_set_gemm_variant"}[knob])
    layers = int(os.environ.get("LAYERS", "8"))
    dims = width7b_dims(layers)
    dev = torch.device("cuda:0")
    eng = MoTEngine(dims, MoTWeights.random(dims, dev, vocab_rows=16), dev)
    nv = 4096
    units = [GenUnit(sample=0, branch=b, ctx_len=0, rope_pos=0) for b in range(3)]
    plan = eng.build_plan(units, [nv], [0])
    eng.load_context(plan)
    eng.set_latent_position_ids(torch.arange(nv) % (dims.max_latent_size ** 2))
    x = torch.randn((nv, dims.patch_latent_dim), device=dev, dtype=torch.bfloat16)
    temb = torch.randn((dims.hidden_size,), device=dev, dtype=torch.bfloat16) * 0.02
    text = torch.randn((plan.Mt, dims.hidden_size), device=dev, dtype=torch.bfloat16) * 0.02
    times = {s: [] for s in settings}
    for r in range(int(os.environ.get("ROUNDS", "8")) + 1):
        for s in settings:
            setter(s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                eng.forward_gen(plan, x, temb, text)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[s].append(e0.elapsed_time(e1) / 2)
    for s in settings:
        t = sorted(times[s])
        print(f"{knob}={s}: median {t[len(t) // 2]:.3f} ms per step ({layers} layers), min {t[0]:.3f}")
    setter(-1 if knob == "variant" else 1)


if __name__ == "__main__":
    main()
