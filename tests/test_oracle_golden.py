"""Pin the CPU checker (oracle/) against outputs of the reference itself.

The golden files were produced by tests/golden/make_golden.py, which runs the
reference's own ``Bagel.generate_image`` on the same seeded weights.  Each step is
teacher-forced from the golden ``x_t`` so that errors do not compound.
Tolerance for velocities, per CFG branch: cosine >= 0.999 and max-abs <= 2e-2, the latter
widened to 1.5x the reference's own distance to exact arithmetic where bf16 rounding makes
2e-2 unreachable for any implementation (see tests/helpers.py:check_velocity).
"""
import pytest
import torch

from tests.helpers import OracleRun, load_golden, have_golden, metrics, check_velocity, V_MAX_ABS, V_COSINE
from tests.scenarios import SCENARIOS

FAST = [n for n, s in SCENARIOS.items() if not s.slow]
SLOW = [n for n, s in SCENARIOS.items() if s.slow]


def _check_scenario(name):
    if not have_golden(name):
        pytest.skip(f"golden file for {name} not present")
    sc = SCENARIOS[name]
    g = load_golden(name)
    run = OracleRun(sc)
    assert torch.equal(run.gi["packed_init_noises"], g["init_noise"]), "noise draw order differs from the reference"
    if "cond_ctx.k0" in g:
        past = run.ctx["cond"][0]
        last = sc.dims.num_layers - 1
        for key, got in (("k0", past.key_cache[0]), ("v0", past.value_cache[0]),
                         ("k_last", past.key_cache[last]), ("v_last", past.value_cache[last])):
            ma, cos = metrics(got, g["cond_ctx." + key])
            # K/V are O(1..5) bf16 values; a few ulp (2^-8 relative each) of drift over the layers
            assert ma <= 0.03 * float(g["cond_ctx." + key].float().abs().max()) and cos >= V_COSINE, (key, ma, cos)
    n_steps = int(g["__meta__"]["n_steps"])
    worst = 0.0
    for i in range(n_steps):
        v_t, (v_c, v_x, v_i) = run.step(i, g[f"step{i}.x_t"])
        for key, got in (("v_cond", v_c), ("v_text", v_x), ("v_img", v_i)):
            gk = f"step{i}.{key}"
            assert (got is None) == (gk not in g), f"branch {key} presence differs at step {i}"
            if got is not None:
                ma, cos, _ = check_velocity(got, g, gk, name)
                worst = max(worst, ma)
        check_velocity(v_t, g, f"step{i}.v_t", name)
    print(f"{name}: worst branch max-abs {worst:.4f}")


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference(name):
    _check_scenario(name)


@pytest.mark.slow
@pytest.mark.parametrize("name", SLOW)
def test_oracle_matches_reference_slow(name):
    _check_scenario(name)


def test_oracle_free_running_final_latent():
    """Free-running sampler (no teacher forcing) lands near the reference's final latent."""
    name = "tiny_t2i"
    sc = SCENARIOS[name]
    g = load_golden(name)
    out = OracleRun(sc).generate()
    ma, cos = metrics(torch.cat(list(out)), g["final_latent"])
    assert cos >= 0.999 and ma <= 0.1, (ma, cos)


@pytest.mark.parametrize("name", ["tiny_taylor", "tiny_taylor_late"])
def test_oracle_taylorseer_matches_reference(name):
    """generate_image with enable_taylorseer=True (a branch's first five steps full, then one in three; skipped steps
    return the Taylor expansion of the last layer's cached output), teacher-forced from the reference's per-step
    inputs.  ``tiny_taylor_late``: the CFG branches start at step 4 and keep their own schedules."""
    from tests.scenarios import TAYLOR_SCENARIOS
    sc = TAYLOR_SCENARIOS[name]
    g = load_golden(sc.name)
    n = sc.num_timesteps - 1
    trace = []
    with torch.no_grad():
        OracleRun(sc).generate(trace=trace, teacher_x=[g[f"step{i}.x_t"] for i in range(n)])
    assert len(trace) == n
    for i, rec in enumerate(trace):
        for key in ("v_cond", "v_text", "v_img"):
            gk = f"step{i}.{key}"
            assert (rec[key] is None) == (gk not in g), (i, key)
            if rec[key] is not None:
                ma, cos = metrics(rec[key], g[gk])
                assert ma <= V_MAX_ABS and cos >= V_COSINE, (i, key, ma, cos)
