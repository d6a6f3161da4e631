"""Scenario-level parity of the HIP path (through Bagel.generate_image and the C-ABI) against
(a) the golden vectors captured from the reference and (b) the CPU checker on the same seeded inputs.

Every Euler step is teacher-forced from the reference's x_t so that errors do not compound; a separate
test runs the sampler free.  Bounds: see tests/helpers.py:check_velocity.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import (OracleRun, load_golden, have_golden, metrics, check_velocity, scenario_state_dict,  # noqa: E402
                           parity_stats, mean_slack, EXACT_RATIO_MEAN, REF_RATIO_MAX)
from tests.scenarios import SCENARIOS  # noqa: E402
from tests.gpu_common import run_scenario_on_gpu  # noqa: E402

HD128 = list(SCENARIOS)  # head_dim 128 and 64 (the 0.5B-shaped plumbing configuration)


@pytest.mark.parametrize("name", HD128)
def test_teacher_forced_steps_match_reference(name):
    if not have_golden(name):
        pytest.skip("no golden file")
    sc = SCENARIOS[name]
    g = load_golden(name)
    run = OracleRun(sc)
    n_steps = int(g["__meta__"]["n_steps"])
    trace = []
    run_scenario_on_gpu(sc, run, trace=trace, max_steps=n_steps, teacher_x=[g[f"step{i}.x_t"] for i in range(n_steps)])
    assert len(trace) == n_steps
    worst = (0.0, 1.0)
    for i, rec in enumerate(trace):
        for key in ("v_cond", "v_text", "v_img", "v_t"):
            gk = f"step{i}.{key}"
            assert (key in rec) == (gk in g), f"step {i}: branch {key} presence differs from the reference"
            if key in rec:
                ma, cos, _ = check_velocity(rec[key], g, gk, name)
                if key != "v_t":
                    worst = (max(worst[0], ma), min(worst[1], cos))
    print(f"{name}: worst branch max-abs {worst[0]:.4f}, min cosine {worst[1]:.6f}")


@pytest.mark.parametrize("name", ["tiny_t2i", "tiny_batch2"])
def test_cfg_kernel_on_reference_branches(name):
    """Feed the reference's own branch velocities to the CFG/Euler kernel: isolates that kernel."""
    from bagel_amd import _lib as k
    sc = SCENARIOS[name]
    g = load_golden(name)
    i = 0
    vc, vx, vi = (g[f"step{i}.{b}"].to("cuda:0") for b in ("v_cond", "v_text", "v_img"))
    x = g[f"step{i}.x_t"].to("cuda:0").clone()
    xb = torch.empty_like(vc)
    vo = torch.empty_like(vc)
    ws = torch.empty(k.cfg_workspace_bytes(*x.shape), dtype=torch.uint8, device="cuda:0")
    from oracle import bagel_oracle as O
    _, dts = O.flow_timesteps(sc.num_timesteps, sc.timestep_shift)
    k.cfg_euler_update(vc, vx, vi, sc.cfg_text_scale, sc.cfg_img_scale, sc.cfg_renorm_type, sc.cfg_renorm_min,
                       float(dts[i]), x, xb, vo, ws)
    mism = (vo.cpu() != g[f"step{i}.v_t"]).float().mean()
    assert mism < 2e-3, f"{mism:.5f} of elements differ from the reference's combined velocity"
    assert float((x.cpu() - g[f"step{i + 1}.x_t"]).abs().max()) < 2e-3


def test_free_running_sampler_final_latent():
    name = "tiny_t2i"
    sc = SCENARIOS[name]
    g = load_golden(name)
    out = run_scenario_on_gpu(sc, OracleRun(sc))
    ma, cos = metrics(torch.cat(list(out)), g["final_latent"])
    assert cos >= 0.999 and ma <= 0.15, (ma, cos)


def test_free_running_sampler_24_layers_latent_and_image():
    """The plumbing configuration (24 layers, head_dim 64, 256 x 256, 10 timesteps) run FREE through all nine Euler
    steps: the final latent against the reference's, and both latents decoded by the same HIP VAE decoder against each
    other as 8-bit images.  Bar: PSNR >= 40 dB (and the latent's cosine)."""
    import numpy as np
    from bagel_amd.config import VAEDims
    from bagel_amd.modeling import AutoEncoder
    from bagel_amd.weights import make_vae_decoder_state_dict
    from oracle import bagel_oracle as O
    name = "c1_0p5b_256"
    sc = SCENARIOS[name]
    g = load_golden(name)
    assert "final_latent" in g
    out = torch.cat(list(run_scenario_on_gpu(sc, OracleRun(sc))))
    ref = g["final_latent"]
    ma, cos = metrics(out, ref)
    rel = float((out.cpu().float() - ref.float()).norm() / ref.float().norm())
    vd = VAEDims()
    ae = AutoEncoder(vd, make_vae_decoder_state_dict(vd, seed=0), "cuda:0")

    def image(latent):
        z = O.unpatchify_latent(sc.dims, latent.detach().float().cpu(), sc.images[0]).to("cuda:0")
        return O.to_uint8_image(ae.decode(z).float().cpu()).numpy().astype(np.float32)

    a, b = image(out), image(ref)
    mse = float(((a - b) ** 2).mean())
    p = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    print(f"{name} free-running: final latent max-abs {ma:.4f} cosine {cos:.6f} relative L2 {rel:.4f}; image PSNR {p:.1f} dB")
    from tests.helpers import PARITY_LOG
    PARITY_LOG.append({"golden": name, "tensor": "final_latent (9 free Euler steps)", "max_abs": ma, "cosine": cos,
                       "relative_l2": rel, "image_psnr_db": float(p)})
    assert cos >= 0.999, (ma, cos)
    assert p >= 40.0, p


def test_matches_cpu_checker_on_fresh_inputs():
    """Same seeded inputs through the HIP path and through oracle/ (no golden file involved)."""
    import dataclasses
    sc = dataclasses.replace(SCENARIOS["tiny_edit3"], noise_seed=1234, weight_seed=3, num_timesteps=4)
    run = OracleRun(sc)
    otrace, trace = [], []
    run.generate(trace=otrace, max_steps=2)
    run_scenario_on_gpu(sc, run, trace=trace, max_steps=2, teacher_x=[r["x_t"] for r in otrace])
    for a, b in zip(trace, otrace):
        for key in ("v_cond", "v_text", "v_img"):
            ma, cos = metrics(a[key], b[key])
            assert cos >= 0.999 and ma <= 4e-2, (key, ma, cos)


def test_random_scenarios_match_cpu_checker():
    """Random batch sizes, image sizes, context lengths per CFG branch (empty contexts included), depths, CFG scales,
    intervals and renorm types through the HIP path and through oracle/ on the same seeded inputs, teacher-forced.
    BAGEL_FUZZ_SCENARIOS=n runs n of them (default 4)."""
    import os
    import random
    from bagel_amd.config import tiny_dims
    from tests.scenarios import Scenario
    rnd = random.Random(2024)
    n_cases = int(os.environ.get("BAGEL_FUZZ_SCENARIOS", 4))
    for case in range(n_cases):
        if case and case % 50 == 0:
            print(f"  random scenarios: {case} / {n_cases}", flush=True)
        B = rnd.randint(1, 3)
        images = [(16 * rnd.randint(2, 14), 16 * rnd.randint(2, 14)) for _ in range(B)]
        lens = lambda: [rnd.choice([1, 2, 5, 31, 62, 63, 64, 65, 130, 200]) for _ in range(B)]   # noqa: E731
        cond = lens()
        text = rnd.choice([None, lens()])
        img = rnd.choice([None, cond, lens()])
        sc = Scenario(f"random{case}", tiny_dims(rnd.randint(1, 3)), images, cond, text, img, num_timesteps=5,
                      timestep_shift=rnd.choice([1.0, 3.0]), cfg_text_scale=rnd.choice([1.0, 2.5, 4.0]),
                      cfg_img_scale=rnd.choice([1.0, 1.5, 2.0]), cfg_interval=rnd.choice([(0.0, 1.0), (0.4, 1.0), (0.3, 0.8)]),
                      cfg_renorm_min=rnd.choice([0.0, 0.3]), cfg_renorm_type=rnd.choice(["global", "channel", "text_channel"]),
                      weight_seed=case % 3, noise_seed=500 + case)
        what = f"case {case}: {images} cond {cond} text {text} img {img} L={sc.dims.num_layers} scales " \
               f"{sc.cfg_text_scale}/{sc.cfg_img_scale} {sc.cfg_interval} {sc.cfg_renorm_type} min {sc.cfg_renorm_min}"
        run = OracleRun(sc)
        exact_run = OracleRun(sc, sd=run.sd, exact=True)
        otrace, trace = [], []
        run.generate(trace=otrace, max_steps=3)
        run_scenario_on_gpu(sc, run, trace=trace, max_steps=3, teacher_x=[r["x_t"] for r in otrace])
        assert len(trace) == len(otrace) == 3, what
        for i, (a, b) in enumerate(zip(trace, otrace)):
            vt_e, (vc_e, vx_e, vm_e) = exact_run.step(i, b["x_t"])
            exact = {"v_cond": vc_e, "v_text": vx_e, "v_img": vm_e, "v_t": vt_e}
            for key in ("v_cond", "v_text", "v_img", "v_t"):
                if b.get(key) is None or key not in a:
                    continue
                ma, cos = metrics(a[key], b[key])
                # a combined velocity amplifies the branches' rounding differences by the CFG scales
                amp = 1.0 if key != "v_t" else max(sc.cfg_text_scale, 1.0) * max(sc.cfg_img_scale, 1.0) * 2.0
                assert cos >= (0.999 if key != "v_t" else 0.995) and ma <= 4e-2 * amp, (what, i, key, ma, cos)
                # and relative to the checker's own rounding: the HIP path may be no further from exact (fp32, no
                # rounding points) arithmetic on the same inputs than the bf16 checker is, up to sampling noise on
                # the mean and a rounding that went the other way in the top binade on the maximum
                if exact[key] is None:
                    continue
                st = parity_stats(a[key], b[key], exact[key])
                assert st["hip_vs_exact_mean"] <= (EXACT_RATIO_MEAN + mean_slack(st)) * st["ref_vs_exact_mean"] + 1e-6, \
                    (what, i, key, "mean distance to exact", st["hip_vs_exact_mean"], st["ref_vs_exact_mean"])
                assert st["hip_vs_exact_max"] <= REF_RATIO_MAX * st["ref_vs_exact_max"] + 1e-6, \
                    (what, i, key, "max distance to exact", st["hip_vs_exact_max"], st["ref_vs_exact_max"])
    print(f"{n_cases} random scenarios agree with the CPU checker")


def test_api_errors():
    sc = SCENARIOS["tiny_t2i"]
    run = OracleRun(sc)
    from tests.gpu_common import gpu_model
    model = gpu_model(sc, run.sd)
    gi = dict(run.gi)
    bad = dict(gi)
    bad["packed_vae_token_indexes"] = gi["packed_vae_token_indexes"].flip(0)
    with pytest.raises(NotImplementedError):
        model.generate_image(past_key_values=None, **bad)


class _Tok:
    """Stand-in tokenizer: a prompt is a string of space-separated token ids."""

    def encode(self, text):
        return [int(t) for t in text.split()]


@pytest.mark.parametrize("name", ["tiny_t2i", "tiny_edit3", "tiny_batch2", "gqa7_t2i", "w7b_1024", "w7b_l28_1024"])
def test_text_prefill_matches_reference(name):
    """Bagel.prepare_prompts + forward_cache_update_text (causal und-mode prefill on HIP kernels) against
    the context keys/values the reference produced for the cond branch."""
    from tests.gpu_common import gpu_model
    from tests.scenarios import NEW_TOKEN_IDS
    from bagel_amd.modeling import NaiveCache
    sc = SCENARIOS[name]
    g = load_golden(name)
    model = gpu_model(sc, scenario_state_dict(sc.dims, sc.weight_seed))
    prompts = sc.prompts("cond")
    n = len(prompts)
    gi, newlens, newrope = model.prepare_prompts([0] * n, [0] * n, [" ".join(map(str, p)) for p in prompts], _Tok(),
                                                 NEW_TOKEN_IDS)
    assert newlens == [len(p) + 2 for p in prompts] and newrope == newlens
    past = model.forward_cache_update_text(NaiveCache(sc.dims.num_layers), **gi)
    last = sc.dims.num_layers - 1
    from tests.helpers import check_tensor_vs_reference
    for key, got in (("k0", past.key_cache[0]), ("v0", past.value_cache[0]),
                     ("k_last", past.key_cache[last]), ("v_last", past.value_cache[last])):
        ref = g["cond_ctx." + key]
        assert got.shape == ref.shape
        exact = g.get("cond_ctx.exact." + key)
        assert exact is not None, f"golden {name!r} holds no exact-arithmetic context: run tests/golden/add_exact_context.py"
        check_tensor_vs_reference(got, ref, exact, name, "cond_ctx." + key)


def test_prefill_with_existing_context_matches_checker():
    """Second prompt appended to a non-empty context (causal mask offset by the context length)."""
    from tests.gpu_common import gpu_model, to_gpu_cache
    from tests.scenarios import NEW_TOKEN_IDS
    from oracle import bagel_oracle as O
    from bagel_amd.weights import make_llm_state_dict
    sc = SCENARIOS["tiny_edit3"]
    sd = make_llm_state_dict(sc.dims, seed=sc.weight_seed)
    model = gpu_model(sc, sd)
    p1, p2 = [list(range(10, 10 + 70))], [list(range(100, 100 + 150))]
    with torch.no_grad():
        ogi, l1, r1 = O.prepare_prompts([0], [0], p1, NEW_TOKEN_IDS)
        opast = O.forward_cache_update_text(sd, sc.dims, O.Cache(sc.dims.num_layers), **ogi)
        ogi2, l2, r2 = O.prepare_prompts(l1, r1, p2, NEW_TOKEN_IDS)
        gpast = to_gpu_cache(opast, sc.dims.num_layers)
        opast = O.forward_cache_update_text(sd, sc.dims, opast, **ogi2)
    gi2, gl2, gr2 = model.prepare_prompts(l1, r1, [" ".join(map(str, p2[0]))], _Tok(), NEW_TOKEN_IDS)
    assert (gl2, gr2) == (l2, r2)
    gpast = model.forward_cache_update_text(gpast, **gi2)
    for l in (0, sc.dims.num_layers - 1):
        for got, ref in ((gpast.key_cache[l], opast.key_cache[l]), (gpast.value_cache[l], opast.value_cache[l])):
            assert got.shape == ref.shape
            ma, cos = metrics(got, ref)
            assert ma <= 0.03 * float(ref.float().abs().max()) and cos >= 0.999, (l, ma, cos)


def test_inferencer_text_to_image_end_to_end():
    """InterleaveInferencer(text=...) -> PIL image, all on HIP kernels (prefill, denoise loop, VAE decode),
    against the reference's final latent decoded by the CPU checker.  Bar: PSNR >= 40 dB."""
    from bagel_amd.inferencer import InterleaveInferencer
    from bagel_amd.modeling import AutoEncoder
    from bagel_amd.config import VAEDims
    from bagel_amd.weights import make_llm_state_dict, make_vae_decoder_state_dict
    from tests.gpu_common import gpu_model
    from tests.scenarios import NEW_TOKEN_IDS
    from oracle import bagel_oracle as O
    import numpy as np
    sc = SCENARIOS["tiny_t2i"]
    g = load_golden("tiny_t2i")
    model = gpu_model(sc, scenario_state_dict(sc.dims, sc.weight_seed))
    vd = VAEDims()
    vsd = make_vae_decoder_state_dict(vd, seed=0)
    inf = InterleaveInferencer(model, AutoEncoder(vd, vsd, "cuda:0"), _Tok(), None, None, NEW_TOKEN_IDS)
    torch.manual_seed(sc.noise_seed)
    out = inf(text=" ".join(map(str, sc.prompts("cond")[0])), cfg_text_scale=sc.cfg_text_scale,
              cfg_img_scale=sc.cfg_img_scale, cfg_interval=list(sc.cfg_interval), timestep_shift=sc.timestep_shift,
              num_timesteps=sc.num_timesteps, cfg_renorm_min=sc.cfg_renorm_min, cfg_renorm_type=sc.cfg_renorm_type,
              image_shapes=sc.images[0])
    img = out["image"]
    assert img is not None and img.size == (sc.images[0][1], sc.images[0][0]) and out["text"] is None
    with torch.no_grad():
        ref = O.vae_decode(vsd, vd, O.unpatchify_latent(sc.dims, g["final_latent"], sc.images[0]), bf16=True)
    ref8 = O.to_uint8_image(ref.float()).numpy().astype(np.float32)
    got8 = np.asarray(img).astype(np.float32)
    mse = float(((ref8 - got8) ** 2).mean())
    p = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    print(f"end-to-end T2I image PSNR vs reference pipeline: {p:.1f} dB")
    assert p >= 40.0


@pytest.mark.parametrize("name", ["tiny_taylor", "tiny_taylor_late"])
def test_taylorseer_matches_reference(name):
    """generate_image(enable_taylorseer=True): a branch's first five steps are full, then one in three; skipped steps
    expand the cached last-layer output.  Teacher-forced from the reference's per-step inputs, branch by branch.
    In ``tiny_taylor_late`` the CFG branches start at step 4 and keep their own schedules, so some steps run the layers
    for a subset of the branches only."""
    from tests.gpu_common import run_scenario_on_gpu
    from tests.scenarios import TAYLOR_SCENARIOS
    sc = TAYLOR_SCENARIOS[name]
    g = load_golden(sc.name)
    n = sc.num_timesteps - 1
    trace = []
    run_scenario_on_gpu(sc, OracleRun(sc), trace=trace, teacher_x=[g[f"step{i}.x_t"] for i in range(n)])
    assert len(trace) == n
    for i, rec in enumerate(trace):
        for key in ("v_cond", "v_text", "v_img"):
            gk = f"step{i}.{key}"
            assert (key in rec) == (gk in g), (i, key)
            if key in rec:
                ma, cos = metrics(rec[key], g[gk])
                assert ma <= 2.5e-2 and cos >= 0.999, (i, key, ma, cos)


def test_taylorseer_skipped_steps_ignore_x_t():
    """On a skipped step the velocity does not depend on the step's input (the layers are not run)."""
    import dataclasses
    from tests.gpu_common import run_scenario_on_gpu
    from tests.scenarios import TAYLOR_SCENARIOS
    sc = TAYLOR_SCENARIOS["tiny_taylor"]
    g = load_golden(sc.name)
    n = sc.num_timesteps - 1
    xs = [g[f"step{i}.x_t"] for i in range(n)]
    run = OracleRun(sc)
    a, b = [], []
    run_scenario_on_gpu(sc, run, trace=a, teacher_x=xs)
    xs2 = [x.clone() for x in xs]
    xs2[5] = xs2[5] + 1.0                                     # steps 5 and 6 are skipped ones
    xs2[6] = xs2[6] * -2.0
    run_scenario_on_gpu(sc, run, trace=b, teacher_x=xs2)
    for i in range(n):
        assert torch.equal(a[i]["v_cond"], b[i]["v_cond"]), i


def test_dedup_identical_branches_changes_nothing():
    """Text-to-image gives the image-free CFG branch the conditional branch's context: with
    dedup_identical_branches that branch is not run.  Same latents bit for bit, with and without; a branch with a
    different context is never aliased."""
    import dataclasses
    from tests.gpu_common import gpu_model, to_gpu_cache
    sc = SCENARIOS["tiny_batch2"]             # batch of two images; the conditional cache also serves as the image-free one
    run = OracleRun(sc)
    model = gpu_model(sc, run.sd)
    L = sc.dims.num_layers
    cond = to_gpu_cache(run.ctx["cond"][0], L)
    text = to_gpu_cache(run.ctx["text"][0], L)
    gi, gt, gm = run.gi, run.gt, run.gm

    def go(img_cache, dedup, trace):
        return model.generate_image(
            past_key_values=cond, num_timesteps=sc.num_timesteps, timestep_shift=sc.timestep_shift,
            cfg_renorm_min=sc.cfg_renorm_min, cfg_renorm_type=sc.cfg_renorm_type, cfg_interval=list(sc.cfg_interval),
            cfg_text_scale=sc.cfg_text_scale, cfg_text_past_key_values=text,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"], cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"], cfg_img_scale=sc.cfg_img_scale,
            cfg_img_past_key_values=img_cache, cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"], cfg_img_key_values_lens=gm["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], trace=trace,
            dedup_identical_branches=dedup, **gi)

    ta, tb, tc = [], [], []
    a = go(cond, False, ta)
    b = go(cond, True, tb)                       # the same cache object: aliased
    assert len(ta) == len(tb) and model.engine.kws is not None
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    for ra, rb in zip(ta, tb):
        assert set(ra) == set(rb)
        for k in ra:
            assert torch.equal(ra[k], rb[k]), k
    # an equal-valued but separately stored context is not aliased (nothing is compared by value): still the same result
    from bagel_amd.modeling import NaiveCache
    other = NaiveCache(L)
    for l in range(L):
        other.key_cache[l], other.value_cache[l] = cond.key_cache[l].clone(), cond.value_cache[l].clone()
    c = go(other, True, tc)
    for x, y in zip(a, c):
        assert torch.equal(x, y)
