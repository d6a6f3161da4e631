"""Drive the HIP path of a scenario on cuda:0 (used by the -m gpu tests and by smoke())."""
from __future__ import annotations

import torch

from bagel_amd.modeling import Bagel, NaiveCache

_MODELS = {}


def gpu_model(sc, sd):
    key = (sc.dims, sc.weight_seed)
    if key not in _MODELS:
        _MODELS.clear()  # one model resident at a time
        torch.cuda.empty_cache()
        _MODELS[key] = Bagel(sc.dims, state_dict=sd, device="cuda:0")
    return _MODELS[key]


def to_gpu_cache(cache, num_layers):
    """CPU checker Cache -> device NaiveCache."""
    out = NaiveCache(num_layers)
    for l in range(num_layers):
        if cache.key_cache[l] is not None:
            out.key_cache[l] = cache.key_cache[l].to("cuda:0")
            out.value_cache[l] = cache.value_cache[l].to("cuda:0")
    return out


def run_scenario_on_gpu(sc, run, trace=None, max_steps=None, teacher_x=None):
    """``run`` is a tests.helpers.OracleRun: it provides the weights, the three context caches (built on
    the CPU by the checker's prefill) and the reference-identical index tensors."""
    model = gpu_model(sc, run.sd)
    L = sc.dims.num_layers
    gi, gt, gm = run.gi, run.gt, run.gm
    return model.generate_image(
        past_key_values=to_gpu_cache(run.ctx["cond"][0], L), num_timesteps=sc.num_timesteps,
        timestep_shift=sc.timestep_shift, cfg_renorm_min=sc.cfg_renorm_min, cfg_renorm_type=sc.cfg_renorm_type,
        cfg_interval=list(sc.cfg_interval), cfg_text_scale=sc.cfg_text_scale,
        cfg_text_past_key_values=to_gpu_cache(run.ctx["text"][0], L),
        cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
        cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
        cfg_text_key_values_lens=gt["cfg_key_values_lens"],
        cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
        cfg_img_scale=sc.cfg_img_scale, cfg_img_past_key_values=to_gpu_cache(run.ctx["img"][0], L),
        cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
        cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"],
        cfg_img_key_values_lens=gm["cfg_key_values_lens"],
        cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"],
        max_steps=max_steps, trace=trace, teacher_x=teacher_x, enable_taylorseer=sc.enable_taylorseer, **gi)
