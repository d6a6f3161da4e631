"""Benchmark of the image-generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): BAGEL-7B-MoT dims, 1024x1024 text-to-image, num_timesteps=50
(49 Euler steps), text + image CFG with cfg_interval (0.4, 1.0): 41 steps run three branches, 8 run one,
131 LLM forwards per image; random-init weights and a synthetic 34-token context (no checkpoint or
tokenizer exists offline).  One "step" of this benchmark = one complete ``Bagel.generate_image`` call for
the whole batch (N images on N GPUs: weak scaling, three units per rank per CFG step) followed by the FLUX
VAE decode of every image to uint8.
Inputs (weights, context KV, noise) are resident in HBM before the timed region.

Prints ONE JSON line.  ``roofline`` describes the dominant kernel (the generation expert's gate+up GEMM): achieved
TFLOP/s from HIP events around its launches inside the timed region against the dense bf16 MFMA peak, plus
fabric-side bytes per launch from separate counter passes.  ``cpu_baseline`` is the CPU checker timed on this
host's cores on a bounded sample (a reported baseline, not a target).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layers", type=int, default=None, help="debug only: fewer layers (result is then INVALID)")
    ap.add_argument("--resolution", type=int, default=1024)
    ap.add_argument("--num-timesteps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="debug only: process-group backend (gloo lets several ranks "
                                                      "rehearse on one GPU together with --share-gpu)")
    ap.add_argument("--force-gather", action="store_true",
                    help="debug only: at one GPU, open a one-rank RCCL group and run the per-step all-gather anyway")
    ap.add_argument("--share-gpu", action="store_true", help="debug only: every rank uses cuda:0 (result is then INVALID)")
    ap.add_argument("--taylorseer", action="store_true",
                    help="enable_taylorseer=True (two of three forwards skipped after step 4): a different workload, "
                         "not the headline configuration")
    ap.add_argument("--dedup-identical-branches", action="store_true",
                    help="skip the image-free CFG branch where it has the conditional branch's context (always, in "
                         "text-to-image): same results, a different amount of work - not the headline configuration")
    return ap.parse_args()


def synthetic_context(dims, n_samples, ctx_len, device, seed):
    """NaiveCache with ``ctx_len`` random context tokens per sample (post-norm keys have unit RMS)."""
    from bagel_amd.modeling import NaiveCache
    cache = NaiveCache(dims.num_layers)
    if ctx_len == 0:
        return cache
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for l in range(dims.num_layers):
        cache.key_cache[l] = torch.randn((n_samples * ctx_len, dims.num_kv_heads, dims.head_dim), device=device,
                                         dtype=torch.bfloat16, generator=g)
        cache.value_cache[l] = torch.randn((n_samples * ctx_len, dims.num_kv_heads, dims.head_dim), device=device,
                                           dtype=torch.bfloat16, generator=g) * 0.5
    return cache


def cpu_baseline(dims_full, resolution, num_timesteps):
    """The CPU checker (oracle/, a port of the reference's PyTorch path) timed on this host's cores on a
    bounded sample: one three-branch CFG flow step at BAGEL-7B width, 4 of 28 layers, same resolution (10-15 s).
    Scaled to images/sec by layers and by the 131/3 step-equivalents per image."""
    import dataclasses
    from bagel_amd.config import width7b_dims
    from tests.helpers import OracleRun
    from tests.scenarios import Scenario
    layers = 4
    sc = Scenario("cpu_baseline", width7b_dims(layers), [(resolution, resolution)], [32], None, [32],
                  num_timesteps=num_timesteps)
    run = OracleRun(sc)
    x = run.gi["packed_init_noises"]
    t0 = time.perf_counter()
    run.step(0, x)
    dt = time.perf_counter() - t0
    forwards_per_image = 131 if num_timesteps == 50 else 3 * (num_timesteps - 1)
    per_forward_full = dt / 3 * (dims_full.num_layers / layers)
    return {"value": 1.0 / (per_forward_full * forwards_per_image), "unit": "images/sec",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/bagel_oracle.py, one 3-branch CFG flow step at 7B width, {layers} of "
                      f"{dims_full.num_layers} layers, {resolution}x{resolution}: {dt:.2f} s; scaled by layers and "
                      f"{forwards_per_image} forwards/image"}


def measured_counters(kernel_prefix):
    """Counters of the dominant kernel's three-branch (M = 12288) launches from the rocprofv3 --pmc passes over this
    script (tools/run_pmc_bench.sh -> tools/pmc_to_profile.py -> profiles/bench_pmc.json).  Counters cannot be read
    without the profiler, so they come from that committed file; absent file -> traffic null."""
    path = os.path.join(ROOT, "profiles", "bench_pmc.json")
    none = {"traffic": None, "traffic_source": "no profiles/bench_pmc.json"}
    if not os.path.exists(path):
        return none
    table = json.load(open(path))
    for k in table["kernels"]:
        if k["kernel"].startswith(kernel_prefix) and k["size_class"] == "large" and "fetch_bytes_per_launch" in k:
            out = {"traffic": k["fetch_bytes_per_launch"] + k.get("write_bytes_per_launch", 0.0),
                   "traffic_source": "HBM/Infinity-Cache side bytes per M = 12288 launch, rocprofv3 --pmc FETCH_SIZE (x2, the "
                                     "gfx950 correction) and WRITE_SIZE passes over this script: " + table.get("command", ""),
                   "algorithmic_bytes": 2.0 * (12288 * 3584 + 37888 * 3584 + 12288 * 18944)}
            if "clock_ghz_held" in k:
                held = MFMA_BF16_PEAK_TFLOPS * k["clock_ghz_held"] / 2.4
                out.update({"clock_ghz_held_profiled": k["clock_ghz_held"], "mfma_busy_frac_profiled": k["mfma_busy_frac"],
                            "peak_at_held_clock": held})
            return out
    return none


def main():
    args = parse()
    if os.environ.get("BENCH_FAULT_TIMEOUT"):       # debug: dump every thread's stack if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_FAULT_TIMEOUT"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world == 1 and args.force_gather:
        import torch.distributed as dist
        from bagel_amd import parallel
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        parallel.StepExchange.needs_gather = property(lambda self: True)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from bagel_amd.config import BAGEL_7B
    from bagel_amd.engine import MoTWeights
    from bagel_amd.modeling import Bagel
    import dataclasses
    dims = BAGEL_7B if args.layers is None else dataclasses.replace(BAGEL_7B, num_layers=args.layers)
    weights = MoTWeights.random(dims, dev, seed=0, vocab_rows=1024)
    model = Bagel(dims, device=dev, weights=weights)
    new_token_ids = dict(bos_token_id=1, eos_token_id=2, start_of_image=3, end_of_image=4)

    B = world                     # weak scaling: one image per GPU
    res = args.resolution
    ctx_len = 34                  # 32-token prompt + bos/eos
    cond = synthetic_context(dims, B, ctx_len, dev, 1)
    empty = synthetic_context(dims, B, 0, dev, 2)
    torch.manual_seed(42)
    gi = model.prepare_vae_latent([ctx_len] * B, [ctx_len] * B, [(res, res)] * B, new_token_ids)
    gt = model.prepare_vae_latent_cfg([0] * B, [0] * B, [(res, res)] * B)
    gm = model.prepare_vae_latent_cfg([ctx_len] * B, [ctx_len] * B, [(res, res)] * B)
    gi["packed_init_noises"] = gi["packed_init_noises"].to(dev)

    def one_image_batch():
        return model.generate_image(
            past_key_values=cond, num_timesteps=args.num_timesteps, timestep_shift=3.0, cfg_renorm_min=0.0,
            cfg_renorm_type="global", cfg_interval=[0.4, 1.0], cfg_text_scale=4.0, cfg_text_past_key_values=empty,
            cfg_text_packed_position_ids=gt["cfg_packed_position_ids"],
            cfg_text_packed_query_indexes=gt["cfg_packed_query_indexes"],
            cfg_text_key_values_lens=gt["cfg_key_values_lens"],
            cfg_text_packed_key_value_indexes=gt["cfg_packed_key_value_indexes"],
            cfg_img_scale=1.5, cfg_img_past_key_values=cond,
            cfg_img_packed_position_ids=gm["cfg_packed_position_ids"],
            cfg_img_packed_query_indexes=gm["cfg_packed_query_indexes"],
            cfg_img_key_values_lens=gm["cfg_key_values_lens"],
            cfg_img_packed_key_value_indexes=gm["cfg_packed_key_value_indexes"], enable_taylorseer=args.taylorseer,
            dedup_identical_branches=args.dedup_identical_branches, **gi)

    # FLUX VAE decoder (random-init at the true FLUX dims); images are decoded by the rank that owns them
    from bagel_amd.config import VAEDims
    from bagel_amd.modeling import AutoEncoder
    from bagel_amd.weights import make_vae_decoder_state_dict
    vd = VAEDims()
    ae = AutoEncoder(vd, make_vae_decoder_state_dict(vd, seed=0), dev)
    p_, c_ = dims.latent_patch_size, dims.latent_channel
    hh = ww = res // dims.latent_downsample

    def decode(latent):
        z = latent.reshape(1, hh, ww, p_, p_, c_)
        z = torch.einsum("nhwpqc->nchpwq", z).reshape(1, c_, hh * p_, ww * p_)
        img = ae.decode(z)
        return ((img * 0.5 + 0.5).clamp(0, 1)[0].permute(1, 2, 0) * 255).to(torch.uint8)

    # everything above (28 GB of weights, contexts) was queued asynchronously: let it finish before the first step.
    # (A rocprofv3 --pmc pass crashed inside the profiler at the first launch issued while several hundred
    # initialisation kernels were still in flight; drained like this the same pass runs.)
    torch.cuda.synchronize()
    if os.environ.get("BENCH_DUMP_MAPS"):           # debug: the process's address map before the first step
        with open(os.environ["BENCH_DUMP_MAPS"], "w") as f:
            f.write(open("/proc/self/maps").read())

    def one_step():
        latents = one_image_batch()
        return [decode(latents[b]) for b in range(B) if b % world == rank]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    # per-launch timing of the dominant kernel with HIP events on the launch stream, during the timed region
    from bagel_amd import _lib as K
    K.profile_gate_up(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    K.profile_gate_up(False)
    events = K.profile_collect()            # (milliseconds, flops) per gate+up launch of the timed region
    ranks_info = {"world_size_seen": 1, "backend": None, "per_rank_ms_per_step": [elapsed / args.steps * 1e3],
                  "devices": [torch.cuda.get_device_name(dev)]}
    if world > 1:
        import torch.distributed as dist
        per_rank = [None] * world
        dist.all_gather_object(per_rank, (elapsed / args.steps * 1e3, torch.cuda.get_device_name(dev), local_rank))
        ranks_info = {"world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                      "per_rank_ms_per_step": [p[0] for p in per_rank], "devices": [f"{p[1]} (local rank {p[2]})" for p in per_rank]}
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        n_euler = args.num_timesteps - 1
        ms_per_image_batch = elapsed / args.steps * 1e3
        out = {
            "metric": "images/sec (node) 1024² T2I 50-step CFG; ms per flow step at 1/2/4/8 GPU",
            "value": B * args.steps / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_image_batch, "ms_per_flow_step": ms_per_image_batch / n_euler,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"BAGEL-7B-MoT dims ({dims.num_layers} layers), {res}x{res} T2I, "
                                   f"num_timesteps={args.num_timesteps}, cfg_text 4.0 + cfg_img 1.5, "
                                   f"cfg_interval (0.4,1.0), global renorm, random-init weights, 34-token context; "
                                   f"one step = one generate_image call for the batch + VAE decode to uint8 of every image",
                       "ranks": ranks_info, "taylorseer": bool(args.taylorseer), "dedup_identical_branches": bool(args.dedup_identical_branches), "global_batch": B, "units_per_rank_per_cfg_step": 3, "parallelism": f"cfg-branch x batch over {world} GPU"},
        }
        if events:
            # the dominant launch shape: all CFG branches packed (M = 3 x 4096 rows); the cond-only steps' smaller
            # launches of the same kernel are left out so that achieved, traffic and duration describe one shape
            fmax = max(f for _, f in events)
            ms = [m for m, f in events if f == fmax]
            flops = fmax
            avg = sum(ms) / len(ms)
            ach = flops / (avg * 1e-3) / 1e12
            pmc = measured_counters("gemm256_pipelined_kernel<2, 256, 4")
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / MFMA_BF16_PEAK_TFLOPS, **pmc,
                               "kernel": "gemm256_pipelined_kernel<EPI_SWIGLU, 256, 4> (gen-expert gate+up projection)",
                               "launches_timed": len(ms), "avg_launch_ms": avg, "avg_flops_per_launch": flops}
        if world == 1 and not args.no_cpu_baseline and args.layers is None:   # reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(dims, res, args.num_timesteps)
        print(json.dumps(out))
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
