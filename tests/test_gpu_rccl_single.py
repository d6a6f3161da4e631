"""The per-step velocity exchange through RCCL on the GPU.  The pool has one GPU per box, so this runs a one-rank
"nccl" group with the gather forced on: it exercises the device-side send/receive staging, the bf16
all_gather_into_tensor and the row re-ordering of bagel_amd/parallel.py (the multi-rank unit dealing is covered on
the CPU with gloo in test_parallel_gloo.py)."""
import os

import pytest
import torch
import torch.distributed as dist

from tests.helpers import OracleRun
from tests.scenarios import SCENARIOS

pytestmark = pytest.mark.gpu


def test_generate_image_with_forced_rccl_gather(monkeypatch):
    from tests.gpu_common import run_scenario_on_gpu
    from bagel_amd import parallel
    sc = SCENARIOS["tiny_batch2"]
    run = OracleRun(sc)
    base = [x.clone() for x in run_scenario_on_gpu(sc, run)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        calls = []
        real = dist.all_gather_into_tensor

        def counting(recv, send, group=None):
            calls.append(tuple(send.shape))
            return real(recv, send, group=group)

        monkeypatch.setattr(dist, "all_gather_into_tensor", counting)
        monkeypatch.setattr(parallel.StepExchange, "needs_gather", property(lambda self: True))
        out = run_scenario_on_gpu(sc, run)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert len(calls) == sc.num_timesteps - 1 and all(c[1] == sc.dims.patch_latent_dim for c in calls)
    for a, b in zip(out, base):
        assert torch.equal(a, b)
