"""The whole N > 1 generation path on real kernels: two or three processes share the box's one GPU (RCCL refuses two ranks
on one device, so the group is gloo; the exchange itself is the same all_gather_into_tensor call), split the
(sample, CFG-branch) units of every step, and must reproduce the single-process result bit for bit - CFG steps and
cond-only steps, batch 2 with global renorm and an editing case with three different contexts."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, name, q):
    import torch.distributed as dist
    from tests.gpu_common import run_scenario_on_gpu
    from tests.helpers import OracleRun
    from tests.scenarios import SCENARIOS, TAYLOR_SCENARIOS, EXTRA_SCENARIOS
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = {**SCENARIOS, **TAYLOR_SCENARIOS, **EXTRA_SCENARIOS}[name]
        calls = []
        real = dist.all_gather_into_tensor

        def counting(recv, send, group=None):
            calls.append(tuple(send.shape))
            return real(recv, send, group=group)

        dist.all_gather_into_tensor = counting
        out = run_scenario_on_gpu(sc, OracleRun(sc))
        torch.cuda.synchronize()
        # numpy arrays travel through the queue by value; torch tensors would be shared by file descriptor, which the
        # parent can only collect while this process is still alive
        q.put((rank, len(calls), [x.float().cpu().numpy() for x in out]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("tiny_batch2", 2), ("tiny_edit3", 2), ("tiny_t2i", 2), ("tiny_t2i", 3),
                                        ("tiny_textcfg", 2), ("tiny_taylor", 2), ("tiny_taylor_late", 3), ("tiny_batch12", 2)])
def test_ranks_reproduce_one(name, world):
    """tiny_t2i has cond-only steps (outside the CFG interval) in which all ranks but one own no unit at all;
    tiny_edit3 splits one image's three branches 2 + 1; tiny_textcfg has two branches per sample; tiny_batch12 packs
    36 units = 72 text rows into the single process's forward (more than one 64-row launch of the understanding
    expert's weight-streaming kernel) against 36 rows per rank."""
    from tests.gpu_common import run_scenario_on_gpu
    from tests.helpers import OracleRun
    from tests.scenarios import SCENARIOS, TAYLOR_SCENARIOS, EXTRA_SCENARIOS
    sc = {**SCENARIOS, **TAYLOR_SCENARIOS, **EXTRA_SCENARIOS}[name]
    base = [x.float().cpu() for x in run_scenario_on_gpu(sc, OracleRun(sc))]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, n_gathers, out = q.get(timeout=240)
        assert n_gathers == sc.num_timesteps - 1      # one exchange per Euler step
        got[rank] = [torch.from_numpy(x) for x in out]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):      # x_t is replicated: both ranks hold every sample's final latent
        assert len(got[r]) == len(base)
        for a, b in zip(got[r], base):
            assert torch.equal(a, b)
