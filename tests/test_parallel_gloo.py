"""The N>1 path on CPU: two gloo ranks split the (sample, CFG-branch) units of a step, exchange their
velocity blocks with one all-gather and must rebuild the same branch-major velocity tensor a single rank has.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bagel_amd.parallel import exchange_velocities, plan_exchange, unit_owner


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unit_velocity(sample, branch, nv, width):
    """Stand-in for a unit's forward: a deterministic function of (sample, branch, row)."""
    r = torch.arange(nv, dtype=torch.float32)[:, None]
    c = torch.arange(width, dtype=torch.float32)[None, :]
    return (1000 * sample + 100 * branch + r + c / 128).to(torch.bfloat16)


def _worker(rank, world, port, sample_nv, n_br, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    width = 64
    # global branch-major unit list, as Bagel.generate_image builds it
    units = [(b, c) for c in range(n_br) for b in range(len(sample_nv))]
    for active in (units, [(b, 0) for b in range(len(sample_nv))]):   # CFG step, cond-only step
        ex = plan_exchange([u[0] for u in active], [u[1] for u in active], sample_nv, n_br, rank, world)
        local = [active[i] for i in ex.local_units]
        local_v = torch.cat([_unit_velocity(b, c, sample_nv[b], width) for b, c in local]) if local else None
        got = exchange_velocities(local_v, ex, width, torch.device("cpu"), torch.bfloat16)
        want = torch.cat([_unit_velocity(b, c, sample_nv[b], width) for b, c in active])
        assert torch.equal(got, want), f"rank {rank}: gathered velocities differ"
    q.put((rank, [unit_owner(b, c, n_br, world) for b, c in units]))
    dist.destroy_process_group()


def _run(world, sample_nv, n_br):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, sample_nv, n_br, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return [q.get(timeout=10) for _ in range(world)]


def test_two_ranks_three_branches_mixed_sizes():
    res = _run(2, [16, 24, 8], 3)
    owners = res[0][1]
    assert sorted(set(owners)) == [0, 1]
    assert abs(owners.count(0) - owners.count(1)) <= 1


def test_two_ranks_single_image():
    _run(2, [64], 3)


def test_four_and_eight_ranks_one_image_per_rank():
    """The layouts `bench.py --gpus 4 / 8` runs: one image per rank, three branches; in the cond-only steps every
    rank owns exactly one unit, and with a single image on four ranks one rank owns none."""
    for world, nv in ((4, [8, 8, 8, 8]), (8, [8] * 8), (4, [12])):
        res = _run(world, nv, 3)
        owners = res[0][1]
        counts = [owners.count(r) for r in range(world)]
        assert max(counts) - min(counts) <= 1
        assert all(r[1] == owners for r in res)


def test_owner_balance():
    for world in (1, 2, 4, 8):
        for B in (1, 2, 8):
            own = [unit_owner(b, c, 3, world) for b in range(B) for c in range(3)]
            counts = [own.count(r) for r in range(world)]
            assert max(counts) - min(counts) <= 1
            # cond units keep their rank between CFG steps and cond-only steps
            assert all(unit_owner(b, 0, 3, world) == (3 * b) % world for b in range(B))


def test_single_rank_is_identity():
    ex = plan_exchange([0, 1, 0, 1], [0, 0, 1, 1], [4, 6], 2, 0, 1)
    v = torch.randn(20, 64).to(torch.bfloat16)
    assert exchange_velocities(v, ex, 64, torch.device("cpu"), torch.bfloat16) is v
    assert ex.local_units == [0, 1, 2, 3]
