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
        ex = plan_exchange([u[0] for u in active], [u[1] for u in active], sample_nv, rank, world)
        local = [active[i] for i in ex.local_units]
        local_v = torch.cat([_unit_velocity(b, c, sample_nv[b], width) for b, c in local]) if local else None
        got = exchange_velocities(local_v, ex, width, torch.device("cpu"), torch.bfloat16)
        want = torch.cat([_unit_velocity(b, c, sample_nv[b], width) for b, c in active])
        assert torch.equal(got, want), f"rank {rank}: gathered velocities differ"
    q.put((rank, [unit_owner(b, c, len(sample_nv), world) for b, c in units]))
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
    """CFG steps AND cond-only steps are balanced to within one unit for one, two (text CFG only,
    eval/gen/gen_images_mp.py:74-80) and three branches; a cond unit keeps its rank between the two kinds of step."""
    for world in (1, 2, 3, 4, 8):
        for B in (1, 2, 3, 8, 16):
            for n_br in (1, 2, 3):
                own = [unit_owner(b, c, B, world) for c in range(n_br) for b in range(B)]
                counts = [own.count(r) for r in range(world)]
                assert max(counts) - min(counts) <= 1, (world, B, n_br, counts)
                cond = [unit_owner(b, 0, B, world) for b in range(B)]
                ccounts = [cond.count(r) for r in range(world)]
                assert max(ccounts) - min(ccounts) <= 1, (world, B, n_br, ccounts)
                assert cond == own[:B]


def test_cond_units_are_a_prefix_of_each_ranks_units():
    """generate_image shares one KV workspace between the two kinds of step: a rank's cond-only units must be the
    first of its CFG-step units."""
    for world in (2, 4, 8):
        for B in (1, 3, 8):
            for n_br in (2, 3):
                units = [(b, c) for c in range(n_br) for b in range(B)]
                nv = [8] * B
                for rank in range(world):
                    full = plan_exchange([u[0] for u in units], [u[1] for u in units], nv, rank, world).local_units
                    cond = plan_exchange(list(range(B)), [0] * B, nv, rank, world).local_units
                    assert [units[i] for i in full][:len(cond)] == [(b, 0) for b in cond]


def test_identity_gather_is_detected():
    ex = plan_exchange([0, 1, 2, 3], [0, 0, 0, 0], [8, 8, 8, 8], 1, 4)      # cond-only step, one image per rank
    assert ex.identity and ex.local_units == [1]
    ex = plan_exchange([0, 1, 0, 1], [0, 0, 1, 1], [8, 8], 0, 2)             # two units per rank: re-ordered
    assert not ex.identity


def test_single_rank_is_identity():
    ex = plan_exchange([0, 1, 0, 1], [0, 0, 1, 1], [4, 6], 0, 1)
    v = torch.randn(20, 64).to(torch.bfloat16)
    assert exchange_velocities(v, ex, 64, torch.device("cpu"), torch.bfloat16) is v
    assert ex.local_units == [0, 1, 2, 3]
