"""Sharding of a flow step over the GPUs of one node.

The unit of work is a (sample, CFG branch) pair: within a step every unit reads the same-step ``x_t`` of
its sample and its own read-only KV context and produces one velocity block ``[Nv, 64]``; no unit reads
another's activations (separate forward_inference calls in the reference, bagel.py:1413-1464).  Units are
dealt round-robin over the ranks, each rank runs its units as one packed forward, and ONE all-gather per
step (RCCL over xGMI; <= 0.5 MiB per unit) gives every rank every velocity, after which each rank runs the
CFG-combine + Euler kernel for all samples so that ``x_t`` stays replicated and no second exchange is
needed.  The ``global`` renorm (bagel.py:1485-1488) needs the norm over all samples' velocities, which is
why the combine runs after the gather.  The reference itself never splits an image across GPUs
(``cfg_type="parallel"`` is accepted and ignored, bagel.py:1269,1380).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def world_info(group=None) -> Tuple[int, int]:
    """(rank, world_size); (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def unit_owner(sample: int, branch: int, n_samples: int, world: int) -> int:
    """Round-robin over the BRANCH-major unit index (all cond units, then all text-free units, then all image-free
    units).  The cond unit of sample b lands on rank b % world in CFG steps and in cond-only steps alike, so its KV
    context is staged once, and both kinds of step are balanced to within one unit for any number of branches: the
    cond-only steps spread over every rank (a sample-major deal puts all cond units of a two-branch request - text
    CFG only, eval/gen/gen_images_mp.py:74-80 - on the even ranks of an even world)."""
    return (branch * n_samples + sample) % world


@dataclass
class StepExchange:
    """How one kind of step (all branches / cond only) is split and re-assembled."""
    local_units: List[int]        # indices into the branch-major global unit list owned by this rank
    rows_max: int                 # padded per-rank row count of the gather buffer
    index: torch.Tensor           # [n_branches_active * Nv] row of the gathered buffer for every global row
    world: int
    identity: bool = False        # the gathered buffer already is the global row order (index = 0, 1, 2, ...)
    send: Optional[torch.Tensor] = None   # [rows_max, width] and [world * rows_max, width]: allocated at the first
    recv: Optional[torch.Tensor] = None   # step of an image and reused by every later step of this kind

    @property
    def needs_gather(self) -> bool:
        return self.world > 1


def plan_exchange(unit_sample: Sequence[int], unit_branch: Sequence[int], sample_nv: Sequence[int], rank: int,
                  world: int) -> StepExchange:
    """``unit_*`` describe the global branch-major unit list of this kind of step."""
    owners = [unit_owner(s, b, len(sample_nv), world) for s, b in zip(unit_sample, unit_branch)]
    rows = [0] * world
    offset = []
    for s, o in zip(unit_sample, owners):
        offset.append(rows[o])
        rows[o] += sample_nv[s]
    rows_max = max(rows) if rows else 0
    idx = []
    for s, o, off in zip(unit_sample, owners, offset):
        base = o * rows_max + off
        idx.append(torch.arange(base, base + sample_nv[s], dtype=torch.long))
    index = torch.cat(idx) if idx else torch.empty(0, dtype=torch.long)
    local = [i for i, o in enumerate(owners) if o == rank]
    identity = bool(torch.equal(index, torch.arange(index.numel(), dtype=torch.long)))
    return StepExchange(local, rows_max, index, world, identity)


def exchange_velocities(local_v: Optional[torch.Tensor], ex: StepExchange, width: int, device, dtype,
                        group=None) -> torch.Tensor:
    """All-gather the ranks' velocity blocks and return them in global branch-major row order.

    ``local_v`` is this rank's ``[rows_local, width]`` block (None when it owns no unit in this step).
    """
    if not ex.needs_gather:
        return local_v
    if ex.send is None or ex.send.shape[1] != width or ex.send.dtype != dtype or ex.send.device != torch.device(device):
        ex.send = torch.zeros((ex.rows_max, width), device=device, dtype=dtype)     # pad rows stay zero
        ex.recv = torch.empty((ex.world * ex.rows_max, width), device=device, dtype=dtype)
        ex.index = ex.index.to(device)
    send, recv = ex.send, ex.recv
    if local_v is not None and local_v.shape[0] > 0:
        send[:local_v.shape[0]].copy_(local_v)
    if send.is_cuda and dist.get_backend(group) == "gloo":
        # gloo stages device tensors through the host on its own threads; with kernels still queued on the launch
        # stream that hand-off was seen to hang (ROCm), so the rehearsal backend gets a drained stream.  RCCL
        # ("nccl") orders the gather behind the stream's work itself.
        torch.cuda.current_stream().synchronize()
    dist.all_gather_into_tensor(recv, send, group=group)
    if ex.identity:            # one equal-sized unit per rank, in rank order: nothing to re-order
        return recv[:ex.index.numel()]
    return recv.index_select(0, ex.index)
