"""Multi-GPU framing: the framebuffer is tiled by rows across ranks, one process per GPU.

Pixels are independent (reference README.md:81-84, docs/renderer.md:34-35), so rank g of P renders the
contiguous row slab ``row_slab(H, g, P)`` with a full replica of the (small) primitive arrays and the
slabs are collected on one rank by a single gather per frame -- RCCL over xGMI when the process group
is "nccl", gloo in the CPU tests.  Nothing is reduced, so there is no ring all-reduce on the path.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def row_slab(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [r0, r1) owned by ``rank``: contiguous, in rank order, sizes differing by at most one
    (the first ``height % world`` ranks take the extra row).  Slabs may be empty when world > height."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} of {world}")
    base, extra = divmod(height, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def all_slabs(height: int, world: int) -> List[Tuple[int, int]]:
    return [row_slab(height, g, world) for g in range(world)]


class GatherHandle:
    """Completion handle of one frame's gather (possibly several point-to-point requests)."""

    def __init__(self, works=()):
        self._works = [w for w in works if w is not None]

    def wait(self) -> None:
        for w in self._works:
            w.wait()
        self._works = []


def gather_rows(slab: torch.Tensor, full: Optional[torch.Tensor], height: int, dst: int = 0,
                group: Optional[dist.ProcessGroup] = None, async_op: bool = False) -> GatherHandle:
    """Collect per-rank row slabs ``slab`` ((h_g, W, ...) contiguous) into ``full`` ((H, W, ...), only
    needed on ``dst``).  Equal slabs use one ``gather`` whose receive list are views of ``full`` (no
    staging copy); ragged slabs fall back to one batched send/recv group.  With ``async_op`` the call
    returns at once and the handle's ``wait()`` orders later work on the current stream after the transfer,
    which lets the next frame's render overlap this frame's gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    slabs = all_slabs(height, world)
    if world == 1:
        if full is not None and full.data_ptr() != slab.data_ptr():
            full.copy_(slab)
        return GatherHandle()
    equal = len({r1 - r0 for r0, r1 in slabs}) == 1
    views = None
    if rank == dst:
        if full is None or full.shape[0] != height:
            raise ValueError("the destination rank must pass the full (H, W, ...) buffer")
        views = [full[r0:r1] for r0, r1 in slabs]
    if equal:
        work = dist.gather(slab, gather_list=views, dst=dst, group=group, async_op=async_op)
        return GatherHandle([work] if async_op else [])
    ops = []
    if rank == dst:
        r0, r1 = slabs[dst]
        if full[r0:r1].data_ptr() != slab.data_ptr():
            full[r0:r1].copy_(slab)
        for g, (a, b) in enumerate(slabs):
            if g != dst and b > a:
                ops.append(dist.P2POp(dist.irecv, views[g], g, group))
    elif slab.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, slab, dst, group))
    handle = GatherHandle(dist.batch_isend_irecv(ops) if ops else [])
    if not async_op:
        handle.wait()
    return handle


def exchange_frames(send: torch.Tensor, recv: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                    async_op: bool = False) -> GatherHandle:
    """Gather a BATCH of P = world frames at once, frame k of the batch on rank k.

    ``send`` (P, h, ...) holds this rank's row slab of P consecutive frames; on return ``recv`` (P, h, ...) holds,
    on rank k, slab g of frame k from every rank g -- with equal slabs ``recv.view(H, ...)`` IS the assembled frame k.
    Per frame this moves exactly what a gather to one root moves, but as one all-to-all per P frames every rank
    receives over all of its xGMI links at once instead of rank 0 receiving everything over its own: a 2048 x 2048
    fp32 rgb+depth frame is 67 MB, and a single root's links cap a fixed-root gather at a few thousand frames/s.
    NCCL/RCCL: one ``all_to_all_single``; other backends (gloo in the CPU tests): one batched send/recv group."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if send.shape != recv.shape or send.shape[0] != world:
        raise ValueError(f"send and recv must both be ({world}, h, ...), got {tuple(send.shape)} and {tuple(recv.shape)}")
    if not (send.is_contiguous() and recv.is_contiguous()):
        raise ValueError("send and recv must be contiguous")
    if world == 1:
        recv.copy_(send)
        return GatherHandle()
    if dist.get_backend(group) == "nccl":
        work = dist.all_to_all_single(recv, send, group=group, async_op=async_op)
        return GatherHandle([work] if async_op else [])
    recv[rank].copy_(send[rank])
    ops = []
    for g in range(world):
        if g != rank:
            ops.append(dist.P2POp(dist.isend, send[g], g, group))
            ops.append(dist.P2POp(dist.irecv, recv[g], g, group))
    handle = GatherHandle(dist.batch_isend_irecv(ops))
    if not async_op:
        handle.wait()
    return handle
