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


def balanced_slabs(height: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Two half-slabs per rank, g and P+g of 2P equal ones.  For a scene whose density is symmetric about the image
    centre and grows towards it, slab g (upper half: the farther down, the busier) and slab P+g (lower half: the farther
    down, the emptier) add up to about the same load for every rank.  Needs height % (2 * world) == 0."""
    if height % (2 * world):
        raise ValueError(f"balanced slabs need height % {2 * world} == 0, got {height}")
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"rank {rank} of {world}")
    hh = height // (2 * world)
    a, b = rank, world + rank
    return [(a * hh, (a + 1) * hh), (b * hh, (b + 1) * hh)]


def all_slabs(height: int, world: int) -> List[Tuple[int, int]]:
    return [row_slab(height, g, world) for g in range(world)]


def cost_weighted_slabs(tile_row_cost, height: int, world: int, granule: int = 16,
                        per_tile_row: float = 0.0) -> List[Tuple[int, int]]:
    """Contiguous row slabs, in rank order, whose WORK is equal rather than their height.  With tile bins the cost of a
    row is not uniform (SURVEY 8e's caveat): at BASELINE config 5 the middle slabs of eight see twice the candidates of
    the outer ones and set the pace.  ``tile_row_cost[i]`` = work of tile row i (``granule`` image rows; e.g.
    ``renderer.bin_statistics(...)["tile_row_cost"]`` of one probe frame, identical on every rank because the scene
    replicas are), ``per_tile_row`` = a fixed cost added to every tile row (empty tiles still cost their background
    stores).  Boundaries are multiples of ``granule``; every rank gets at least one tile row; the prefix sums of the
    slabs' costs are as close to k / world of the total as whole tile rows allow."""
    import numpy as np
    if world < 1:
        raise ValueError(f"world {world}")
    cost = np.asarray(tile_row_cost, dtype=np.float64) + float(per_tile_row)
    nrows = (height + granule - 1) // granule
    if cost.shape != (nrows,):
        raise ValueError(f"{cost.shape[0] if cost.ndim == 1 else cost.shape} tile rows of cost for {nrows} tile rows of image")
    if world > nrows:
        raise ValueError(f"{world} ranks for {nrows} tile rows")
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for k in range(1, world):
        want = cum[-1] * k / world
        at = int(np.argmin(np.abs(cum - want)))
        at = min(max(at, cuts[-1] + 1), nrows - (world - k))     # at least one tile row for this rank and for those after it
        cuts.append(at)
    cuts.append(nrows)
    return [(cuts[g] * granule, min(cuts[g + 1] * granule, height)) for g in range(world)]


def owner_slabs(height: int, world: int, owner_frac: float, granule: int = 16) -> List[List[Tuple[int, int]]]:
    """Owner-weighted row partition for the batched collection: ``rows[k][g]`` = the rows rank g renders of the frame
    that rank k assembles.  The owner takes about ``owner_frac`` of the frame, the others share the rest equally; the
    slabs of one frame are contiguous, in rank order, and sized in multiples of ``granule`` rows (the tile height).
    With rotating owners every rank still renders one frame's worth of rows per batch of P frames, but only
    (1 - owner_frac) / (P - 1) of a frame crosses each xGMI link -- at P = 2 an equal split is bound by the one link
    between the two GPUs, four times over."""
    if world < 2:
        return [[(0, height)]]
    if not (1.0 / world <= owner_frac < 1.0):
        raise ValueError(f"owner_frac must be in [1/P, 1), got {owner_frac}")
    small = int((1.0 - owner_frac) * height / (world - 1)) // granule * granule
    small = max(small, min(granule, height // world))
    big = height - (world - 1) * small
    if big < small or small < 1:
        raise ValueError(f"cannot split {height} rows over {world} ranks with owner_frac {owner_frac}")
    rows = []
    for k in range(world):
        at, frame = 0, []
        for g in range(world):
            n = big if g == k else small
            frame.append((at, at + n))
            at += n
        rows.append(frame)
    return rows


class GatherHandle:
    """Completion handle of one frame's gather (possibly several point-to-point requests)."""

    def __init__(self, works=()):
        self._works = [w for w in works if w is not None]

    def wait(self) -> None:
        for w in self._works:
            w.wait()
        self._works = []


def gather_rows(slab: torch.Tensor, full: Optional[torch.Tensor], height: int, dst: int = 0,
                group: Optional[dist.ProcessGroup] = None, async_op: bool = False,
                slabs: Optional[List[Tuple[int, int]]] = None) -> GatherHandle:
    """Collect per-rank row slabs ``slab`` ((h_g, W, ...) contiguous) into ``full`` ((H, W, ...), only
    needed on ``dst``).  Equal slabs use one ``gather`` whose receive list are views of ``full`` (no
    staging copy); ragged slabs fall back to one batched send/recv group.  With ``async_op`` the call
    returns at once and the handle's ``wait()`` orders later work on the current stream after the transfer,
    which lets the next frame's render overlap this frame's gather.  ``slabs`` = the ranks' row ranges when they are
    not ``row_slab``'s (``cost_weighted_slabs``); the same list on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if slabs is None:
        slabs = all_slabs(height, world)
    elif len(slabs) != world or slabs[0][0] != 0 or slabs[-1][1] != height or \
            any(slabs[g][1] != slabs[g + 1][0] for g in range(world - 1)):
        raise ValueError(f"slabs {slabs} do not partition {height} rows over {world} ranks")
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


def exchange_frames_uneven(send: torch.Tensor, recv: torch.Tensor, send_rows: List[int], recv_rows: List[int],
                           group: Optional[dist.ProcessGroup] = None, async_op: bool = False) -> GatherHandle:
    """``exchange_frames`` for slabs of different heights (``owner_slabs``): ``send`` (sum(send_rows), ...) holds this
    rank's slab of frame k at rows [sum(send_rows[:k]), +send_rows[k]); on return ``recv`` (sum(recv_rows), ...) holds
    the slab of every rank g of THIS rank's frame at [sum(recv_rows[:g]), +recv_rows[g]) -- i.e. the assembled frame.
    NCCL/RCCL: one ``all_to_all_single`` with split sizes; other backends: one batched send/recv group."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(send_rows) != world or len(recv_rows) != world:
        raise ValueError("send_rows / recv_rows need one entry per rank")
    if send.shape[0] != sum(send_rows) or recv.shape[0] != sum(recv_rows) or send.shape[1:] != recv.shape[1:]:
        raise ValueError(f"buffers {tuple(send.shape)} / {tuple(recv.shape)} do not match the row counts")
    if not (send.is_contiguous() and recv.is_contiguous()):
        raise ValueError("send and recv must be contiguous")
    if send_rows[rank] != recv_rows[rank]:
        raise ValueError("a rank's own slab must have the same height on both sides")
    s_at = [sum(send_rows[:k]) for k in range(world)]
    r_at = [sum(recv_rows[:g]) for g in range(world)]
    if world == 1:
        recv.copy_(send)
        return GatherHandle()
    if dist.get_backend(group) == "nccl":
        work = dist.all_to_all_single(recv, send, output_split_sizes=list(recv_rows), input_split_sizes=list(send_rows),
                                      group=group, async_op=async_op)
        return GatherHandle([work] if async_op else [])
    recv[r_at[rank]:r_at[rank] + recv_rows[rank]].copy_(send[s_at[rank]:s_at[rank] + send_rows[rank]])
    ops = []
    for g in range(world):
        if g != rank:
            ops.append(dist.P2POp(dist.isend, send[s_at[g]:s_at[g] + send_rows[g]], g, group))
            ops.append(dist.P2POp(dist.irecv, recv[r_at[g]:r_at[g] + recv_rows[g]], g, group))
    handle = GatherHandle(dist.batch_isend_irecv(ops))
    if not async_op:
        handle.wait()
    return handle


class FrameBatcher:
    """Scheduling of the batched collection: frames are rendered into `send[b][k]` (b = batch buffer, k = frame within
    the batch of P = world frames), a full batch is exchanged (`exchange_frames`) so that frame k lands on rank k, and a
    batch buffer is reused only after its exchange has completed.  Stream handling is the caller's, through three
    hooks, so the same schedule runs on HIP streams (bench.py) and on CPU tensors under gloo (the tests):

        render(i, slot)          fill `slot` ((h, ...) view of the send buffer) with this rank's slab of frame i
        before_exchange()        make the exchange wait for every render issued so far
        after_reuse_wait()       make later renders wait until the exchange that just completed has released a buffer

    `frame(b)` is the receive buffer of batch buffer b: on rank k it holds frame (batch * P + k), slab g at [g]."""

    def __init__(self, world: int, slab_shape, dtype, device, render, before_exchange=None, after_reuse_wait=None,
                 n_batches: int = 2, group: Optional[dist.ProcessGroup] = None,
                 send_rows: Optional[List[int]] = None, recv_rows: Optional[List[int]] = None):
        """With ``send_rows`` / ``recv_rows`` (``owner_slabs``: this rank's slab height per frame of a batch, and every
        rank's slab height of this rank's own frame) the buffers are flat -- send (sum(send_rows), *slab_shape[1:]),
        recv (H, ...) = the assembled frame -- and slot k is rows [sum(send_rows[:k]), +send_rows[k]) of send."""
        self.world, self.group, self.n_batches = world, group, n_batches
        self.send_rows, self.recv_rows = send_rows, recv_rows
        if send_rows is None:
            self.send = [torch.empty((world, *slab_shape), dtype=dtype, device=device) for _ in range(n_batches)]
            self.recv = [torch.empty((world, *slab_shape), dtype=dtype, device=device) for _ in range(n_batches)]
        else:
            tail = tuple(slab_shape[1:])
            self.send = [torch.empty((sum(send_rows), *tail), dtype=dtype, device=device) for _ in range(n_batches)]
            self.recv = [torch.empty((sum(recv_rows), *tail), dtype=dtype, device=device) for _ in range(n_batches)]
            self._send_at = [sum(send_rows[:k]) for k in range(world)]
        self.pending: List[Optional[GatherHandle]] = [None] * n_batches
        self.delivered: List[int] = [-1] * n_batches          # index of the batch each receive buffer holds
        self._render, self._before, self._after = render, before_exchange, after_reuse_wait
        self.count = 0

    def slot(self, i: int) -> Tuple[int, int]:
        return (i // self.world) % self.n_batches, i % self.world

    def slot_view(self, b: int, k: int) -> torch.Tensor:
        """The part of send buffer b that takes this rank's slab of frame k of a batch."""
        if self.send_rows is None:
            return self.send[b][k]
        return self.send[b][self._send_at[k]:self._send_at[k] + self.send_rows[k]]

    def submit(self, *render_args) -> None:
        """Render the next frame; exchange its batch when it is the last of it."""
        i = self.count
        self.count += 1
        b, k = self.slot(i)
        if k == 0 and self.pending[b] is not None:
            self.pending[b].wait()                    # this buffer's previous batch has left
            self.pending[b] = None
            if self._after:
                self._after()
        self._render(i, self.slot_view(b, k), *render_args)
        if k == self.world - 1:
            self._exchange(b, i // self.world)

    def submit_batch(self, render_batch, *render_args) -> None:
        """Render a whole batch at once -- ``render_batch(first_frame, send_buffer, *args)`` fills all P slots of the
        (P, h, ...) send buffer, e.g. with one ``srh_render_views`` call -- and exchange it.  Only at a batch boundary."""
        if self.count % self.world:
            raise RuntimeError("submit_batch in the middle of a batch")
        batch_index = self.count // self.world
        b = batch_index % self.n_batches
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None
            if self._after:
                self._after()
        render_batch(self.count, self.send[b], *render_args)
        self.count += self.world
        self._exchange(b, batch_index)

    def _exchange(self, b: int, batch_index: int) -> None:
        if self._before:
            self._before()
        if self.send_rows is None:
            self.pending[b] = exchange_frames(self.send[b], self.recv[b], group=self.group, async_op=True)
        else:
            self.pending[b] = exchange_frames_uneven(self.send[b], self.recv[b], self.send_rows, self.recv_rows,
                                                     group=self.group, async_op=True)
        self.delivered[b] = batch_index

    def flush(self) -> None:
        """Deliver an unfinished batch too (its unrendered slots travel as they are) and wait for everything."""
        i = self.count
        if i % self.world:
            self._exchange((i // self.world) % self.n_batches, i // self.world)
            self.count = (i // self.world + 1) * self.world
        for b in range(self.n_batches):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def frame(self, b: int) -> torch.Tensor:
        return self.recv[b]
