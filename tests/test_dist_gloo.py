"""Row-slab framing over torch.distributed with the gloo backend, world_size 2 (CPU).

The hip kernel cannot run here, so each rank renders its slab with the CPU oracle standing in for the kernel;
what is under test is the partition, the single-gather collection into strided (H, 4W) framebuffers, and that
the assembled frame equals the full-frame render."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from surf_renderer_amd.dist import all_slabs, gather_rows, row_slab


def test_row_slabs_partition_the_image():
    for height in (1, 2, 7, 48, 2048):
        for world in (1, 2, 3, 8):
            slabs = all_slabs(height, world)
            assert slabs[0][0] == 0 and slabs[-1][1] == height
            for (a0, a1), (b0, b1) in zip(slabs, slabs[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = [b - a for a, b in slabs]
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == height
    with pytest.raises(ValueError):
        row_slab(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, height, width, q, cost=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import np_oracle
        from surf_renderer_amd import synthetic
        from surf_renderer_amd.dist import cost_weighted_slabs
        from surf_renderer_amd.scene import scene_to_numpy
        scene = scene_to_numpy(synthetic.demo_scene(width, height, with_planes=True), round_fp32=True)
        slabs = cost_weighted_slabs(cost, height, world) if cost is not None else None
        r0, r1 = slabs[rank] if slabs else row_slab(height, rank, world)
        part = np_oracle.render(scene, rows=(r0, r1))
        # the bench's framebuffer layout: one (rows, 4W) slab, [W x rgb | W x depth] per row
        if rank == 0:
            frame = torch.zeros((height, 4 * width), dtype=torch.float32)
            slab = frame[r0:r1]
        else:
            frame = None
            slab = torch.zeros((r1 - r0, 4 * width), dtype=torch.float32)
        image = slab.as_strided((r1 - r0, width, 3), (4 * width, 3, 1), slab.storage_offset())
        depth = slab.as_strided((r1 - r0, width), (4 * width, 1), slab.storage_offset() + 3 * width)
        image.copy_(torch.from_numpy(part["image"].astype(np.float32)))
        depth.copy_(torch.from_numpy(part["depth"].astype(np.float32)))
        gather_rows(slab, frame, height, dst=0, async_op=(height % 2 == 0), slabs=slabs).wait()
        if rank == 0:
            full = np_oracle.render(scene)
            fb = frame.numpy()
            got_img = fb[:, :3 * width].reshape(height, width, 3)
            got_depth = fb[:, 3 * width:]
            ok = np.array_equal(got_img, full["image"].astype(np.float32)) and \
                np.array_equal(got_depth, full["depth"].astype(np.float32))
            q.put(bool(ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height", [48, 37])          # equal slabs -> gather; ragged -> batched send/recv
def test_two_rank_row_slab_gather(height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, height, 40, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world", [2, 3])
def test_cost_weighted_slabs_assemble_the_whole_frame(world):
    """Ranks render slabs of equal WORK (here: a made-up cost per tile row, heavy in the middle) and rank 0 still
    receives every row exactly once."""
    height = 80
    cost = [1, 1, 9, 30, 2]                              # five tile rows of 16 image rows
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, 40, q, cost)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_cost_weighted_slabs_balance_the_work():
    from surf_renderer_amd.dist import cost_weighted_slabs
    rng = np.random.RandomState(0)
    for height, world in ((2048, 8), (2048, 4), (2048, 2), (1000, 3), (16, 1)):
        nrows = (height + 15) // 16
        x = np.linspace(-1, 1, nrows)
        cost = 100.0 * np.exp(-4 * x * x) + rng.uniform(0, 5, nrows)       # busiest in the middle, like BASELINE config 5
        slabs = cost_weighted_slabs(cost, height, world)
        assert slabs[0][0] == 0 and slabs[-1][1] == height
        assert all(slabs[g][1] == slabs[g + 1][0] for g in range(world - 1))
        assert all(a % 16 == 0 and b > a for a, b in slabs)
        work = [cost[a // 16:(b + 15) // 16].sum() for a, b in slabs]
        equal = [cost[a // 16:(b + 15) // 16].sum() for a, b in
                 [(g * (height // world) // 16 * 16, (g + 1) * (height // world) // 16 * 16) for g in range(world)]]
        if world > 1 and nrows >= 8 * world:
            assert max(work) <= 1.15 * sum(work) / world          # within a tile row's worth of the mean
            assert max(work) < max(equal) or world == 2
    with pytest.raises(ValueError):
        cost_weighted_slabs([1.0] * 3, 64, 2)                     # 4 tile rows, 3 costs
    with pytest.raises(ValueError):
        cost_weighted_slabs([1.0] * 2, 32, 3)                     # more ranks than tile rows


def _exchange_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from surf_renderer_amd.dist import exchange_frames
        h, w = 3, 5
        # slab of rank g for frame k of the batch carries the value 100 g + k in every element
        send = torch.stack([torch.full((h, w), 100.0 * rank + k) for k in range(world)])
        recv = torch.full((world, h, w), -1.0)
        exchange_frames(send, recv, async_op=(rank % 2 == 0)).wait()
        want = torch.stack([torch.full((h, w), 100.0 * g + rank) for g in range(world)])
        q.put((rank, bool(torch.equal(recv, want))))
        bad = torch.zeros((world + 1, h, w))
        try:
            exchange_frames(bad, bad.clone())
            q.put((rank, False))
        except ValueError:
            q.put((rank, True))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_batched_frame_exchange_puts_frame_k_on_rank_k(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(2 * world)]
    assert all(ok for _, ok in got), got


def _batcher_worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from surf_renderer_amd.dist import FrameBatcher
        h, w = 2, 3

        def render(i, slot):
            slot.fill_(1000.0 * i + rank)              # "slab of rank `rank` for frame i"

        def render_batch(first, send):
            for k in range(world):
                send[k].fill_(1000.0 * (first + k) + rank)

        fb = FrameBatcher(world, (h, w), torch.float32, "cpu", render)
        ok = True
        i = 0
        while i < n_frames:
            # whole batches alternately in one go (what one srh_render_views call does) and frame by frame
            if i % world == 0 and n_frames - i >= world and (i // world) % 2 == 0:
                fb.submit_batch(render_batch)
                i += world
            else:
                fb.submit()
                i += 1
        fb.flush()
        # after the flush every receive buffer holds the last batch delivered into it
        for b in range(fb.n_batches):
            batch = fb.delivered[b]
            if batch < 0:
                continue
            frame = batch * world + rank               # the frame this rank assembled in that batch
            for g in range(world):
                want = 1000.0 * frame + g
                rendered = frame < n_frames            # slots of a partial last batch were never rendered
                if rendered and not bool(torch.all(fb.frame(b)[g] == want)):
                    ok = False
        q.put((rank, ok, fb.count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 8), (3, 7), (2, 5)])
def test_frame_batcher_schedule(world, n_frames):
    """The batched collection's schedule (bench.py runs it on HIP streams): frame k of a batch ends up complete on
    rank k, batch buffers are reused only after their exchange, and a partial last batch is delivered on flush."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batcher_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [q.get(timeout=5) for _ in range(world)]
    assert all(ok for _, ok, _ in got), got
    assert all(count == -(-n_frames // world) * world for _, _, count in got)


def test_balanced_slabs_partition_the_image():
    from surf_renderer_amd.dist import balanced_slabs
    for height, world in ((2048, 8), (2048, 2), (96, 3), (16, 1)):
        rows = []
        for g in range(world):
            parts = balanced_slabs(height, g, world)
            assert len(parts) == 2 and all(b - a == height // (2 * world) for a, b in parts)
            rows += [r for a, b in parts for r in range(a, b)]
        assert sorted(rows) == list(range(height))
    with pytest.raises(ValueError):
        balanced_slabs(100, 0, 8)


def test_owner_slabs_partition_every_frame():
    from surf_renderer_amd.dist import owner_slabs
    for height, world, frac in ((2048, 2, 0.875), (2048, 4, 0.4), (96, 3, 0.5), (2048, 8, 0.125), (48, 2, 0.9)):
        rows = owner_slabs(height, world, frac)
        assert len(rows) == world
        per_rank = [0] * world
        for k, frame in enumerate(rows):
            assert frame[0][0] == 0 and frame[-1][1] == height
            for (a0, a1), (b0, b1) in zip(frame, frame[1:]):
                assert a1 == b0
            sizes = [b - a for a, b in frame]
            assert sizes[k] == max(sizes) and len({s for g, s in enumerate(sizes) if g != k}) == 1
            for g, s in enumerate(sizes):
                per_rank[g] += s
        assert len(set(per_rank)) == 1 and per_rank[0] == height      # every rank renders one frame's worth per batch
    assert owner_slabs(2048, 2, 0.875)[0] == [(0, 1792), (1792, 2048)]
    assert owner_slabs(2048, 2, 0.875)[1] == [(0, 256), (256, 2048)]
    with pytest.raises(ValueError):
        owner_slabs(64, 2, 0.2)


def _owner_worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from surf_renderer_amd.dist import FrameBatcher, owner_slabs
        height, w = 64, 3
        rows = owner_slabs(height, world, 0.75)
        send_rows = [rows[k][rank][1] - rows[k][rank][0] for k in range(world)]
        recv_rows = [rows[rank][g][1] - rows[rank][g][0] for g in range(world)]

        def render(i, slot):
            a, b = rows[i % world][rank]
            assert slot.shape[0] == b - a
            # every row carries frame number and absolute row index: the assembled frame is then checkable row by row
            slot.copy_((1000.0 * i + torch.arange(a, b, dtype=torch.float32))[:, None].expand(b - a, w))

        fb = FrameBatcher(world, (0, w), torch.float32, "cpu", render, send_rows=send_rows, recv_rows=recv_rows)
        for _ in range(n_frames):
            fb.submit()
        fb.flush()
        ok = True
        for b in range(fb.n_batches):
            batch = fb.delivered[b]
            if batch < 0:
                continue
            frame = batch * world + rank
            if frame >= n_frames:
                continue
            want = (1000.0 * frame + torch.arange(height, dtype=torch.float32))[:, None].expand(height, w)
            ok = ok and bool(torch.equal(fb.frame(b), want))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 6), (3, 9), (2, 5)])
def test_owner_weighted_batches_assemble_whole_frames(world, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert all(results.values()), results
