#!/usr/bin/env python3
"""Per-rank rendering cost of owner-weighted batches (dist.owner_slabs), measured on ONE GPU: for P ranks and an owner
fraction f, the time rank g needs for its slabs of the P frames of a batch (one library call per frame, three streams),
and the bytes each xGMI link would carry.  Numbers quoted in DESIGN.md section 5."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from surf_renderer_amd import renderer, synthetic
from surf_renderer_amd.dist import owner_slabs, row_slab
W = H = 2048
scene = synthetic.disk_cloud_scene(100_000, W, H)
buf = renderer.flatten_scene(scene, "cuda:0"); cam = renderer.camera_struct(scene["camera"])
n_str = 3
streams = [torch.cuda.Stream() for _ in range(n_str)]
scratch = [buf.new_workspace(W, H) for _ in range(n_str)]
def views(slab):
    hh = slab.shape[0]
    return (slab.as_strided((hh, W, 3), (4 * W, 3, 1), slab.storage_offset()), slab.as_strided((hh, W), (4 * W, 1), slab.storage_offset() + 3 * W))
def cost(rows):
    sends = [torch.empty((sum(b - a for a, b in rows), 4 * W), device="cuda:0") for _ in range(2)]
    n = len(rows)
    def batch(i):
        at = 0
        for k, (a, b) in enumerate(rows):
            j = (n * i + k) % n_str
            with torch.cuda.stream(streams[j]):
                renderer.render_buffers(buf, cam, rows=(a, b), out=(*views(sends[i % 2][at:at + b - a]), None), workspace=scratch[j])
            at += b - a
    for i in range(10): batch(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): batch(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 100
for P, fracs in ((2, (0.875, 0.9375)), (4, (0.4, 0.625, 0.8125)), (8, (0.3, 0.5625, 0.78125))):
    for f in fracs:
        rows_all = owner_slabs(H, P, f)
        costs = [cost([rows_all[k][g] for k in range(P)]) for g in range(P)]
        small = rows_all[0][1][1] - rows_all[0][1][0]
        link_mb = small * W * 16 / 1e6
        print(f"P={P} f={f}: per-rank ms per batch of {P} frames: " + " ".join(f"{c * 1e3:.3f}" for c in costs) + f" | slowest -> {P / max(costs):.0f} fps | {link_mb:.1f} MB per link and batch = {link_mb / 64:.3f} ms at 64 GB/s", flush=True)
