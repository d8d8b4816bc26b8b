#!/usr/bin/env python3
"""Per-rank cost of the row-tiled multi-GPU path, measured on ONE GPU: renders the row slab every rank of a
`--world`-rank job would own (same scene, same kernels, no gather) and prints ms per frame for each.
    python tools/slab_bench.py --world 8
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surf_renderer_amd import renderer, synthetic
from surf_renderer_amd.dist import row_slab

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--prims", type=int, default=100_000)
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--inflight", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
W = H = a.size
scene = synthetic.disk_cloud_scene(a.prims, W, H)
buf = renderer.flatten_scene(scene, dev)
cam = renderer.camera_struct(scene["camera"])
out = []
for g in range(a.world):
    r0, r1 = row_slab(H, g, a.world)
    h = r1 - r0
    streams = [torch.cuda.Stream(dev) for _ in range(a.inflight)]
    ws = [buf.new_workspace(W, H) for _ in range(a.inflight)]
    img = [torch.empty((h, W, 3), device=dev) for _ in range(a.inflight)]
    dep = [torch.empty((h, W), device=dev) for _ in range(a.inflight)]
    def run(n):
        for i in range(n):
            b = i % a.inflight
            with torch.cuda.stream(streams[b]):
                renderer.render_buffers(buf, cam, rows=(r0, r1), out=(img[b], dep[b], None), workspace=ws[b])
        torch.cuda.synchronize()
    run(30)
    t0 = time.perf_counter(); run(a.steps); dt = time.perf_counter() - t0
    out.append(round(1e3 * dt / a.steps, 4))
print(json.dumps({"world": a.world, "ms_per_frame_by_rank": out, "max": max(out), "fps_bound": 1e3 / max(out)}))
