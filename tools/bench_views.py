#!/usr/bin/env python3
"""Whole frames through the batched entry point (srh_render_views): V frames per library call, S calls in flight on
S streams.  Diagnostic: us per frame against the per-frame pipeline of bench.py."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from surf_renderer_amd import renderer, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=4)
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--calls", type=int, default=100)
ap.add_argument("--prims", type=int, default=100_000)
ap.add_argument("--size", type=int, default=2048)
args = ap.parse_args()
dev = torch.device("cuda:0")
W = H = args.size
sc = synthetic.disk_cloud_scene(args.prims, W, H)
buf = renderer.flatten_scene(sc, dev)
cam = renderer.camera_struct(sc["camera"], "numpy")
V, S = args.views, args.streams
streams = [torch.cuda.Stream(dev) for _ in range(S)]
outs = [torch.empty((V, H, 4 * W), dtype=torch.float32, device=dev) for _ in range(S)]
wss = [None] * S

def call(j):
    img = outs[j].as_strided((V, H, W, 3), (H * 4 * W, 4 * W, 3, 1), 0)
    dep = outs[j].as_strided((V, H, W), (H * 4 * W, 4 * W, 1), 3 * W)
    with torch.cuda.stream(streams[j]):
        wss[j] = renderer.render_views_buffers(buf, [cam] * V, img, dep, rows=(0, H), workspace=wss[j],
                                               image_row_stride=4 * W, depth_row_stride=4 * W)
for i in range(2 * S):
    call(i % S)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.calls):
    call(i % S)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"views {V} streams {S}: {1e6 * dt / (args.calls * V):.1f} us per frame")
