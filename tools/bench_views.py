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

# GAN-shaped batch (diffrend/torch/GAN/gan.py:325-378): every view its own splat set and its own light 0 -- one
# render_views call with per-view overrides against one render() per element
import numpy as np                                           # noqa: E402
from surf_renderer_amd import render, render_views          # noqa: E402
B, M, R = 64, 4096, 128
rng = np.random.RandomState(0)
base = synthetic.disk_cloud_scene(M, R, R, radius=0.05, seed=1)
cams = [dict(base["camera"], eye=[*map(float, 4.0 * e / np.linalg.norm(e)), 1.0]) for e in rng.normal(size=(B, 3))]
pos = torch.tensor(np.concatenate([rng.uniform(-1, 1, (B, M, 3)), np.ones((B, M, 1))], 2).astype(np.float32), device=dev)
nrm = torch.tensor(np.concatenate([rng.normal(size=(B, M, 3)), np.zeros((B, M, 1))], 2).astype(np.float32), device=dev)
lp = torch.tensor(np.asarray(base["lights"]["pos"], dtype=np.float32), device=dev).repeat(B, 1, 1)
lp[:, 0, :3] = torch.tensor(rng.uniform(-6, 6, (B, 3)).astype(np.float32), device=dev)
ov = [{"disk.pos": pos[v], "disk.normal": nrm[v], "lights.pos": lp[v]} for v in range(B)]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


t_batch = timed(lambda: render_views(base, cams, device=dev, overrides=ov, want_nearest=False))
t_loop = timed(lambda: [render({**base, "camera": cams[v], "lights": dict(base["lights"], pos=lp[v]),
                                "objects": {"disk": dict(base["objects"]["disk"], pos=pos[v], normal=nrm[v])}}, device=dev)
                        for v in range(B)], n=3)
print(f"GAN-shaped batch, {B} views x {R}x{R}, {M} splats each, different per view: render_views {1e3 * t_batch:.2f} ms "
      f"= {B / t_batch:.0f} views/s; one render() per view {1e3 * t_loop:.2f} ms = {B / t_loop:.0f} views/s")

