#!/usr/bin/env python3
"""Host time of one eager renderer.render_buffers call (tiny scene: the GPU work hides behind it), with a profile of
where it goes -- the multi-GPU bench's owner-weighted path and every eager caller pay this per slab."""
import cProfile, io, pstats, sys, time
import torch
sys.path.insert(0, ".")
from surf_renderer_amd import renderer, synthetic

scene = synthetic.json_scene("basic.json", 128, 128)
buf = renderer.flatten_scene(scene, "cuda:0")
cam = renderer.camera_struct(scene["camera"])
out = (torch.empty((128, 128, 3), device="cuda:0"), torch.empty((128, 128), device="cuda:0"), None)
f = lambda: renderer.render_buffers(buf, cam, out=out)   # noqa: E731
for _ in range(200):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000):
    f()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"render_buffers: {1e6 * (t1 - t0) / 2000:.1f} us of host time per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    f()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3000])
