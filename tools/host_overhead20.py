#!/usr/bin/env python3
"""Where a 20-frame timed region spends its wall time on the host (measurement): submission loop, final wait, and the
GPU-side span between the first and the last kernel as CUDA events see it."""
import sys, time, json
import torch
sys.path.insert(0, ".")
from surf_renderer_amd import renderer, synthetic
from surf_renderer_amd.pipeline import FramePipeline

scene = synthetic.disk_cloud_scene()
dev = torch.device("cuda", 0)
buf = renderer.flatten_scene(scene, device=dev)
cam = renderer.camera_struct(scene["camera"])
pipe = FramePipeline(buf, cam, n_inflight=3, graphs=True)
for _ in range(600):
    pipe.submit()
pipe.sync(); torch.cuda.synchronize()
res = []
for rep in range(8):
    e0 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e1 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        if i < 3:
            e0[i].record(pipe.streams[i])
        pipe.submit()
        if i >= 17:
            e1[i - 17].record(pipe.streams[i % 3])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    span = max(e0[0].elapsed_time(b) for b in e1)
    res.append({"submit_us": round(1e6 * (t1 - t0), 1), "wait_us": round(1e6 * (t2 - t1), 1), "total_us": round(1e6 * (t2 - t0), 1),
                "gpu_span_us": round(1e3 * span, 1)})
for r in res:
    print(json.dumps(r))
