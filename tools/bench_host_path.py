#!/usr/bin/env python3
"""What a caller of the reference's interface sees: render(scene) with ndarray leaves in host memory, i.e. scene
upload over PCIe on every call, and optionally the outputs copied back to host arrays the way the reference returns
them.  Workload = bench.py's (2048x2048, 100k discs).  These are the PCIe-inclusive rates DESIGN.md quotes beside
the resident-in-HBM headline; they are never bench.py's `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from surf_renderer_amd import render, renderer, synthetic


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    scene = synthetic.disk_cloud_scene()
    w, h = scene["camera"]["viewport"][2:]

    def to_host():
        res = render(scene, device="cuda:0")
        return {k: res[k].cpu() for k in ("image", "depth", "nearest")}

    def to_host_pinned():
        return render(scene, device="cuda:0").numpy()

    rows = [("render(scene): host scene in, device tensors out", timed(lambda: render(scene, device="cuda:0"), 100)),
            ("render(scene, validate=False): same without the host-side input checks",
             timed(lambda: render(scene, device="cuda:0", validate=False), 100)),
            ("render(scene) + outputs copied to host with .cpu() (pageable)", timed(to_host, 20)),
            ("render(scene).numpy(): outputs to host ndarrays through pinned memory", timed(to_host_pinned, 50))]
    buf = renderer.flatten_scene(scene, "cuda:0")
    cam = renderer.camera_struct(scene["camera"])
    out = (torch.empty((h, w, 3), device="cuda:0"), torch.empty((h, w), device="cuda:0"),
           torch.empty((h, w), dtype=torch.int32, device="cuda:0"))
    rows.append(("render_buffers(): scene and outputs resident, one C-ABI call per frame",
                 timed(lambda: renderer.render_buffers(buf, cam, out=out), 300)))
    for name, dt in rows:
        print(json.dumps({"path": name, "ms_per_frame": round(1e3 * dt, 4), "frames_per_s": round(1 / dt, 1)}), flush=True)


if __name__ == "__main__":
    main()
