#!/usr/bin/env python3
"""Secondary measurements: BASELINE configs 1-4 (forward frames/s, and forward+backward for config 4) on one GPU.
Prints one JSON line per config.  Not the headline benchmark (that is bench.py = config 5)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from surf_renderer_amd import renderer, synthetic


def timed(fn, steps=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def forward_rate(name, scene, steps=50):
    buf = renderer.flatten_scene(scene, "cuda:0")
    cam = renderer.camera_struct(scene["camera"])
    w, h = renderer.frame_size(cam)
    out = (torch.empty((h, w, 3), device="cuda:0"), torch.empty((h, w), device="cuda:0"),
           torch.empty((h, w), dtype=torch.int32, device="cuda:0"))
    dt = timed(lambda: renderer.render_buffers(buf, cam, out=out), steps)
    # the same frames through the pipeline bench.py uses: three in flight on three streams, one hipGraph each
    from surf_renderer_amd.pipeline import FramePipeline
    pipe = FramePipeline(buf, cam, n_inflight=3, graphs=True)
    dp = timed(pipe.submit, 20 * steps, 50)
    pipe.verify()
    print(json.dumps({"config": name, "width": w, "height": h, "prims": buf.total, "ms_per_frame": 1e3 * dt,
                      "frames_per_s": 1 / dt, "gtests_per_s": buf.total * w * h / dt / 1e9,
                      "ms_per_frame_pipelined": 1e3 * dp, "frames_per_s_pipelined": 1 / dp,
                      "what": "ms_per_frame: one eager srh_render_fwd call after another on one stream (a frame's latency); "
                              "pipelined: three frames in flight, graph replays, checked bit for bit"}), flush=True)


def main():
    forward_rate("1: scenes/basic.json 128x128", synthetic.json_scene("basic.json", 128, 128))
    forward_rate("2: bunny.splat discs 512x512", synthetic.bunny_splat_scene(512, 512))
    forward_rate("3a: halfbox_sphere_cube.json 1024x1024", synthetic.json_scene("halfbox_sphere_cube.json", 1024, 1024))
    forward_rate("3b: mixed plane/sphere/disc/triangle, 4 lights, 1024x1024", synthetic.demo_scene(1024, 1024, with_planes=True))
    mesh = synthetic.bunny_mesh_scene(1024, 1024)
    forward_rate("4: bunny.obj triangles 1024x1024 (forward)", mesh)
    # forward + analytic backward through render(): d image / d vertex 0, d image / d normal
    tri = mesh["objects"]["triangle"]
    face = torch.tensor(np.asarray(tri["face"], dtype=np.float32), device="cuda:0", requires_grad=True)
    normal = torch.tensor(np.asarray(tri["normal"], dtype=np.float32), device="cuda:0", requires_grad=True)
    mesh["objects"]["triangle"] = {"face": face, "normal": normal,
                                   "material_idx": torch.tensor(np.asarray(tri["material_idx"]), device="cuda:0")}
    for key in ("pos",):
        mesh["lights"][key] = torch.tensor(np.asarray(mesh["lights"][key], dtype=np.float32), device="cuda:0")
    mesh["colors"] = torch.tensor(np.asarray(mesh["colors"], dtype=np.float32), device="cuda:0")
    mesh["materials"]["albedo"] = torch.tensor(np.asarray(mesh["materials"]["albedo"], dtype=np.float32), device="cuda:0")

    def fwd_bwd():
        face.grad = None
        normal.grad = None
        res = renderer.render(mesh, device="cuda:0", validate=False)
        res["image"].sum().backward()

    dt = timed(fwd_bwd, steps=20, warmup=3)
    print(json.dumps({"config": "4: bunny.obj 1024x1024 forward + backward through render() (Python flatten included)",
                      "ms_per_iteration": 1e3 * dt, "iterations_per_s": 1 / dt}), flush=True)


def batched_views():
    """f4: 64 views of bunny.splat at 128x128 (the GAN's regime: many small renders of one scene)."""
    scene = synthetic.bunny_splat_scene(128, 128)
    rng = np.random.RandomState(0)
    cams = []
    for _ in range(64):
        eye = rng.normal(size=3)
        eye = 10.0 * eye / np.linalg.norm(eye)
        cams.append(dict(scene["camera"], eye=[float(eye[0]), float(eye[1]), float(eye[2]), 1.0]))
    for streams in (1, 8):
        dt = timed(lambda: renderer.render_views(scene, cams, device="cuda:0", streams=streams, batch=0), steps=10, warmup=2)
        print(json.dumps({"config": f"f4: 64 views x bunny.splat 128x128, one call per view on {streams} stream(s), "
                                    "scene upload included", "ms_per_batch": 1e3 * dt, "views_per_s": 64 / dt}), flush=True)
    for n in (64, 1024):
        many = (cams * ((n + 63) // 64))[:n]
        dt = timed(lambda: renderer.render_views(scene, many, device="cuda:0"), steps=10, warmup=2)
        print(json.dumps({"config": f"f4: {n} views x bunny.splat 128x128, srh_render_views (view = grid dimension), "
                                    "scene upload included", "ms_per_batch": 1e3 * dt, "views_per_s": n / dt}), flush=True)


if __name__ == "__main__":
    main()
    batched_views()
