#!/usr/bin/env python3
"""Backward and shadow-pass timings on one GPU (run it plain for the JSON lines, or under
`rocprofv3 --kernel-trace --stats` for per-kernel durations).  Secondary measurements, not the headline bench.

  cases   bwd_mesh      BASELINE config 4: bunny.obj triangles at 1024^2, forward + backward through render()
          bwd_mesh_tch  the same under the torch backend's Phong semantics
          bwd_plane     one plane filling a 2048^2 frame (+ 3 discs): every pixel's gradient lands on one primitive
          bwd_discs     100k discs at 2048^2 (config 5's scene)
          shadow_*      render(shading='torch', shadow=True): primary pass + shadow pass
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from surf_renderer_amd import renderer, synthetic

DEV = "cuda:0"


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def leaves(scene, kind, names):
    grp = scene["objects"][kind]
    out = {}
    for n in names:
        out[n] = torch.tensor(np.asarray(grp[n], dtype=np.float32), device=DEV, requires_grad=True)
    grp = dict(grp, **out)
    grp["material_idx"] = torch.tensor(np.asarray(grp["material_idx"]), device=DEV)
    scene["objects"][kind] = grp
    scene["lights"] = dict(scene["lights"], pos=torch.tensor(np.asarray(scene["lights"]["pos"], dtype=np.float32), device=DEV))
    scene["colors"] = torch.tensor(np.asarray(scene["colors"], dtype=np.float32), device=DEV)
    scene["materials"] = dict(scene["materials"],
                              albedo=torch.tensor(np.asarray(scene["materials"]["albedo"], dtype=np.float32), device=DEV))
    return list(out.values())


def plane_scene(w, h):
    sc = synthetic.demo_scene(w, h, with_planes=True)
    objs = sc["objects"]
    sc["objects"] = {"plane": {"pos": np.array([[0.0, 0.0, -2.0, 1.0]], dtype=np.float32),
                               "normal": np.array([[0.0, 0.1, 1.0, 0.0]], dtype=np.float32),
                               "material_idx": np.array([0], dtype=np.int32)},
                     "disk": objs["disk"]}
    return sc


def fwd_bwd(scene, params, steps, **kw):
    def it():
        for p in params:
            p.grad = None
        res = renderer.render(scene, device=DEV, validate=False, **kw)
        (res["image"].sum() + res["depth"].clamp(max=100.0).sum()).backward()
    return timed(it, steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="bwd_mesh,bwd_mesh_tch,bwd_plane,bwd_discs,shadow_mesh,shadow_discs")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    for case in args.cases.split(","):
        if case == "bwd_mesh" or case == "bwd_mesh_tch":
            sc = synthetic.bunny_mesh_scene(1024, 1024)
            ps = leaves(sc, "triangle", ("face", "normal"))
            kw = {"shading": "torch"} if case.endswith("tch") else {}
            dt = fwd_bwd(sc, ps, args.steps, **kw)
            out = {"case": case, "what": "bunny.obj 1024x1024 forward + backward through render()", "ms_per_iteration": 1e3 * dt,
                   "iterations_per_s": 1 / dt}
        elif case in ("bwd_mesh_resident", "bwd_mesh_resident_tch"):
            sc = synthetic.bunny_mesh_scene(1024, 1024)
            ps = leaves(sc, "triangle", ("face", "normal"))
            rs = renderer.ResidentScene(sc, device=DEV, shading="torch" if case.endswith("tch") else "numpy", validate=False)

            def it():
                for p_ in ps:
                    p_.grad = None
                res = rs.render()
                (res["image"].sum() + res["depth"].clamp(max=100.0).sum()).backward()
            dt = timed(it, args.steps * 5, 5)
            out = {"case": case, "what": "bunny.obj 1024x1024 forward + backward, ResidentScene (scene flattened once)",
                   "ms_per_iteration": 1e3 * dt, "iterations_per_s": 1 / dt}
        elif case in ("bwd_mesh_captured", "bwd_mesh_captured_tch"):
            sc = synthetic.bunny_mesh_scene(1024, 1024)
            ps = leaves(sc, "triangle", ("face", "normal"))
            rs = renderer.ResidentScene(sc, device=DEV, shading="torch" if case.endswith("tch") else "numpy", validate=False)
            step = rs.capture_step(lambda res: res["image"].sum() + res["depth"].clamp(max=100.0).sum())
            dt = timed(step.replay, args.steps * 5, 5)
            out = {"case": case, "what": "bunny.obj 1024x1024 forward + backward, ResidentScene.capture_step (one hipGraph "
                                         "per iteration)", "ms_per_iteration": 1e3 * dt, "iterations_per_s": 1 / dt}
        elif case == "bwd_plane":
            sc = plane_scene(2048, 2048)
            ps = leaves(sc, "plane", ("pos", "normal"))
            dt = fwd_bwd(sc, ps, args.steps)
            out = {"case": case, "what": "one plane filling 2048x2048 + 3 discs, forward + backward", "ms_per_iteration": 1e3 * dt}
        elif case == "bwd_discs":
            sc = synthetic.disk_cloud_scene(100_000, 2048, 2048)
            ps = leaves(sc, "disk", ("pos", "normal"))
            dt = fwd_bwd(sc, ps, args.steps)
            out = {"case": case, "what": "100k discs 2048x2048, forward + backward", "ms_per_iteration": 1e3 * dt}
        elif case == "shadow_mesh":
            sc = synthetic.bunny_mesh_scene(512, 512)
            dt = timed(lambda: renderer.render(sc, device=DEV, validate=False, shading="torch", shadow=True), max(2, args.steps // 5), 1)
            out = {"case": case, "what": "bunny.obj (4968 triangles) 512x512, shading='torch', shadow=True", "ms_per_frame": 1e3 * dt}
        elif case == "shadow_discs":
            sc = synthetic.disk_cloud_scene(20_000, 512, 512)
            dt = timed(lambda: renderer.render(sc, device=DEV, validate=False, shading="torch", shadow=True), max(2, args.steps // 5), 1)
            out = {"case": case, "what": "20k discs 512x512, shading='torch', shadow=True", "ms_per_frame": 1e3 * dt}
        elif case in ("shadow_cfg5", "shadow_mesh_allpairs", "shadow_discs_allpairs"):
            # the shadow pass alone over a resident frame: light-space bins (default) or the all-pairs loop
            sc = {"shadow_cfg5": lambda: synthetic.disk_cloud_scene(100_000, 2048, 2048),
                  "shadow_mesh_allpairs": lambda: synthetic.bunny_mesh_scene(512, 512),
                  "shadow_discs_allpairs": lambda: synthetic.disk_cloud_scene(20_000, 512, 512)}[case]()
            buf = renderer.flatten_scene(sc, DEV)
            cam = renderer.camera_struct(sc["camera"], "torch")
            image, depth, nearest = renderer.render_buffers(buf, cam, shading="torch")
            out = {"case": case, "what": f"shadow pass alone, {buf.total} primitives, {image.shape[1]}x{image.shape[0]}, "
                                         f"{buf.lights.n_lights} lights"}
            for label, ap in (("binned_ms", False), ("all_pairs_ms", True)):
                if ap and case == "shadow_cfg5":
                    continue                       # ~1e12 fp64 pair tests: tens of seconds; priced by the smaller cases
                dt = timed(lambda: renderer.shadow_pass(buf, cam, None, image, depth, nearest, all_pairs=ap),
                           max(2, args.steps // 5), 1)
                out[label] = 1e3 * dt
        else:
            raise SystemExit(f"unknown case {case}")
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
