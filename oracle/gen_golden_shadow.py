#!/usr/bin/env python3
"""Generate tests/golden/s1*.npz: the reference torch backend's render(scene, shadow=True), forward, CPU, float32.

Test infrastructure; runs only in the build container.  The reference source is run unmodified, but its shadow branch
converts a mask with `.type(torch.cuda.FloatTensor)` (torch/renderer.py:311), which raises on a machine without a
GPU.  This script therefore aliases `torch.cuda.FloatTensor` to `torch.FloatTensor` *in its own process* before the
call -- the only deviation, and one that changes where the tensor lives, not a value in it.  Stored: the scene, the
keyword arguments, image / depth / nearest (the reference does not return the visibility itself; the image carries it:
a light that the reference finds blocked contributes nothing to the pixel).
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import torch  # noqa: E402

import gen_golden_tch as G  # noqa: E402  (imports the reference, to_torch(), f32())
from oracle.golden_io import pack_scene  # noqa: E402
from surf_renderer_amd import synthetic  # noqa: E402

torch.cuda.FloatTensor = torch.FloatTensor          # see the docstring


def emit(name, sc, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        lit = G.ref_tch.render(G.to_torch(sc), tiled=False, shadow=False, **kw)
        res = G.ref_tch.render(G.to_torch(sc), tiled=False, shadow=True, **kw)
    flat = pack_scene(sc)
    flat["out/image"] = res["image"].numpy()
    flat["out/depth"] = res["depth"].numpy()
    flat["out/nearest"] = res["nearest"].numpy().astype(np.int64)
    flat["kwargs"] = np.asarray(json.dumps(kw))
    np.savez_compressed(os.path.join(G.OUT, name + ".npz"), **flat)
    changed = (np.abs(res["image"].numpy() - lit["image"].numpy()).max(axis=-1) > 1e-6).mean()
    print(f"{name:34s} {flat['out/depth'].shape} pixels darkened by a shadow {changed:6.1%}")


def main():
    # s1a: the mixed scene of t2 (all four primitive types; spheres and triangles shadow the planes)
    a = synthetic.demo_scene(64, 48, with_planes=True)
    a["camera"]["near"] = 0.5
    a["lights"]["attenuation"] = G.f32([[1, 0, 0], [0.2, 0.05, 0], [1, 0, 0.001], [0.7, 0.02, 0.0005]])
    a["lights"]["ambient"] = G.f32([0.02, 0.015, 0.01])
    a["materials"]["coeffs"] = G.f32([[1, 0, 0], [0.8, 0.2, 4], [0.6, 0.4, 16], [0.9, 0.1, 2], [0.5, 0.5, 8], [0.7, 0.3, 32]])
    emit("s1a_mixed_shadow_64x48", a)
    emit("s1a_mixed_shadow_64x48_ds", a, double_sided=True)
    # s1b: a disc cloud shadowing itself
    b = synthetic.disk_cloud_scene(600, 64, 64, radius=0.12, seed=23)
    b["lights"]["attenuation"] = G.f32([[1, 0, 0]] * 4)
    b["lights"]["ambient"] = G.f32([0.01, 0.01, 0.01])
    b["materials"]["coeffs"] = G.f32([[0.9, 0.1, 3.0]])
    emit("s1b_disk_cloud_shadow_64x64_ds", b, double_sided=True)


if __name__ == "__main__":
    main()
