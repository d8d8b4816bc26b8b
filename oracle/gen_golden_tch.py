#!/usr/bin/env python3
"""Generate tests/golden/t*.npz by running the UNMODIFIED reference torch backend (forward, CPU, float32).

Test infrastructure; runs only in the build container.  Cases exercise the superset shading model of
diffrend/torch/renderer.py:82-125,136-355: attenuation, ambient, specular, per-light relu, double_sided, use_quartic,
orthonormal camera, far+1 background.  Stored: the scene, the keyword arguments, and image / depth / nearest / normal /
pos as the reference returns them.
"""
import contextlib
import copy
import io
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SRH_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

import torch  # noqa: E402

from oracle.golden_io import pack_scene  # noqa: E402
from surf_renderer_amd import synthetic  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    import diffrend.torch.renderer as ref_tch  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def to_torch(sc):
    t = {"camera": dict(sc["camera"], proj_type=sc["camera"].get("proj_type", "perspective"))}
    for k in ("eye", "at", "up"):
        t["camera"][k] = torch.tensor(np.asarray(sc["camera"][k], dtype=np.float32))
    t["lights"] = {"pos": torch.tensor(np.asarray(sc["lights"]["pos"], dtype=np.float32)),
                   "color_idx": torch.tensor(np.asarray(sc["lights"]["color_idx"])),
                   "attenuation": torch.tensor(np.asarray(sc["lights"]["attenuation"], dtype=np.float32)),
                   "ambient": torch.tensor(np.asarray(sc["lights"]["ambient"], dtype=np.float32))}
    t["colors"] = torch.tensor(np.asarray(sc["colors"], dtype=np.float32))
    t["materials"] = {"albedo": torch.tensor(np.asarray(sc["materials"]["albedo"], dtype=np.float32)),
                      "coeffs": torch.tensor(np.asarray(sc["materials"]["coeffs"], dtype=np.float32))}
    t["objects"] = {}
    for kind, grp in sc["objects"].items():
        t["objects"][kind] = {k: torch.tensor(np.asarray(v, dtype=np.float32)) if k != "material_idx"
                              else torch.tensor(np.asarray(v)) for k, v in grp.items()}
    if "tonemap" in sc:
        t["tonemap"] = {"type": "gamma", "gamma": torch.tensor([float(np.ravel(sc["tonemap"]["gamma"])[0])])}
    return t


def emit(name, sc, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        res = ref_tch.render(to_torch(sc), tiled=False, shadow=False, **kw)
    flat = pack_scene(sc)
    H, W = res["depth"].shape
    flat["out/image"] = res["image"].numpy()
    flat["out/depth"] = res["depth"].numpy()
    flat["out/nearest"] = res["nearest"].numpy().astype(np.int64)
    flat["out/normal"] = res["normal"].numpy()
    flat["out/pos"] = res["pos"].reshape(H, W, 3).numpy()
    flat["kwargs"] = np.asarray(json.dumps(kw))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    hit = (flat["out/depth"] <= sc["camera"]["far"]).mean()
    print(f"{name:34s} {flat['out/depth'].shape} hit {hit:6.1%} image max {flat['out/image'].max():.4f}")


def main():
    os.makedirs(OUT, exist_ok=True)
    # t1: the torch demos' starter scene (torch/params.py:6-92) with a mix of attenuation laws
    t1 = synthetic.splat_basic_scene(64, 48)
    t1["lights"]["attenuation"] = f32([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0.1, 0.01], [0.5, 0, 0.02], [1, 0, 0], [0, 0.05, 0]])
    emit("t1_splat_basic_64x48", t1)
    emit("t1_splat_basic_64x48_quartic_ds", t1, use_quartic=True, double_sided=True)

    # t2: all four primitive types, specular materials, ambient light, up not orthogonal to the view direction
    t2 = synthetic.demo_scene(64, 48, with_planes=True)
    t2["camera"]["near"] = 0.5
    t2["lights"]["attenuation"] = f32([[1, 0, 0], [0.2, 0.05, 0], [1, 0, 0.001], [0.7, 0.02, 0.0005]])
    t2["lights"]["ambient"] = f32([0.02, 0.015, 0.01])
    t2["materials"]["coeffs"] = f32([[1, 0, 0], [0.8, 0.2, 4], [0.6, 0.4, 16], [0.9, 0.1, 2], [0.5, 0.5, 8], [0.7, 0.3, 32]])
    emit("t2_mixed_specular_64x48", t2)
    emit("t2_mixed_specular_64x48_ds", t2, double_sided=True)

    # t3: disc cloud as the GAN renders it (double sided), normals facing both ways
    t3 = synthetic.disk_cloud_scene(1500, 64, 64, radius=0.07, seed=21)
    nrm = t3["objects"]["disk"]["normal"].copy()
    nrm[::2] *= -1.0
    t3["objects"]["disk"]["normal"] = nrm
    t3["lights"]["attenuation"] = f32([[1, 0, 0]] * 4)
    t3["lights"]["ambient"] = f32([0.01, 0.01, 0.01])
    t3["materials"]["coeffs"] = f32([[1, 0, 0]])
    emit("t3_disk_cloud_64x64_ds", t3, double_sided=True)
    emit("t3_disk_cloud_64x64", t3)

    # t4: orthographic projection (torch/utils.py:461-468; the reference's ortho branch only works while the image
    # fits one 4096-pixel tile), the mixed scene and a disc cloud
    t4 = synthetic.demo_scene(64, 48, with_planes=True)
    t4["camera"].update(proj_type="ortho", near=0.5, fovy=float(np.deg2rad(100.0)), focal_length=4.0)
    t4["lights"]["attenuation"] = f32([[1, 0, 0], [0.2, 0.05, 0], [1, 0, 0.001], [0.7, 0.02, 0.0005]])
    t4["lights"]["ambient"] = f32([0.02, 0.015, 0.01])
    t4["materials"]["coeffs"] = f32([[1, 0, 0], [0.8, 0.2, 4], [0.6, 0.4, 16], [0.9, 0.1, 2], [0.5, 0.5, 8], [0.7, 0.3, 32]])
    emit("t4_mixed_ortho_64x48", t4)
    t5 = synthetic.disk_cloud_scene(1500, 64, 64, radius=0.07, seed=22)
    t5["camera"].update(proj_type="orthographic", fovy=float(np.deg2rad(60.0)), focal_length=2.0)
    t5["lights"]["attenuation"] = f32([[1, 0, 0]] * 4)
    t5["lights"]["ambient"] = f32([0.01, 0.01, 0.01])
    t5["materials"]["coeffs"] = f32([[0.9, 0.1, 3.0]])
    emit("t5_disk_cloud_ortho_64x64_ds", t5, double_sided=True)


if __name__ == "__main__":
    main()
