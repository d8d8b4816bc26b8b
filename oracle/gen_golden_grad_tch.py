#!/usr/bin/env python3
"""Generate tests/golden/g10_torch_autograd_phong*.npz by running the UNMODIFIED reference torch backend under autograd
with its full shading model (attenuation, specular coefficients, ambient, per-light relu, optional double_sided /
use_quartic): the gradients that `render(scene, shading='torch')` must reproduce.

Test infrastructure; runs only in the build container (needs /root/reference).  Stored: the scene, the upstream
gradients, and d loss / d input for every differentiable input as torch autograd computes it through
diffrend/torch/renderer.py:82-125,136-355 (float32, as the reference computes).
"""
import contextlib
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

with contextlib.redirect_stdout(io.StringIO()):
    import diffrend.torch.renderer as ref_tch  # noqa: E402


def f32(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def build_scene():
    return {
        "camera": {"viewport": [0, 0, 48, 36], "fovy": float(np.deg2rad(60.0)), "focal_length": 1.0,
                   "eye": [0.3, 1.0, 10.0, 1.0], "up": [0.0, 1.0, 0.0, 0.0], "at": [0.0, 0.0, 0.0, 1.0],
                   "near": 0.1, "far": 100.0},
        "lights": {"pos": f32([[2.5, 3.8, 9.5, 1], [-3.9, 1.3, 8.2, 1], [0.2, -4.7, 7.0, 1]]),
                   "color_idx": np.array([1, 2, 3]),
                   "attenuation": f32([[1, 0, 0], [0.4, 0.05, 0.002], [0.8, 0, 0.004]]),
                   "ambient": f32([0.03, 0.02, 0.025])},
        "colors": f32([[0, 0, 0], [.8, .3, .2], [.2, .7, .3], [.3, .3, .9]]),
        "materials": {"albedo": f32([[.5, .5, .5], [.9, .4, .2], [.2, .8, .6]]),
                      "coeffs": f32([[1.0, 0.0, 0.0], [0.7, 0.3, 6.0], [0.5, 0.5, 12.0]])},
        "objects": {
            "plane": {"pos": f32([[0, 0, -6, 1]]), "normal": f32([[0.1, -0.05, 1.5, 0]]), "material_idx": np.array([0])},
            "disk": {"pos": f32([[-2, 1, 1, 1], [1.5, -1, 2, 1], [0.5, 2, -1, 1]]),
                     "normal": f32([[0.2, 0.1, 1, 0], [-0.3, 0.2, 0.9, 0], [0, -0.4, 2, 0]]),
                     "radius": f32([1.6, 1.4, 2.2]), "material_idx": np.array([1, 2, 1])},
            "sphere": {"pos": f32([[-3.0, -2.0, 0.5, 1], [3.2, 1.8, -0.5, 1]]), "radius": f32([1.3, 1.1]),
                       "material_idx": np.array([2, 1])},
            "triangle": {"face": f32([[[-4, -3, -2, 1], [0, -3.5, -2.5, 1], [-2.5, 1, -1.5, 1]],
                                      [[1, 0, -3, 1], [4.5, -1, -3.5, 1], [3, 3, -2.5, 1]]]),
                         "normal": f32([[-0.05, 0.15, 1, 0], [0.1, 0.05, 1, 0]]), "material_idx": np.array([2, 0])},
        },
        "tonemap": {"type": "gamma", "gamma": 0.8},
    }


def emit(name, camera=None, **kw):
    sc = build_scene()
    if camera:
        sc["camera"].update(camera)
    rng = np.random.RandomState(7)
    H, W = 36, 48
    g_img = f32(rng.uniform(-1, 1, size=(H, W, 3)))
    g_dep = f32(rng.uniform(-1, 1, size=(H, W)))

    def leaf(a):
        return torch.tensor(np.asarray(a, dtype=np.float32), requires_grad=True)

    leaves = {}
    tsc = {"camera": dict(sc["camera"], proj_type=sc["camera"].get("proj_type", "perspective")),
           "tonemap": {"type": "gamma", "gamma": torch.tensor([0.8])}}
    for k in ("eye", "at", "up"):
        tsc["camera"][k] = torch.tensor(sc["camera"][k], dtype=torch.float32)
    tsc["lights"] = {"pos": leaf(sc["lights"]["pos"]), "color_idx": torch.tensor(sc["lights"]["color_idx"]),
                     "attenuation": leaf(sc["lights"]["attenuation"]), "ambient": leaf(sc["lights"]["ambient"])}
    leaves["lights.pos"] = tsc["lights"]["pos"]
    leaves["lights.attenuation"] = tsc["lights"]["attenuation"]
    leaves["lights.ambient"] = tsc["lights"]["ambient"]
    tsc["colors"] = leaf(sc["colors"]); leaves["colors"] = tsc["colors"]
    tsc["materials"] = {"albedo": leaf(sc["materials"]["albedo"]), "coeffs": leaf(sc["materials"]["coeffs"])}
    leaves["materials.albedo"] = tsc["materials"]["albedo"]
    leaves["materials.coeffs"] = tsc["materials"]["coeffs"]
    tsc["objects"] = {}
    for kind, grp in sc["objects"].items():
        tg = {"material_idx": torch.tensor(grp["material_idx"])}
        for nm, val in grp.items():
            if nm != "material_idx":
                tg[nm] = leaf(val)
                leaves[f"{kind}.{nm}"] = tg[nm]
        tsc["objects"][kind] = tg

    with contextlib.redirect_stdout(io.StringIO()):
        res = ref_tch.render(tsc, tiled=False, shadow=False, **kw)
    image, depth = res["image"], res["depth"]
    hit = depth <= sc["camera"]["far"]
    loss = torch.sum(image * torch.tensor(g_img, dtype=torch.float32)) + \
        torch.sum(torch.where(hit, depth * torch.tensor(g_dep, dtype=torch.float32), torch.zeros_like(depth)))
    loss.backward()

    out = pack_scene(sc)
    out["grad_in/image"] = g_img
    out["grad_in/depth"] = g_dep
    out["ref/image"] = image.detach().numpy()
    out["ref/depth"] = depth.detach().numpy()
    out["ref/nearest"] = res["nearest"].detach().numpy().astype(np.int64)
    out["kwargs"] = np.asarray(json.dumps(kw))
    for k, v in leaves.items():
        out["grad/" + k] = v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape), dtype=np.float32)
        print(f"{k:22s} |grad| max {np.abs(out['grad/' + k]).max():.4g}")
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("hit fraction", float(hit.float().mean()), "->", path)


if __name__ == "__main__":
    only = sys.argv[1:]
    if not only or "g10" in only:
        emit("g10_torch_autograd_phong")
        emit("g10_torch_autograd_phong_ds_quartic", double_sided=True, use_quartic=True)
    if not only or "g11" in only:
        # orthographic projection (torch/utils.py:461-468): per-ray origins on the image plane, one direction; the
        # reference's ortho branch works while the image fits one 4096-pixel tile (48 x 36 does)
        emit("g11_torch_autograd_ortho", camera={"proj_type": "ortho", "fovy": float(np.deg2rad(100.0)), "focal_length": 4.0})
