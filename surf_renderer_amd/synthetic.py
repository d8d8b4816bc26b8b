"""Synthetic and recipe scenes used by the benchmark, the tests and the golden generator.

``disk_cloud_scene`` is the headline workload of BASELINE.json (100 000 disk splats at 2048 x 2048,
SURVEY.md section 8d): a seeded cloud of small discs in [-1,1]^3 seen from (0,0,4).  It is modelled on how
the reference builds random splat scenes (diffrend/splats.py:243-275 -- random centres and normals
around the origin -- and diffrend/torch/full_diff_renderer_demo.py:39-65 -- constant radius, one
grey material), with homogeneous w fixed (points w=1, normals w=0).

The other builders restate the *data* of scenes the reference ships as Python literals
(numpy/renderer.py:299-358, torch/params.py:6-92) or as recipes in its demos, so that the GPU box,
which has no copy of the reference, can still render them.
"""
from __future__ import annotations

import copy
import os
from typing import Any, Dict, Optional

import numpy as np

from . import scene as sio

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")

_COLORS8 = [[0.0, 0.0, 0.0], [0.8, 0.1, 0.1], [0.2, 0.2, 0.2], [0.2, 0.8, 0.2],
            [0.2, 0.2, 0.8], [0.8, 0.2, 0.8], [0.8, 0.8, 0.2], [0.2, 0.8, 0.8]]
_ALBEDO6 = [[0.0, 0.0, 0.0], [0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [0.5, 0.5, 0.5],
            [0.9, 0.1, 0.1], [0.1, 0.6, 0.8]]


def _f32(a) -> np.ndarray:
    """float64 array holding fp32-representable values (what the device arrays will hold)."""
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def disk_cloud_scene(n: int = 100_000, width: int = 2048, height: int = 2048, radius: float = 0.02,
                     seed: int = 20240) -> Dict[str, Any]:
    """BASELINE config 5.  All draws come from one ``RandomState(seed)`` in a fixed order (centres,
    then normals) so every rank of a multi-GPU run builds the identical scene."""
    rng = np.random.RandomState(seed)
    pos = rng.uniform(-1.0, 1.0, size=(n, 3))
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[nrm[:, 2] < 0] *= -1.0                      # face the camera half-space (+z)
    return {
        "camera": {"proj_type": "perspective", "viewport": [0, 0, int(width), int(height)],
                   "fovy": float(np.deg2rad(45.0)), "focal_length": 1.0,
                   "eye": [0.0, 0.0, 4.0, 1.0], "up": [0.0, 1.0, 0.0, 0.0], "at": [0.0, 0.0, 0.0, 1.0],
                   "near": 0.1, "far": 1000.0},
        "lights": {"pos": _f32([[10, 0, 0, 1], [-10, 0, 0, 1], [0, 10, 0, 1], [0, 0, 10, 1]]),
                   "color_idx": np.array([1, 2, 3, 1], dtype=np.int64)},
        "colors": _f32([[0, 0, 0], [.8, .1, .1], [.2, .8, .2], [.2, .2, .8]]),
        "materials": {"albedo": _f32([[0.6, 0.6, 0.6]])},
        "objects": {"disk": {
            "pos": _f32(np.concatenate([pos, np.ones((n, 1))], axis=1)),
            "normal": _f32(np.concatenate([nrm, np.zeros((n, 1))], axis=1)),
            "radius": _f32(np.full(n, radius)),
            "material_idx": np.zeros(n, dtype=np.int64)}},
        "tonemap": {"type": "gamma", "gamma": 0.8},
    }


def demo_scene(width: int = 320, height: int = 240, with_planes: bool = False) -> Dict[str, Any]:
    """The mixed scene the numpy backend renders when run as a script (values from
    numpy/renderer.py:299-358): 3 discs, 2 spheres, 2 triangles, 2 lights, eye (0,1,10) looking at
    the origin with up = +y (not orthogonal to the view direction, so quirk Q1 is active).
    ``with_planes`` adds a floor and a back wall plus two more lights -> all four primitive types
    (SURVEY golden G2; the 'mixed' scene of BASELINE config 3b)."""
    sc: Dict[str, Any] = {
        "camera": {"viewport": [0, 0, int(width), int(height)], "fovy": float(np.deg2rad(90.0)),
                   "focal_length": 1.0, "eye": [0.0, 1.0, 10.0, 1.0], "up": [0.0, 1.0, 0.0, 0.0],
                   "at": [0.0, 0.0, 0.0, 1.0], "near": 1.0, "far": 1000.0},
        "lights": {"pos": _f32([[20, 20, 20, 1], [-15, 3, 15, 1]]), "color_idx": np.array([2, 1]),
                   "attenuation": _f32([[0, 1, 0], [0, 0, 1]])},
        "colors": _f32([[0, 0, 0], [.8, .1, .1], [.2, .2, .2]]),
        "materials": {"albedo": _f32(_ALBEDO6[:4] + [[.9, .1, .1], [.1, .1, .8]])},
        "objects": {
            "disk": {"normal": _f32([[0, 0, 1, 0], [0, 1, 0, 0], [-1, -1, 1, 0]]),
                     "pos": _f32([[0, -1, 3, 1], [0, -1, 0, 1], [10, 5, -5, 1]]),
                     "radius": _f32([4, 7, 4]), "material_idx": np.array([4, 3, 5])},
            "sphere": {"pos": _f32([[-8, 4, -8, 1], [10, 0, -4, 1]]), "radius": _f32([3, 2]),
                       "material_idx": np.array([3, 3])},
            "triangle": {"face": _f32([[[-20, -18, -10, 1], [10, -18, -10, 1], [-2.5, 18, -10, 1]],
                                       [[15, -18, -10, 1], [25, -18, -10, 1], [20, 18, -10, 1]]]),
                         "normal": _f32([[0, 0, 1, 0], [0, 0, 1, 0]]),
                         "material_idx": np.array([5, 4])},
        },
        "tonemap": {"type": "gamma", "gamma": 0.8},
    }
    if with_planes:
        planes = {"pos": _f32([[0, -9, 0, 1], [0, 0, -25, 1]]),
                  "normal": _f32([[0, 1, 0, 0], [0.1, 0, 2, 0]]),
                  "material_idx": np.array([2, 1])}
        sc["objects"] = {"plane": planes, **sc["objects"]}
        sc["lights"] = {"pos": _f32([[20, 20, 20, 1], [-15, 3, 15, 1], [0, 30, 5, 1], [5, 2, 30, 1]]),
                        "color_idx": np.array([2, 1, 3, 4])}
        sc["colors"] = _f32(_COLORS8[:5])
    return sc


def splat_basic_scene(width: int = 320, height: int = 240) -> Dict[str, Any]:
    """The 'starter scene for rendering splats' every torch demo deep-copies (values from
    torch/params.py:6-92): 3 discs, 7 lights, gamma 0.8, near 0.1."""
    return {
        "camera": {"proj_type": "perspective", "viewport": [0, 0, int(width), int(height)],
                   "fovy": float(np.deg2rad(90.0)), "focal_length": 1.0, "eye": [0.0, 1.0, 10.0, 1.0],
                   "up": [0.0, 1.0, 0.0, 0.0], "at": [0.0, 0.0, 0.0, 1.0], "near": 0.1, "far": 1000.0},
        "lights": {"pos": _f32([[10, 0, 0, 1], [-10, 0, 0, 1], [0, 10, 0, 1], [0, -10, 0, 1],
                                [0, 0, 10, 1], [0, 0, -10, 1], [20, 20, 20, 1]]),
                   "color_idx": np.array([1, 3, 4, 5, 6, 7, 1]),
                   "attenuation": _f32([[1, 0, 0]] * 7), "ambient": _f32([.01, .01, .01])},
        "colors": _f32(_COLORS8),
        "materials": {"albedo": _f32(_ALBEDO6), "coeffs": _f32([[1, 0, 0]] * 6)},
        "objects": {"disk": {"normal": _f32([[0, 0, 1, 0], [0, 1, 0, 0], [-1, -1, 1, 0]]),
                             "pos": _f32([[0, -1, 3, 1], [0, -1, 0, 1], [10, 5, -5, 1]]),
                             "radius": _f32([4, 7, 4]), "material_idx": np.array([4, 3, 5])}},
        "tonemap": {"type": "gamma", "gamma": 0.8},
    }


def bunny_splat_scene(width: int = 512, height: int = 512, path: Optional[str] = None) -> Dict[str, Any]:
    """BASELINE config 2: ``data/bunny.splat`` as discs with the reference's scalability recipe
    (torch/test_optimization.py:626-655): v <- (v - mean) / (max - min) over *all* coordinates,
    radius x 2, one 0.6 grey material, fovy 5 degrees, focal length 2."""
    spl = sio.load_splat(path or os.path.join(ASSETS, "data", "bunny.splat"))
    v = spl["v"]
    v = (v - np.mean(v, axis=0)) / (v.max() - v.min())
    m = v.shape[0]
    sc = splat_basic_scene(width, height)
    sc["camera"]["fovy"] = float(np.deg2rad(5.0))
    sc["camera"]["focal_length"] = 2.0
    sc["objects"] = {"disk": {"pos": _f32(np.concatenate([v, np.ones((m, 1))], axis=1)),
                              "normal": _f32(np.concatenate([spl["vn"], np.zeros((m, 1))], axis=1)),
                              "radius": _f32(spl["r"].ravel() * 2),
                              "material_idx": np.zeros(m, dtype=np.int64)}}
    sc["materials"] = {"albedo": _f32([[0.6, 0.6, 0.6]])}
    return sc


def bunny_mesh_scene(width: int = 1024, height: int = 1024, path: Optional[str] = None) -> Dict[str, Any]:
    """BASELINE config 4: ``data/bunny.obj`` as a triangle batch, vertices normalised so the widest
    axis spans 1 (torch/full_diff_renderer_demo.py:29-34, use_mesh branch :46-58), one 0.6 grey
    material, camera pulled in to fovy 8 degrees so the mesh fills the frame."""
    obj = sio.load_obj(path or os.path.join(ASSETS, "data", "bunny.obj"))
    v = obj["v"]
    v = (v - np.mean(v, axis=0)) / np.max(np.max(v, axis=0) - np.min(v, axis=0))
    spec = sio.obj_to_triangle_spec({"v": v, "f": obj["f"]})
    m = spec["face"].shape[0]
    sc = splat_basic_scene(width, height)
    sc["camera"]["fovy"] = float(np.deg2rad(8.0))
    sc["objects"] = {"triangle": {"face": _f32(spec["face"]), "normal": _f32(spec["normal"]),
                                  "material_idx": np.zeros(m, dtype=np.int64)}}
    sc["materials"] = {"albedo": _f32([[0.6, 0.6, 0.6]])}
    return sc


def json_scene(name: str, width: Optional[int] = None, height: Optional[int] = None) -> Dict[str, Any]:
    """One of the shipped JSON scenes under assets/scenes (BASELINE configs 1 and 3a), expanded, with
    fp32-representable geometry and an optional viewport override."""
    vp = (width, height) if width is not None else None
    sc = sio.load_scene(os.path.join(ASSETS, "scenes", name), viewport=vp)
    for grp in sc["objects"].values():
        for key in ("face", "normal", "pos", "radius"):
            if key in grp:
                grp[key] = _f32(grp[key])
    return sc


def clone(scene: Dict[str, Any]) -> Dict[str, Any]:
    return copy.deepcopy(scene)
