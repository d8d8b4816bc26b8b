"""Scene and model IO for the hip backend.

The hip backend consumes the same scene dict / "diffrend 0.1" JSON files as the
reference backends.  This module holds the host-side loaders that turn files into the
*expanded* scene dict (the form ``render(scene)`` consumes) and the flattening step that
turns an expanded scene into contiguous fp32 device arrays in the reference's
concatenation order.

Reference behaviour reproduced here (read for behaviour, code is our own):
  * OBJ / OFF / .splat parsing ............ diffrend/model.py:90-199
  * face normals (zero-area -> divide by 1)  diffrend/model.py:18-32
  * triangle spec (w=1 faces, w=0 normals) . diffrend/model.py:202-211
  * JSON ``objects.obj[]`` expansion with scale -> rotate -> translate
    ........................................ diffrend/torch/render.py:9-78
  * axis/angle rotation via unit quaternion  diffrend/numpy/quaternion.py:12-30,76-86
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

# primitive type codes shared with include/srh.h (SRH_PRIM_*)
PRIM_DISK, PRIM_PLANE, PRIM_SPHERE, PRIM_TRIANGLE = 0, 1, 2, 3
PRIM_CODE = {"disk": PRIM_DISK, "plane": PRIM_PLANE, "sphere": PRIM_SPHERE, "triangle": PRIM_TRIANGLE}
PRIM_NAME = {v: k for k, v in PRIM_CODE.items()}


# ----------------------------------------------------------------------------------------
# model files
# ----------------------------------------------------------------------------------------
def _records(path: str):
    """Yield (tag, fields) for every non-empty, non-comment line of a text model file."""
    with open(path, "r") as fh:
        for raw in fh:
            parts = raw.split()
            if not parts or parts[0].startswith("#"):
                continue
            yield parts[0], parts[1:]


def load_obj(path: str) -> Dict[str, np.ndarray]:
    """Wavefront OBJ: ``v`` rows and ``f`` rows only; ``a/b/c`` face tokens keep ``a``; indices
    become 0-based.  vn / vt / mtl records are ignored (model.py:119-144)."""
    verts: List[List[float]] = []
    faces: List[List[int]] = []
    for tag, vals in _records(path):
        if tag == "v":
            verts.append([float(x) for x in vals])
        elif tag == "f":
            faces.append([int(tok.split("/")[0]) - 1 for tok in vals])
    return {"v": np.asarray(verts, dtype=np.float64), "f": np.asarray(faces, dtype=np.int64)}


def load_splat(path: str) -> Dict[str, Any]:
    """.splat: triplets of ``v`` (centre), ``vn`` (normal), ``r`` (radius; one value per disk, kept
    as an (M,1) column like the reference does, model.py:90-116)."""
    v: List[List[float]] = []
    vn: List[List[float]] = []
    r: List[List[float]] = []
    for tag, vals in _records(path):
        row = [float(x) for x in vals]
        if tag == "v":
            v.append(row)
        elif tag == "vn":
            vn.append(row)
        elif tag == "r":
            r.append(row)
    return {"v": np.asarray(v), "vn": np.asarray(vn), "r": np.asarray(r), "type": "splat"}


def load_off(path: str) -> Dict[str, np.ndarray]:
    """Object File Format; the header counts may share the ``OFF`` line (model.py:147-188)."""
    toks: List[List[str]] = []
    with open(path, "r") as fh:
        for raw in fh:
            parts = raw.split()
            if parts:
                toks.append(parts)
    if not toks or not toks[0][0].startswith("OFF"):
        raise ValueError(f"{path}: not an OFF file")
    head = toks[0][1:] if len(toks[0]) > 1 else toks[1]
    body = toks[1:] if len(toks[0]) > 1 else toks[2:]
    nv, nf, ne = (int(x) for x in head[:3])
    verts = [[float(x) for x in row] for row in body[:nv]]
    faces = [[int(x) for x in row[1:]] for row in body[nv:nv + nf]]
    edges = [[int(x) for x in row] for row in body[nv + nf:nv + nf + ne]]
    return {"v": np.asarray(verts), "f": np.asarray(faces), "e": np.asarray(edges)}


def load_model(path: str) -> Dict[str, Any]:
    ext = os.path.splitext(path)[1].lower()
    try:
        fn = {".obj": load_obj, ".off": load_off, ".splat": load_splat}[ext]
    except KeyError:
        raise ValueError(f"unsupported model extension {ext!r} ({path})")
    return fn(path)


def face_normals(v: np.ndarray, f: np.ndarray, unnormalized: bool = False) -> np.ndarray:
    """cross(v1-v0, v2-v0), unit length; zero-area faces keep the zero vector (model.py:18-32)."""
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(b - a, c - a)
    if unnormalized:
        return n
    length = np.sqrt(np.sum(n * n, axis=-1, keepdims=True))
    length[length == 0] = 1.0
    return n / length


def obj_to_triangle_spec(obj: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """(F,3,4) homogeneous vertices with w=1 and (F,4) face normals with w=0 (model.py:202-211)."""
    tri = obj["v"][obj["f"]]
    nrm = face_normals(obj["v"], obj["f"])
    face = np.concatenate([tri, np.ones(tri.shape[:-1] + (1,))], axis=-1)
    normal = np.concatenate([nrm, np.zeros((nrm.shape[0], 1))], axis=-1)
    return {"face": face, "normal": normal}


def circum_circles(v: np.ndarray, f: np.ndarray) -> Dict[str, np.ndarray]:
    """Circumscribed circle of every triangle (centre, radius), from cross / dot products as the reference does
    (model.py:35-63): with a = p1-p2, b = p2-p3, c = p1-p3 and n = a x b,
    radius = |a||b||c| / (2|n|), centre = alpha p1 + beta p2 + gamma p3 with the barycentric weights below."""
    p1, p2, p3 = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    p23, p13, p12 = p2 - p3, p1 - p3, p1 - p2
    n2 = np.sum(np.cross(p12, p23) ** 2, axis=-1)
    length = lambda x: np.sqrt(np.sum(x ** 2, axis=-1))  # noqa: E731
    radius = length(p12) * length(p23) * length(p13) / (2 * np.sqrt(n2))
    inv = 1.0 / (2 * n2)
    alpha = np.sum(p23 ** 2, axis=-1) * np.sum(p12 * p13, axis=-1) * inv
    beta = np.sum(p13 ** 2, axis=-1) * np.sum(-p12 * p23, axis=-1) * inv
    gamma = np.sum(p12 ** 2, axis=-1) * np.sum(p13 * p23, axis=-1) * inv
    centre = alpha[:, None] * p1 + beta[:, None] * p2 + gamma[:, None] * p3
    return {"center": centre, "radius": radius}


def obj_to_splat(obj: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Mesh -> disc splats: one disc per face, the face's circumscribed circle with the face normal
    (model.py:66-75; this is how the reference's data/bunny.splat was made from data/bunny.obj)."""
    cc = circum_circles(obj["v"], obj["f"])
    return {"v": cc["center"], "r": cc["radius"], "vn": face_normals(obj["v"], obj["f"]), "type": "splat"}


def write_splat(path: str, splat: Dict[str, np.ndarray]) -> None:
    """The reference's text format: `v x y z` / `vn x y z` / `r radius` per disc (model.py:78-87)."""
    with open(path, "w") as fh:
        for v, vn, r in zip(splat["v"], splat["vn"], np.asarray(splat["r"]).reshape(len(splat["v"]), -1)):
            fh.write("v " + " ".join(repr(float(x)) for x in v) + "\n")
            fh.write("vn " + " ".join(repr(float(x)) for x in vn) + "\n")
            fh.write("r " + " ".join(repr(float(x)) for x in r) + "\n")


def axis_angle_matrix(axis, angle: float) -> np.ndarray:
    """3x3 rotation about ``axis`` by ``angle`` radians, built the way the reference does: unit
    quaternion (cos a/2, sin a/2 * axis/|axis|) then the s = 2/|q|^2 matrix
    (numpy/quaternion.py:19-30, 76-86)."""
    ax = np.asarray(axis, dtype=np.float64)[:3]
    ax = ax / np.sqrt(np.sum(ax ** 2))
    w = np.cos(angle / 2.0)
    x, y, z = ax * np.sin(angle / 2.0)
    s = 2.0 / (w * w + x * x + y * y + z * z)
    return np.array([
        [1 - s * (y ** 2 + z ** 2), s * (x * y - w * z), s * (x * z + w * y)],
        [s * (x * y + w * z), 1 - s * (x ** 2 + z ** 2), s * (y * z - w * x)],
        [s * (x * z - w * y), s * (y * z + w * x), 1 - s * (x ** 2 + y ** 2)],
    ])


def transform_vertices(v: np.ndarray, scale=None, rotate=None, translate=None) -> np.ndarray:
    """scale -> rotate -> translate on row vectors (torch/render.py:9-34)."""
    if scale is not None:
        v = v * np.asarray(scale, dtype=np.float64)[None, :]
    if rotate is not None:
        rot = axis_angle_matrix(rotate["axis"], np.deg2rad(rotate["angle_deg"]))
        v = v @ rot.T
    if translate is not None:
        v = v + np.asarray(translate, dtype=np.float64)[None, :]
    return v


def load_scene(path: str, viewport: Optional[Tuple[int, int]] = None) -> Dict[str, Any]:
    """Read a "diffrend 0.1" JSON scene and expand its ``objects.obj[]`` list into one
    ``objects.triangle`` batch (torch/render.py:37-78).  Other object keys already present in
    the file (disk / sphere / plane / triangle) are kept, in file order, ahead of the expanded
    meshes.  ``viewport=(W, H)`` overrides the camera viewport.

    Unlike the reference loader this one is silent and produces an *integer* material_idx
    (the reference makes a float array, torch/render.py:67, which numpy fancy indexing rejects).
    """
    with open(path, "r") as fh:
        scene = json.load(fh)
    base = os.path.dirname(os.path.abspath(path))
    objects = dict(scene.get("objects", {}))
    meshes = objects.pop("obj", None)
    if meshes:
        faces, normals, mats = [], [], []
        for ent in meshes:
            mesh = load_obj(os.path.join(base, ent["path"]))
            mesh["v"] = transform_vertices(mesh["v"], ent.get("scale"), ent.get("rotate"), ent.get("translate"))
            spec = obj_to_triangle_spec(mesh)
            faces.append(spec["face"])
            normals.append(spec["normal"])
            mats.append(np.full(spec["face"].shape[0], int(ent["material_idx"]), dtype=np.int64))
        tri = {"face": np.concatenate(faces), "normal": np.concatenate(normals),
               "material_idx": np.concatenate(mats)}
        if "triangle" in objects:
            old = objects["triangle"]
            tri = {k: np.concatenate([np.asarray(old[k]), tri[k]]) for k in tri}
        objects["triangle"] = tri
    scene["objects"] = objects
    if viewport is not None:
        scene["camera"]["viewport"] = [0, 0, int(viewport[0]), int(viewport[1])]
    return scene


# ----------------------------------------------------------------------------------------
# expanded dict -> numpy leaves
# ----------------------------------------------------------------------------------------
def _np(x, dtype=None) -> np.ndarray:
    """numpy view of a list / ndarray / torch tensor leaf."""
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=dtype)


_OBJ_FIELDS = {
    "disk": ("pos", "normal", "radius"),
    "plane": ("pos", "normal"),
    "sphere": ("pos", "radius"),
    "triangle": ("face", "normal"),
}


def scene_to_numpy(scene: Dict[str, Any], dtype=np.float64, round_fp32: bool = False) -> Dict[str, Any]:
    """Copy of an expanded scene with ndarray leaves (what ``diffrend.numpy.renderer.render``
    and our oracle consume): float leaves in ``dtype``, index leaves int64, ``radius`` ravelled.
    ``round_fp32`` first rounds every geometry / light / colour value to fp32, i.e. to the values
    the device arrays hold, so that CPU and hip paths see identical inputs.  The caller's scene
    is never modified.  Camera ``eye/at/up`` keep their container type (the reference treats
    list-typed ``at``/``up`` specially, numpy/ops.py:95-100)."""
    def fl(x):
        a = _np(x, np.float64)
        if round_fp32:
            a = a.astype(np.float32).astype(np.float64)
        return a.astype(dtype)

    out: Dict[str, Any] = {}
    cam = dict(scene["camera"])
    for key in ("eye", "at", "up"):
        val = cam[key]
        if hasattr(val, "detach"):
            val = val.detach().cpu().numpy().astype(np.float64)
        cam[key] = list(val) if isinstance(val, (list, tuple)) else np.asarray(val, dtype=np.float64)
    cam["viewport"] = [int(t) for t in _np(cam["viewport"]).ravel()]
    for key in ("fovy", "focal_length", "near", "far"):
        cam[key] = float(_np(cam[key]).ravel()[0])
    out["camera"] = cam

    lights = scene["lights"]
    out["lights"] = {"pos": fl(lights["pos"]), "color_idx": _np(lights["color_idx"]).astype(np.int64).ravel()}
    for opt in ("attenuation", "ambient"):
        if opt in lights:
            out["lights"][opt] = fl(lights[opt])
    out["colors"] = fl(scene["colors"])
    out["materials"] = {"albedo": fl(scene["materials"]["albedo"])}
    if "coeffs" in scene["materials"]:
        out["materials"]["coeffs"] = fl(scene["materials"]["coeffs"])

    objs: Dict[str, Any] = {}
    for kind, grp in scene["objects"].items():
        if kind not in _OBJ_FIELDS:
            raise ValueError(f"unknown object type {kind!r} (expanded scenes hold disk/plane/sphere/triangle; "
                             f"use load_scene() to expand an 'obj' list)")
        g = {name: fl(grp[name]) for name in _OBJ_FIELDS[kind]}
        if "radius" in g:
            g["radius"] = g["radius"].ravel()
        g["material_idx"] = _np(grp["material_idx"]).astype(np.int64).ravel()
        objs[kind] = g
    out["objects"] = objs
    if "tonemap" in scene:
        tm = scene["tonemap"]
        out["tonemap"] = {"type": tm["type"], "gamma": float(_np(tm["gamma"]).ravel()[0])}
    return out


# ----------------------------------------------------------------------------------------
# camera helpers (host side)
# ----------------------------------------------------------------------------------------
def unit_up(given, up64: np.ndarray) -> np.ndarray:
    """The camera's y axis, up / |up| (numpy/ops.py:109).  A list-typed ``up`` is a float32 array at that point
    (:99), so norm and division happen in float32 -- repeated here with the same numpy calls -- and everything after
    (the cross product with the float64 z) is float64 again."""
    if isinstance(given, (list, tuple)):
        up32 = np.asarray(given, dtype=np.float32).reshape(-1)[:3]
        return (up32 / np.linalg.norm(up32, 2)).astype(np.float64)
    up64 = np.asarray(up64, dtype=np.float64).reshape(-1)[:3]
    return up64 / np.linalg.norm(up64, 2)


