"""Golden-vector container (test infrastructure).

One ``.npz`` per golden case: the *inputs* (an expanded scene dict, flattened to ``scene/...`` keys,
geometry stored as float32 because every value is fp32-representable) and the *expected outputs*
of the reference numpy backend (``out/image``, ``out/depth`` float64, ``out/nearest`` int64).
Only data -- no code of the reference travels in these files.
"""
from __future__ import annotations

import json
from typing import Any, Dict

import numpy as np

_CAM_VECS = ("eye", "at", "up")


def pack_scene(scene: Dict[str, Any]) -> Dict[str, np.ndarray]:
    flat: Dict[str, np.ndarray] = {}
    cam = scene["camera"]
    meta = {"objects_order": list(scene["objects"].keys()),
            "camera_list_typed": [k for k in _CAM_VECS if isinstance(cam[k], (list, tuple))],
            "has_tonemap": "tonemap" in scene, "proj_type": str(cam.get("proj_type", "perspective"))}
    for k in _CAM_VECS:
        flat[f"scene/camera/{k}"] = np.asarray(cam[k], dtype=np.float64)
    flat["scene/camera/viewport"] = np.asarray(cam["viewport"], dtype=np.int64)
    for k in ("fovy", "focal_length", "near", "far"):
        flat[f"scene/camera/{k}"] = np.asarray(cam[k], dtype=np.float64)
    flat["scene/lights/pos"] = np.asarray(scene["lights"]["pos"], dtype=np.float32)
    flat["scene/lights/color_idx"] = np.asarray(scene["lights"]["color_idx"], dtype=np.int64)
    flat["scene/colors"] = np.asarray(scene["colors"], dtype=np.float32)
    flat["scene/materials/albedo"] = np.asarray(scene["materials"]["albedo"], dtype=np.float32)
    # optional inputs of the torch backend's shading model
    if "attenuation" in scene["lights"]:
        flat["scene/lights/attenuation"] = np.asarray(scene["lights"]["attenuation"], dtype=np.float32)
    if "ambient" in scene["lights"]:
        flat["scene/lights/ambient"] = np.asarray(scene["lights"]["ambient"], dtype=np.float32)
    if "coeffs" in scene["materials"]:
        flat["scene/materials/coeffs"] = np.asarray(scene["materials"]["coeffs"], dtype=np.float32)
    for kind, grp in scene["objects"].items():
        for name, val in grp.items():
            dt = np.int64 if name == "material_idx" else np.float32
            arr = np.asarray(val)
            if dt is np.float32 and not np.array_equal(arr.astype(np.float32).astype(np.float64),
                                                        arr.astype(np.float64)):
                raise ValueError(f"{kind}.{name} is not fp32-representable")
            flat[f"scene/objects/{kind}/{name}"] = arr.astype(dt)
    if "tonemap" in scene:
        flat["scene/tonemap/gamma"] = np.asarray(scene["tonemap"]["gamma"], dtype=np.float64).ravel()[:1]
    flat["meta"] = np.asarray(json.dumps(meta))
    return flat


def unpack_scene(npz) -> Dict[str, Any]:
    """Rebuild the ndarray-leaf scene dict (float64 leaves) the oracle and the reference consume."""
    meta = json.loads(str(npz["meta"]))
    cam: Dict[str, Any] = {}
    for k in _CAM_VECS:
        v = np.asarray(npz[f"scene/camera/{k}"], dtype=np.float64)
        cam[k] = [float(t) for t in v] if k in meta["camera_list_typed"] else v
    cam["viewport"] = [int(t) for t in npz["scene/camera/viewport"]]
    for k in ("fovy", "focal_length", "near", "far"):
        cam[k] = float(npz[f"scene/camera/{k}"])
    if meta.get("proj_type", "perspective") != "perspective":
        cam["proj_type"] = meta["proj_type"]
    scene: Dict[str, Any] = {
        "camera": cam,
        "lights": {"pos": npz["scene/lights/pos"].astype(np.float64),
                   "color_idx": npz["scene/lights/color_idx"].astype(np.int64)},
        "colors": npz["scene/colors"].astype(np.float64),
        "materials": {"albedo": npz["scene/materials/albedo"].astype(np.float64)},
        "objects": {},
    }
    for key, (a, b) in {"scene/lights/attenuation": ("lights", "attenuation"), "scene/lights/ambient": ("lights", "ambient"),
                        "scene/materials/coeffs": ("materials", "coeffs")}.items():
        if key in npz.files:
            scene[a][b] = npz[key].astype(np.float64)
    for kind in meta["objects_order"]:
        grp = {}
        prefix = f"scene/objects/{kind}/"
        for key in npz.files:
            if key.startswith(prefix):
                name = key[len(prefix):]
                arr = npz[key]
                grp[name] = arr.astype(np.int64) if name == "material_idx" else arr.astype(np.float64)
        scene["objects"][kind] = grp
    if meta["has_tonemap"]:
        scene["tonemap"] = {"type": "gamma", "gamma": float(npz["scene/tonemap/gamma"][0])}
    return scene


def save_case(path: str, scene: Dict[str, Any], out: Dict[str, np.ndarray], note: str = "") -> None:
    flat = pack_scene(scene)
    flat["out/image"] = np.asarray(out["image"], dtype=np.float64)
    flat["out/depth"] = np.asarray(out["depth"], dtype=np.float64)
    flat["out/nearest"] = np.asarray(out["nearest"], dtype=np.int64)
    flat["note"] = np.asarray(note)
    np.savez_compressed(path, **flat)


def load_case(path: str):
    npz = np.load(path, allow_pickle=False)
    out = {"image": npz["out/image"], "depth": npz["out/depth"], "nearest": npz["out/nearest"]}
    return unpack_scene(npz), out, str(npz["note"])
