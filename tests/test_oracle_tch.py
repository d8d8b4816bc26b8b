"""The torch-semantics oracle (oracle/np_oracle_tch.py) against outputs of the reference's torch backend
(tests/golden/t*.npz, oracle/gen_golden_tch.py).  The reference computes this path in float32, the oracle in
float64, so: `nearest` may differ on a small fraction of silhouette pixels (<= 0.5 %), and where it agrees depth must
match to 2e-5 relative, image / normal to 3e-4 absolute, pos to 2e-4."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import np_oracle_tch
from oracle.golden_io import unpack_scene

CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "t*.npz")))


def load_tch_case(name):
    npz = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    scene = unpack_scene(npz)
    want = {k: npz["out/" + k] for k in ("image", "depth", "nearest", "normal", "pos")}
    return scene, want, json.loads(str(npz["kwargs"]))


def assert_tch_parity(got, want, far, max_mismatch=0.005):
    hit_w = want["depth"] <= far
    hit_g = got["depth"] <= far
    same = (hit_w == hit_g) & ((got["nearest"] == want["nearest"]) | ~hit_w)
    assert 1.0 - same.mean() <= max_mismatch, f"{1.0 - same.mean():.3%} of pixels differ in hit / nearest"
    ok = same & hit_w
    np.testing.assert_allclose(got["depth"][ok], want["depth"][ok], rtol=2e-5)
    np.testing.assert_allclose(got["depth"][same & ~hit_w], far + 1.0)
    np.testing.assert_allclose(got["image"][same], want["image"][same], atol=3e-4)
    np.testing.assert_allclose(got["normal"][ok], want["normal"][ok], atol=3e-4)
    np.testing.assert_allclose(got["pos"][ok], want["pos"][ok], atol=2e-4, rtol=2e-5)


def test_golden_set():
    assert len(CASES) >= 6


@pytest.mark.parametrize("case", CASES)
def test_tch_oracle_matches_reference_torch_backend(case):
    scene, want, kw = load_tch_case(case)
    got = np_oracle_tch.render(scene, **kw)
    assert_tch_parity(got, want, scene["camera"]["far"])


SHADOW_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "s1*.npz")))


def load_shadow_case(name):
    npz = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return unpack_scene(npz), {k: npz["out/" + k] for k in ("image", "depth", "nearest")}, json.loads(str(npz["kwargs"]))


@pytest.mark.parametrize("case", SHADOW_CASES)
def test_shadow_oracle_matches_reference_torch_backend(case):
    """render(shadow=True) of the reference (oracle/gen_golden_shadow.py; the reference returns the shaded image, not
    the visibility): the oracle's image must match on every pixel whose `nearest` agrees, and the fixture must
    really hold shadows (the unshadowed oracle image differs from it on > 20 % of the pixels)."""
    assert len(SHADOW_CASES) >= 3
    scene, want, kw = load_shadow_case(case)
    got = np_oracle_tch.render(scene, shadow=True, **kw)
    far = scene["camera"]["far"]
    same = (got["nearest"] == want["nearest"]) | (want["depth"] > far)
    assert same.mean() >= 0.995
    np.testing.assert_allclose(got["image"][same], want["image"][same], atol=3e-4)
    plain = np_oracle_tch.render(scene, **kw)
    assert (np.abs(plain["image"] - want["image"]).max(axis=-1) > 3e-4).mean() > 0.2


def test_shadow_rays_against_geometry():
    """Besides the reference's own outputs above, the shadow rays against geometry: a unit disc at height 1 under a
    light at height 5 shadows the floor inside radius 5 / 4, and nothing else; the disc itself and the unshadowed
    floor see the light."""
    f = lambda a: np.asarray(a, dtype=np.float64)
    scene = {
        "camera": {"viewport": [0, 0, 64, 48], "fovy": float(np.deg2rad(50.0)), "focal_length": 1.0,
                   "eye": f([0.0, -9.0, 6.0, 1.0]), "at": f([0.0, 0.0, 0.0, 1.0]), "up": f([0.0, 0.0, 1.0, 0.0]),
                   "near": 0.1, "far": 100.0},
        "lights": {"pos": f([[0, 0, 5, 1]]), "color_idx": np.array([1]), "attenuation": f([[1, 0, 0]]), "ambient": f([0, 0, 0])},
        "colors": f([[0, 0, 0], [1, 1, 1]]),
        "materials": {"albedo": f([[0.8, 0.8, 0.8]]), "coeffs": f([[1, 0, 0]])},
        "objects": {"plane": {"pos": f([[0, 0, 0, 1]]), "normal": f([[0, 0, 1, 0]]), "material_idx": np.array([0])},
                    "disk": {"pos": f([[0, 0, 1, 1]]), "normal": f([[0, 0, 1, 0]]), "radius": f([1.0]), "material_idx": np.array([0])}},
    }
    res = np_oracle_tch.render(scene, shadow=True)
    plain = np_oracle_tch.render(scene)
    vis = res["visibility"][0]
    pos, near = res["pos"], res["nearest"]
    hit = res["depth"] <= 100.0
    floor = hit & (near == 0)
    rad = np.hypot(pos[..., 0], pos[..., 1])
    assert (floor & (rad < 1.2)).sum() > 5 and (floor & (rad > 1.3)).sum() > 100
    assert not vis[floor & (rad < 1.2)].any()                  # inside the shadow
    assert vis[floor & (rad > 1.3)].all()                      # outside it
    assert vis[hit & (near == 1)].all()                        # the disc does not shadow itself
    assert np.all(res["image"][floor & (rad < 1.2)] == 0) and np.all(plain["image"][floor & (rad < 1.2)] > 0)
    np.testing.assert_array_equal(res["image"][vis & hit], plain["image"][vis & hit])
