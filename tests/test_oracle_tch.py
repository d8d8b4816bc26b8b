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
