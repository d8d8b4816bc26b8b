"""CPU side of the directed adversarial scenes (tests/adversarial_scenes.py): the generator reaches the view, and the
oracle's two ways of evaluating the reference's np.dot products (BLAS as the reference, or explicit ordered sums as the
GPU path) decide the same pixels except where the last bit of a cancelling sum decides."""
import numpy as np
import pytest

from adversarial_scenes import KINDS, huge_scene
from oracle import np_oracle
from surf_renderer_amd.scene import scene_to_numpy


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_variants_agree_on_adversarial_scenes(kind):
    rng = np.random.RandomState(31000 + KINDS.index(kind))
    bad = total = covered = 0
    for _ in range(60):
        s = scene_to_numpy(huge_scene(rng, kind), round_fp32=True)
        with np.errstate(all="ignore"):
            a, b = np_oracle.render(s), np_oracle.render(s, dots="ordered")
        ok = (a["nearest"] == b["nearest"]) & (np.isclose(a["depth"], b["depth"], rtol=1.2e-7, atol=0) |
                                               (a["depth"] == b["depth"]))
        bad += int((~ok).sum())
        total += ok.size
        covered += int(np.isfinite(a["depth"]).any())
    assert covered >= 30, f"only {covered} of 60 scenes had a hit pixel"
    # measured over 220 scenes per kind: disc 0, plane 0, triangle 1.7e-4, sphere 2.0e-3 of the pixels
    assert bad <= 0.01 * total, f"{bad} of {total} pixels differ between the BLAS and the ordered products"
