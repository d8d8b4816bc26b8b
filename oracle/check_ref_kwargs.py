#!/usr/bin/env python3
"""What two keyword arguments of the reference torch backend really do (diffrend/torch/renderer.py:152-162,245-260),
established by running the UNMODIFIED reference in the build container.  Test infrastructure; prints a short report.

  backface_culling=True        only labels the primitives (torch/utils.py:515-536); nothing reads the labels, so every
                               output is unchanged -> the hip backend accepts the keyword as a no-op.
  norm_depth_image_only=True   asks the intersection routines for no normals (`disable_normals`, torch/utils.py:301)
                               and then gathers from them all the same (renderer.py:196 / :218): the call raises
                               TypeError before it reaches the normalisation of :245-249, so no golden output of this
                               path can be generated.  The hip backend implements the formula of :245-249 from the
                               source text; that output is NOT pinned by a reference fixture.
"""
import contextlib
import io
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle.gen_golden_tch import f32, ref_tch, to_torch  # noqa: E402  (imports the reference)
from surf_renderer_amd import synthetic  # noqa: E402


def run(sc, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return ref_tch.render(to_torch(sc), shadow=False, **kw)


def main():
    sc = synthetic.demo_scene(64, 48, with_planes=True)
    sc["camera"]["near"] = 0.5
    sc["lights"]["attenuation"] = f32([[1, 0, 0], [0.2, 0.05, 0], [1, 0, 0.001], [0.7, 0.02, 0.0005]])
    sc["lights"]["ambient"] = f32([0.02, 0.015, 0.01])
    sc["materials"]["coeffs"] = f32([[1, 0, 0], [0.8, 0.2, 4], [0.6, 0.4, 16], [0.9, 0.1, 2], [0.5, 0.5, 8], [0.7, 0.3, 32]])
    for tiled in (False, True):
        a, b = run(sc, tiled=tiled), run(sc, tiled=tiled, backface_culling=True)
        for k in ("image", "depth", "nearest", "normal", "pos"):
            assert np.array_equal(a[k].numpy(), b[k].numpy()), k
        print(f"backface_culling=True (tiled={tiled}): image, depth, nearest, normal, pos unchanged")
        try:
            run(sc, tiled=tiled, norm_depth_image_only=True)
            print(f"norm_depth_image_only=True (tiled={tiled}): returned")
        except TypeError as exc:
            print(f"norm_depth_image_only=True (tiled={tiled}): TypeError: {str(exc).splitlines()[0][:90]}")


if __name__ == "__main__":
    main()
