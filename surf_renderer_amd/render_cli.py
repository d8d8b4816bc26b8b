"""Backend dispatcher the reference's README advertises but never implemented
(`python diffrend/render.py --use [gl|np|tf|tch] --scene ...`, README.md:15 vs diffrend/render.py:12-24):

    python -m surf_renderer_amd.render_cli --use hip --scene assets/scenes/basic.json --out_dir out/

`--use hip` renders with this package; `--use np` / `--use tch` import the reference's own backends when a
DiffRend checkout is importable (they are not part of this package).  Writes image.npy, depth.npy, nearest.npy and
8-bit PNGs like diffrend/torch/render.py:121-129 does.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

import numpy as np

BACKENDS = {"hip": "surf_renderer_amd.renderer", "np": "diffrend.numpy.renderer", "tch": "diffrend.torch.renderer"}


from .frame_writer import write_png  # noqa: E402  (shared with the asynchronous writer)


def to_numpy(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--use", choices=sorted(BACKENDS), default="hip")
    ap.add_argument("--scene", required=True, help="diffrend 0.1 JSON scene")
    ap.add_argument("--out_dir", default="./render_samples")
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--mode", default="auto", help="hip only: auto | exact | fast | binned")
    ap.add_argument("--shading", default="numpy", choices=["numpy", "torch"],
                    help="hip only: which reference backend's semantics to follow (torch: Phong model with "
                         "attenuation / specular / ambient, orthographic cameras, shadows)")
    ap.add_argument("--double_sided", action="store_true", help="hip, torch shading: flip normals towards the viewer")
    ap.add_argument("--shadow", action="store_true", help="hip, torch shading: shadow rays (all pairs)")
    args = ap.parse_args(argv)
    if (args.shadow or args.double_sided) and not (args.use == "hip" and args.shading == "torch"):
        ap.error("--shadow / --double_sided exist only in the torch backend's semantics: add --use hip --shading torch")

    from .scene import load_scene, scene_to_numpy
    vp = (args.width, args.height) if args.width and args.height else None
    scene = load_scene(args.scene, viewport=vp)
    try:
        backend = importlib.import_module(BACKENDS[args.use])
    except ImportError as exc:
        print(f"backend {args.use!r} is not importable here: {exc}", file=sys.stderr)
        return 2
    if args.use == "hip":
        kw = {}
        if args.shading == "torch":
            kw = {"shading": "torch", "double_sided": args.double_sided, "shadow": args.shadow}
        res = backend.render(scene, mode=args.mode, **kw)
    elif args.use == "np":
        res = backend.render(scene_to_numpy(scene))
    else:
        from diffrend.torch.render import make_torch_var
        res = backend.render(make_torch_var(scene_to_numpy(scene)))

    os.makedirs(args.out_dir, exist_ok=True)
    image, depth = to_numpy(res["image"]).squeeze(), to_numpy(res["depth"]).squeeze()
    np.save(os.path.join(args.out_dir, "image.npy"), image)
    np.save(os.path.join(args.out_dir, "depth.npy"), depth)
    if "nearest" in res:
        np.save(os.path.join(args.out_dir, "nearest.npy"), to_numpy(res["nearest"]))
    write_png(os.path.join(args.out_dir, "im.png"), np.uint8(255 * np.clip(np.nan_to_num(image), 0, 1)))
    fin = np.isfinite(depth)
    if fin.any():
        lo, hi = depth[fin].min(), depth[fin].max()
        norm = np.where(fin, (depth - lo) / max(hi - lo, 1e-30), 1.0)
        write_png(os.path.join(args.out_dir, "depth.png"), np.uint8(255 * norm))
    print(f"{args.use}: {image.shape[1]}x{image.shape[0]} -> {args.out_dir}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
