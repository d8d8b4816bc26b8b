#!/usr/bin/env python3
"""Find and dump the first failing scene of the non-finite fuzz for a seed (tests/test_hip_parity.py:
test_fuzz_non_finite_and_degenerate_primitives).  usage: tools/repro_nonfinite.py SEED [MAX_SCENES] [--save file.pkl]"""
import os, sys, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T
from oracle import np_oracle
from surf_renderer_amd.scene import scene_to_numpy
seed = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3000
rng = np.random.RandomState(seed)
for it in range(n):
    scene = T._random_scene(rng)
    T._poison(rng, scene, float(rng.choice([0.02, 0.2, 0.5])))
    ref = T._render(scene, mode="exact")
    bad = None
    for mode, wpt in (("fast", 0), ("binned", 1), ("binned", 4)):
        got = T._render(scene, mode=mode, waves_per_tile=wpt)
        for k in ("nearest", "depth", "image"):
            if not np.array_equal(got[k], ref[k], equal_nan=True):
                ne = ~((got[k] == ref[k]) | (np.isnan(got[k]) & np.isnan(ref[k])))
                bad = (mode, wpt, k, int(ne.sum()))
                ys, xs = np.nonzero(ne if ne.ndim == 2 else ne.any(axis=-1))
                print(f"scene {it}: {mode}/{wpt} {k} differs on {ne.sum()} values; first pixels:")
                for y, x in list(zip(ys, xs))[:8]:
                    print("   ", (int(y), int(x)), "exact nearest/depth", int(ref["nearest"][y, x]), float(ref["depth"][y, x]),
                          "| got", int(got["nearest"][y, x]), float(got["depth"][y, x]))
                break
        if bad:
            break
    W, H = scene["camera"]["viewport"][2:]
    if not bad and W * H <= 64 * 80 and sum(len(g["material_idx"]) for g in scene["objects"].values()) <= 800:
        with np.errstate(all="ignore"):
            want = np_oracle.render(scene_to_numpy(scene, round_fp32=True), dots="ordered")
        if not np.array_equal(ref["nearest"], want["nearest"]):
            ne = ref["nearest"] != want["nearest"]
            bad = ("oracle", 0, "nearest", int(ne.sum()))
            ys, xs = np.nonzero(ne)
            print(f"scene {it}: exact vs ORACLE nearest differs on {ne.sum()} pixels")
            for y, x in list(zip(ys, xs))[:8]:
                print("   ", (int(y), int(x)), "exact", int(ref["nearest"][y, x]), float(ref["depth"][y, x]), "| oracle", int(want["nearest"][y, x]), float(want["depth"][y, x]))
        else:
            d = ref["depth"].astype(np.float64)
            okd = np.isclose(d, want["depth"], rtol=T.DEPTH_RTOL, atol=0) | (d == want["depth"])
            img = ref["image"].astype(np.float64)
            oki = np.isclose(img, want["image"], rtol=T.IMAGE_RTOL, atol=T.IMAGE_ATOL, equal_nan=True) | ((np.abs(want["image"]) > 3e38) & np.isinf(img))
            if not okd.all() or not oki.all():
                bad = ("oracle", 0, "depth/image", int((~okd).sum() + (~oki).sum()))
                print(f"scene {it}: exact vs ORACLE depth off on {(~okd).sum()}, image on {(~oki).sum()} values")
                ys, xs = np.nonzero(~oki.all(axis=-1) | ~okd)
                for y, x in list(zip(ys, xs))[:8]:
                    print("   ", (int(y), int(x)), "nearest", int(ref["nearest"][y, x]), "depth", float(d[y, x]), float(want["depth"][y, x]), "image", img[y, x].tolist(), want["image"][y, x].tolist())
    if bad:
        print("camera", scene["camera"])
        for kind, grp in scene["objects"].items():
            print(kind, {k: np.asarray(v).shape for k, v in grp.items()})
        if "--save" in sys.argv:
            pickle.dump(scene, open(sys.argv[sys.argv.index("--save") + 1], "wb"))
        sys.exit(1)
print("no failure in", n, "scenes")
