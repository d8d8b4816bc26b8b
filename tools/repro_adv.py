#!/usr/bin/env python3
"""Re-create scene N of an adversarial fuzz campaign (tools/fuzz_campaign.py --adversarial --seed S) and show how the
modes differ on it.  usage: tools/repro_adv.py SEED N [--save file.npz]"""
import os, sys, json, pickle
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from adversarial_scenes import KINDS, huge_scene
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
for it in range(n + 1):
    scene = huge_scene(rng, KINDS[it % 4])
if "--save" in sys.argv:
    pickle.dump(scene, open(sys.argv[sys.argv.index("--save") + 1], "wb"))
from surf_renderer_amd import render
def R(**kw):
    res = render(scene, device="cuda:0", **kw); torch.cuda.synchronize()
    return {k: res[k].cpu().numpy() for k in ("image", "depth", "nearest")}
ref = R(mode="exact")
print("camera", {k: (np.asarray(v).tolist() if not isinstance(v, (int, float, str)) else v) for k, v in scene["camera"].items()})
for kind, grp in scene["objects"].items():
    print(kind, {k: np.asarray(v).shape for k, v in grp.items()})
for mode, wpt in (("fast", 0), ("binned", 1), ("binned", 4)):
    got = R(mode=mode, waves_per_tile=wpt)
    bad = got["nearest"] != ref["nearest"]
    print(mode, wpt, "nearest differs on", int(bad.sum()), "depth differs on", int((~((got["depth"] == ref["depth"]) | (np.isnan(got["depth"]) & np.isnan(ref["depth"])))).sum()))
    dbad = ~((got["depth"] == ref["depth"]) | (np.isnan(got["depth"]) & np.isnan(ref["depth"]))) & ~bad
    ys, xs = np.nonzero(dbad)
    for y, x in list(zip(ys, xs))[:6]:
        print("   depth only: pixel", (int(y), int(x)), "nearest", int(ref["nearest"][y, x]), "exact", float(ref["depth"][y, x]), "got", float(got["depth"][y, x]), "image", ref["image"][y, x].tolist(), got["image"][y, x].tolist())
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:12]:
        print("   pixel", (int(y), int(x)), "exact", int(ref["nearest"][y, x]), float(ref["depth"][y, x]), "| got", int(got["nearest"][y, x]), float(got["depth"][y, x]))
