"""Exploration run of the directed adversarial scenes (tests/adversarial_scenes.py) on the GPU: per scene, do the
accelerated modes equal the all-pairs fp64 mode, and on how many pixels does the all-pairs mode differ from the numpy
oracle?  usage: python tools/adv_explore.py [seed] [scenes per kind]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from adversarial_scenes import KINDS, huge_scene          # noqa: E402
from oracle import np_oracle                               # noqa: E402
from surf_renderer_amd import render                       # noqa: E402
from surf_renderer_amd.scene import scene_to_numpy         # noqa: E402


def rend(scene, **kw):
    res = render(scene, device="cuda:0", **kw)
    torch.cuda.synchronize()
    return {k: res[k].cpu().numpy() for k in ("image", "depth", "nearest")}


seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 31000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 220
for ki, kind in enumerate(KINDS):
    rng = np.random.RandomState(seed0 + ki)
    bad_modes = bad_oracle = px_oracle = px_total = 0
    for it in range(n):
        sc = huge_scene(rng, kind)
        ref = rend(sc, mode="exact")
        for mode, wpt in (("fast", 0), ("binned", 1), ("binned", 4)):
            got = rend(sc, mode=mode, waves_per_tile=wpt)
            for k in ("nearest", "depth", "image"):
                bad = ~((got[k] == ref[k]) | (np.isnan(got[k]) & np.isnan(ref[k])))
                if bad.any():
                    bad_modes += 1
                    print(f"MODES {kind} seed {seed0 + ki} scene {it} {mode}/{wpt} {k}: {bad.sum()} values", flush=True)
        with np.errstate(all="ignore"):
            want = np_oracle.render(scene_to_numpy(sc, round_fp32=True), dots=os.environ.get("ADV_DOTS", "ordered"))
        d = ref["depth"].astype(np.float64)
        ok = (ref["nearest"] == want["nearest"]) & (np.isclose(d, want["depth"], rtol=1.2e-7, atol=0) | (d == want["depth"]) |
                                                    ((np.abs(want["depth"]) > 3e38) & np.isinf(d)))
        px_total += ok.size
        if not ok.all():
            bad_oracle += 1
            px_oracle += int((~ok).sum())
            print(f"oracle {kind} scene {it}: {(~ok).sum()} of {ok.size} pixels", flush=True)
    print(f"== {kind}: {n} scenes, mode mismatches {bad_modes}, scenes off the oracle {bad_oracle} "
          f"({px_oracle} of {px_total} pixels)", flush=True)
