#!/usr/bin/env python3
"""Where the render kernel's work is at BASELINE config 5 (measurement; one MI355X): tiles with entries, entries per busy
tile, pixels hit per tile and the finish rounds they imply (ceil(pixels / 64) per tile; hit pixels are a lower bound of
the pixels with a candidate)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from surf_renderer_amd import renderer, synthetic  # noqa: E402

scene = synthetic.disk_cloud_scene()
buf = renderer.flatten_scene(scene, device="cuda:0")
cam = renderer.camera_struct(scene["camera"])
st = renderer.bin_statistics(buf, cam)
image, depth, nearest = renderer.render_buffers(buf, cam, mode="binned")
torch.cuda.synchronize()
hit = torch.isfinite(depth).cpu().numpy()
H, W = hit.shape
t = hit.reshape(H // 16, 16, W // 16, 16).sum(axis=(1, 3))
ent = st["entries"].sum(axis=0)
busy = ent > 0
rounds = np.ceil(t / 64.0)
out = {
    "tiles": int(ent.size), "tiles_with_entries": int(busy.sum()), "entries": int(ent.sum()),
    "entries_per_busy_tile_mean": float(ent[busy].mean()), "entries_per_busy_tile_p50_p90_max": [int(np.percentile(ent[busy], 50)), int(np.percentile(ent[busy], 90)), int(ent.max())],
    "hit_pixels": int(hit.sum()), "hit_fraction": float(hit.mean()),
    "tiles_with_hits": int((t > 0).sum()), "finish_rounds_lower_bound": int(rounds.sum()),
    "rounds_if_perfectly_packed": int(np.ceil(hit.sum() / 64.0)),
    "hits_per_tile_hist_0_1to64_65to128_129to192_193to255_256": [int((t == 0).sum()), int(((t > 0) & (t <= 64)).sum()), int(((t > 64) & (t <= 128)).sum()), int(((t > 128) & (t <= 192)).sum()), int(((t > 192) & (t < 256)).sum()), int((t == 256).sum())],
}
print(json.dumps(out))
