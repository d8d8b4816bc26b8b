import sys, json, numpy as np, torch
sys.path.insert(0, ".")
from surf_renderer_amd import renderer, synthetic
scene = synthetic.disk_cloud_scene()
buf = renderer.flatten_scene(scene, device="cuda:0")
cam = renderer.camera_struct(scene["camera"])
image, depth, nearest = renderer.render_buffers(buf, cam, mode="binned")
torch.cuda.synchronize()
n = nearest.cpu().numpy()
hit = np.isfinite(depth.cpu().numpy())
t = n.reshape(128, 16, 128, 16).sum(axis=(1, 3))
print(json.dumps({"slow_pixels": int(n.sum()), "hit_pixels": int(hit.sum()), "tiles_with_slow": int((t > 0).sum()),
                  "tiles_with_more_than_2": int((t > 2).sum()), "max_per_tile": int(t.max())}))
