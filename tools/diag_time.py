#!/usr/bin/env python3
"""Residency of k_render_binned's waves inside the frame pipeline (library built with -DSRH_DIAG_TIME,
build/diag_time.patch): start / end of every wave of the last eight frames."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from surf_renderer_amd import _lib, renderer, synthetic
from surf_renderer_amd.pipeline import FramePipeline

ap = argparse.ArgumentParser()
ap.add_argument("--inflight", type=int, default=3)
ap.add_argument("--frames", type=int, default=600)
ap.add_argument("--out", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
sc = synthetic.disk_cloud_scene(100_000, 2048, 2048)
buf = renderer.flatten_scene(sc, dev)
cam = renderer.camera_struct(sc["camera"], "numpy")
pipe = FramePipeline(buf, cam, n_inflight=args.inflight)
lib = _lib.load()
for _ in range(300):
    pipe.submit()
pipe.sync()
t0 = time.perf_counter()
for _ in range(args.frames):
    pipe.submit()
pipe.sync()
dt = (time.perf_counter() - t0) / args.frames
ring = np.zeros((8, 16384, 4), dtype=np.uint32)
frame = C.c_uint(0)
assert lib.srh_diag_read(ring.ctypes.data_as(C.POINTER(C.c_uint)), C.byref(frame)) == 0
last = frame.value - 1                       # id of the last frame rendered
tag = os.environ.get("SRH_DIAG_SKIP", "none")
# steady-state frames: skip the last `inflight` (the pipeline drains under them)
ids = [last - args.inflight - k for k in range(8 - args.inflight - 1, -1, -1)]
st = np.concatenate([ring[i & 7, :, 0] for i in ids]).astype(np.int64)
en = np.concatenate([ring[i & 7, :, 1] for i in ids]).astype(np.int64)
base = st.min()
st, en = (st - base) / 100.0, (en - base) / 100.0            # us
life = en - st
# resident render waves over time
ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
t, r = ev[:, 0], np.cumsum(ev[:, 1])
lo, hi = np.percentile(st, 15), np.percentile(en, 85)          # inner window: all frames around are in flight
sel = (t[:-1] >= lo) & (t[1:] <= hi)
w = (t[1:] - t[:-1])[sel]
res = r[:-1][sel]
mean_res = (w * res).sum() / w.sum()
hist = [(w[(res >= a) & (res < b)].sum() / w.sum()) for a, b in ((0, 1024), (1024, 2048), (2048, 3072), (3072, 3584), (3584, 4097), (4097, 1 << 30))]
print(f"diag [{tag}] inflight {args.inflight}: {1e6 * dt:.1f} us/frame; wave life {life.mean():.2f} us; resident render waves "
      f"(inner window {hi - lo:.0f} us): mean {mean_res:.0f}; time share by residency <1024 {hist[0]:.2f}, <2048 {hist[1]:.2f}, "
      f"<3072 {hist[2]:.2f}, <3584 {hist[3]:.2f}, <=4096 {hist[4]:.2f}, >4096 {hist[5]:.2f}")
for k, i in enumerate(ids):
    a, b = (ring[i & 7, :, 0].astype(np.int64) - base) / 100.0, (ring[i & 7, :, 1].astype(np.int64) - base) / 100.0
    print(f"   frame {i}: first wave start {a.min():8.1f}  last start {a.max():8.1f}  last end {b.max():8.1f}  "
          f"(kernel ~{b.max() - a.min():.1f} us)")
sw = np.concatenate([ring[i & 7, :, 2] for i in ids]).astype(np.float64) / 100.0
fi = np.concatenate([ring[i & 7, :, 3] for i in ids]).astype(np.float64) / 100.0
if os.environ.get("SRH_DIAG_CLK"):
    long_ = life > 5.0
    print(f"   shader clock seen by the waves (s_memtime cycles / s_memrealtime): {(fi[long_] * 100.0 / (life[long_] * 100.0)).mean() * 100.0:.1f} MHz")
else:
    print(f"   phases (mean us per wave): sweep {sw.mean():.2f}  finish {fi.mean():.2f}  rest {(life - sw - fi).mean():.2f}")
if args.out:
    np.savez_compressed(args.out, ring=ring, last=last, inflight=args.inflight, us_per_frame=1e6 * dt)
