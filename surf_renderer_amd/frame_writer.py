"""Asynchronous image writer: the other half of the reference's batch renderer.

diffrend/torch/batch_render.py renders views in a loop and hands each result to a writer PROCESS through a bounded
queue (`save_to_file`, :36-53; queue and process set-up, :165-179), so PNG encoding and disk writes overlap rendering.
`FrameWriter` is that writer for the hip backend: same files (`img<suffix>.png`, `depth<suffix>.png`), same
conversions (image -> uint8(255 x); depth: background -> minimum, then 8-bit min-max normalisation), a `None` sentinel
to stop.  Device tensors are brought to the host through pinned memory with an asynchronous copy; the queue carries
float32 ndarrays.  The writer process imports numpy only (spawned, like the reference's, so it never inherits a GPU
context; the package's own names are imported lazily, so importing this module does not pull in torch).

    with FrameWriter(out_dir) as w:
        for i, cam in enumerate(cameras):
            res = render({**scene, "camera": cam}, shading="torch")
            w.put(f"_{i}", res["image"], res["depth"], cam["far"])
"""
from __future__ import annotations

import multiprocessing as mp
import os
import queue as queue_mod
import struct
import zlib
from typing import Any, Dict, Optional, Sequence

import numpy as np

WRITE_TO_FILE_QUEUE_SIZE = 10240          # batch_render.py:34


def write_png(path: str, img: np.ndarray) -> None:
    """Minimal 8-bit grey / RGB PNG writer (the image has no imaging library)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    h, w, ch = img.shape
    color = {1: 0, 3: 2}[ch]
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def read_png(path: str) -> np.ndarray:
    """Inverse of write_png (filter type 0 rows only) -- for tests."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, shape = 8, b"", None
    while at < len(data):
        n, tag = struct.unpack(">I", data[at:at + 4])[0], data[at + 4:at + 8]
        body = data[at + 8:at + 8 + n]
        if tag == b"IHDR":
            w, h, _, color = struct.unpack(">IIBB", body[:10])
            shape = (h, w, {0: 1, 2: 3}[color])
        elif tag == b"IDAT":
            idat += body
        at += 12 + n
    raw = zlib.decompress(idat)
    h, w, ch = shape
    rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, 1 + w * ch)
    assert not rows[:, 0].any()
    return rows[:, 1:].reshape(h, w, ch).squeeze(-1) if ch == 1 else rows[:, 1:].reshape(h, w, ch)


def encode_frame(image: np.ndarray, depth: np.ndarray, camera_far: float):
    """The conversions of save_to_file (batch_render.py:46-50): returns (uint8 image, uint8 depth)."""
    with np.errstate(all="ignore"):
        im = np.uint8(255.0 * np.asarray(image))
        depth = np.array(depth, dtype=np.float32, copy=True)
        depth[depth >= camera_far] = depth.min()
        im_depth = np.uint8(255.0 * (depth - depth.min()) / (depth.max() - depth.min()))
    return im, im_depth


def save_to_file(out_dir: str, q) -> None:
    """Writer process body (batch_render.py:36-53): drain the queue until the None sentinel."""
    os.makedirs(out_dir, exist_ok=True)
    while True:
        res = q.get()
        if res is None:
            break
        im, im_depth = encode_frame(res["image"], res["depth"], res["camera_far"])
        write_png(os.path.join(out_dir, "img" + res["suffix"] + ".png"), im)
        write_png(os.path.join(out_dir, "depth" + res["suffix"] + ".png"), im_depth)
        if res.get("npy"):
            np.save(os.path.join(out_dir, "img" + res["suffix"] + ".npy"), res["image"])
            np.save(os.path.join(out_dir, "depth" + res["suffix"] + ".npy"), res["depth"])


def _to_host(x) -> np.ndarray:
    """float32 ndarray of a tensor or array; device tensors come through pinned memory, asynchronously."""
    if hasattr(x, "detach"):                     # torch tensor
        import torch
        t = x.detach()
        if t.is_cuda:
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            t = host
        return np.asarray(t.numpy(), dtype=np.float32)
    return np.asarray(x, dtype=np.float32)


class FrameWriter:
    """Bounded queue + writer process.  `put` blocks only when `queue_size` frames are waiting (back-pressure, as a
    full queue does in the reference); `close` sends the sentinel and joins.  `written` files are complete once
    `close` returns."""

    def __init__(self, out_dir: str, queue_size: int = WRITE_TO_FILE_QUEUE_SIZE, also_npy: bool = False,
                 start_method: str = "spawn"):
        self.out_dir, self.also_npy = out_dir, bool(also_npy)
        ctx = mp.get_context(start_method)       # 'spawn' as in batch_render.py:172
        self._queue = ctx.Queue(queue_size)
        self._proc = ctx.Process(target=save_to_file, args=(out_dir, self._queue), daemon=True)
        self._proc.start()
        self.submitted = 0

    def put(self, suffix: str, image, depth, camera_far: float) -> None:
        if self._proc is None:
            raise RuntimeError("FrameWriter is closed")
        item = {"suffix": str(suffix), "image": _to_host(image), "depth": _to_host(depth),
                "camera_far": float(camera_far), "npy": self.also_npy}
        while True:
            try:
                self._queue.put(item, timeout=1.0)
                break
            except queue_mod.Full:
                if not self._proc.is_alive():
                    raise RuntimeError(f"the writer process died (exit code {self._proc.exitcode})")
        self.submitted += 1

    def close(self, timeout: Optional[float] = None) -> None:
        """Send the sentinel, wait for the writer to drain the queue and end.  A writer that has died is reported, not
        waited for (the sentinel goes through the same full-queue / liveness loop as `put`); one that is still running
        after `timeout` seconds is terminated and reported."""
        if self._proc is None:
            return
        proc, self._proc = self._proc, None
        while proc.is_alive():
            try:
                self._queue.put(None, timeout=1.0)
                break
            except queue_mod.Full:
                continue
        proc.join(timeout)
        if proc.is_alive():
            proc.terminate()
            proc.join(5.0)
            raise RuntimeError(f"the writer process did not finish within {timeout} s and was terminated")
        if proc.exitcode != 0:
            raise RuntimeError(f"the writer process ended with exit code {proc.exitcode}")

    def __enter__(self) -> "FrameWriter":
        return self

    def __exit__(self, exc_type, exc, tb) -> None:
        self.close()


def render_views_to_files(scene: Dict[str, Any], cameras: Sequence[Dict[str, Any]], out_dir: str, batch: int = 64,
                          queue_size: int = WRITE_TO_FILE_QUEUE_SIZE, **render_kw) -> int:
    """The loop of batch_render_random_camera (batch_render.py:103-115) on the hip backend: the views go to the
    library `batch` at a time (`render_views`), every finished view to the writer process; rendering the next batch
    overlaps encoding the previous one.  Returns the number of views written."""
    from .renderer import render_views
    n = 0
    with FrameWriter(out_dir, queue_size=queue_size) as writer:
        for i in range(0, len(cameras), max(1, int(batch))):
            cams = list(cameras[i:i + batch])
            kw = dict(render_kw)
            if kw.get("overrides") is not None:               # per-view scenes travel with their cameras
                kw["overrides"] = list(kw["overrides"][i:i + batch])
            res = render_views(scene, cams, want_nearest=False, **kw)
            image, depth = _to_host(res["image"]), _to_host(res["depth"])
            for v, cam in enumerate(cams):
                far = cam["far"]
                writer.put(f"_{i + v}", image[v], depth[v], float(far.item() if hasattr(far, "item") else far))
                n += 1
    return n
