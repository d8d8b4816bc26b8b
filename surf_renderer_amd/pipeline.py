"""Frames in flight: the single-process frame pipeline that bench.py times and tests/test_hip_pipeline.py checks.

A frame is three kernels (clear, prep-and-bin, render); the binning is latency-bound, so
`n_inflight` frames run on their own HIP streams with their own scratch and output slab, and the binning of one frame
overlaps the render kernel of another.  Each (output slab, scratch) pair's kernel sequence can be captured once as a
hipGraph and replayed (one host call per frame instead of three launches).  Everything a replay touches is owned by
this object and stays alive with it: the scene buffers, the scratch, the slabs and the graphs.

Reference call it stands for: a loop of ``render(scene)`` calls over one resident scene
(diffrend/torch/batch_render.py:36-53, torch/GAN/gan.py:325-378).
"""
from __future__ import annotations

import sys
from typing import List, Optional, Tuple

import torch

from . import _lib, renderer


def slab_views(slab: torch.Tensor, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rows, 4W) fp32 slab -> image (rows, W, 3) and depth (rows, W) views: [W x rgb | W x depth] per row, so one
    transfer moves both."""
    rows = slab.shape[0]
    off = slab.storage_offset()
    return (slab.as_strided((rows, width, 3), (4 * width, 3, 1), off),
            slab.as_strided((rows, width), (4 * width, 1), off + 3 * width))


class FramePipeline:
    """`n_inflight` frames of one resident scene in flight on their own streams.

    submit(ev)   enqueue the next frame into slab (count % n_inflight): graph replay when graphs are on and no timing
                 event pair is asked for, eager launches otherwise.  Never synchronises the host.
    sync()       wait for everything submitted.
    verify()     render every slab's rows once more, eagerly on the current stream, and compare bit for bit.
    """

    def __init__(self, buf: renderer.SceneBuffers, cam: _lib.SrhCamera, rows: Optional[Tuple[int, int]] = None,
                 n_inflight: int = 3, mode: str = "auto", graphs: bool = True, strict_graphs: bool = False,
                 slabs: Optional[List[torch.Tensor]] = None, schedule: str = "frames", render_streams: int = 2,
                 prioritise_render: bool = True, prioritise_bin: bool = False, rotate: int = 1):
        """``schedule='frames'``: frame k runs whole on stream k % n_inflight.  ``schedule='stages'``: two streams, one
        for every frame's binning kernels and one for every frame's render kernel (``SrhParams.stages``): the
        latency-bound binning of frame k+1 runs beside the render kernel of frame k, render kernels never share the
        GPU with one another, and a frame's (slab, scratch) pair k % n_inflight is reused only after its render."""
        # 'render-only' / 'bin-only': diagnostics -- every stream replays just that half of its frame (the bins of the
        # warm-up frame stay valid: same scene, same camera), to price each half of the pipeline on its own
        if schedule not in ("frames", "stages", "render-only", "bin-only"):
            raise ValueError("schedule must be 'frames' or 'stages'")
        if schedule != "frames" and mode not in ("auto", "binned"):
            raise ValueError("schedule='stages' needs the binned mode")
        self.schedule = schedule
        self.buf, self.cam, self.mode = buf, cam, mode
        self.device = buf.device
        self.width, self.height = renderer.frame_size(cam)
        self.rows = (0, self.height) if rows is None else (int(rows[0]), int(rows[1]))
        h = self.rows[1] - self.rows[0]
        # `rotate` x n_inflight (slab, scratch) pairs: frame k runs on stream k % n_inflight with pair k % n, so a
        # pair always meets the same stream (its reuse is ordered) but comes round only every `rotate` turns
        self.n_streams = int(n_inflight)
        self.n = int(n_inflight) * max(1, int(rotate))
        if rotate != 1 and schedule != "frames":
            raise ValueError("rotate needs schedule='frames'")
        if schedule == "stages":
            # streams[0]: every frame's binning kernels; streams[1:]: the render kernels, frame k on 1 + k % R.  The
            # render streams get the higher priority: binning waves then only take what the render kernels leave.
            self.n_render = max(1, int(render_streams))
            pb, pr = (-1, 0) if prioritise_bin else (0, -1 if prioritise_render else 0)
            self.streams = [torch.cuda.Stream(self.device, priority=pb)] + \
                [torch.cuda.Stream(self.device, priority=pr) for _ in range(self.n_render)]
        else:
            self.n_render = 0
            self.streams = [torch.cuda.Stream(self.device) for _ in range(self.n_streams)]
        self._bin_done = [torch.cuda.Event() for _ in range(self.n)] if schedule == "stages" else []
        self._render_done: List[Optional[torch.cuda.Event]] = [None] * self.n
        self.bin_graphs: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.n
        self.scratch = [buf.new_workspace(self.width, self.height) for _ in range(self.n)]
        self.slabs = slabs if slabs is not None else \
            [torch.empty((h, 4 * self.width), dtype=torch.float32, device=self.device) for _ in range(self.n)]
        if len(self.slabs) != self.n:
            raise ValueError("one output slab per frame in flight")
        self.graphs: List[Optional[torch.cuda.CUDAGraph]] = [None] * self.n
        self.use_graphs = bool(graphs)
        self.count = 0
        self._since_poison: Optional[int] = None
        if self.use_graphs:
            self._capture(strict_graphs)

    def _note(self, b: int, what: str) -> None:
        """A graph replay runs no Python: tell the renderer's bookkeeping what it left in scratch b's bin counters."""
        renderer._ws_note(self.scratch[b], (what, renderer._layout_key(self.buf, self.width, self.height)))

    def _render(self, b: int, ev=None, stages: int = 0) -> None:
        image, depth = slab_views(self.slabs[b], self.width)
        renderer.render_buffers(self.buf, self.cam, rows=self.rows, mode=self.mode, out=(image, depth, None),
                                events=ev, workspace=self.scratch[b], stages=stages)

    def _capture(self, strict: bool) -> None:
        """One hipGraph per (slab, scratch) pair, captured up front on that pair's stream."""
        torch.cuda.synchronize(self.device)

        def capture(stream, b, stages):
            g = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls belong to the capture (a caller may have helper threads)
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                self._render(b, stages=stages)
            return g

        try:
            for b in range(self.n):
                with torch.cuda.stream(self.streams[0 if self.schedule == "stages" else b % self.n_streams]):
                    # warm: module load, allocator -- and the scratch's bin counters: a whole frame leaves them zero, so
                    # the captured frames need no clearing launch.  The two diagnostic schedules replay one half only:
                    # they are captured on a scratch that holds bins (render-only renders them again and again; bin-only
                    # bins over them, so its graph carries the clearing launch).
                    self._render(b, stages=_lib.STAGE_BIN if self.schedule in ("render-only", "bin-only") else 0)
                torch.cuda.synchronize(self.device)
                if self.schedule == "stages":
                    self.bin_graphs[b] = capture(self.streams[0], b, _lib.STAGE_BIN)
                    self.graphs[b] = capture(self.streams[1 + b % self.n_render], b, _lib.STAGE_RENDER)
                else:
                    self.graphs[b] = capture(self.streams[b % self.n_streams], b, self._half())
        except Exception as exc:                                 # capture unsupported here: stay eager, say so
            if strict:
                raise
            print(f"[pipeline] hipGraph capture failed ({exc!r}); eager launches", file=sys.stderr)
            self.graphs = [None] * self.n
            self.bin_graphs = [None] * self.n
            self.use_graphs = False
        torch.cuda.synchronize(self.device)

    @property
    def captured(self) -> int:
        return sum(g is not None for g in self.graphs)

    def submit(self, ev=None) -> int:
        b = self.count % self.n
        self.count += 1
        if self._since_poison is not None:
            self._since_poison += 1
        if self.schedule == "stages":
            sb, sr = self.streams[0], self.streams[1 + (self.count - 1) % self.n_render]
            with torch.cuda.stream(sb):
                if self._render_done[b] is not None:
                    sb.wait_event(self._render_done[b])          # this pair's previous frame has been rendered
                if self.bin_graphs[b] is not None:
                    self.bin_graphs[b].replay()
                    self._note(b, "binned")
                else:
                    self._render(b, stages=_lib.STAGE_BIN)
                self._bin_done[b].record(sb)
            with torch.cuda.stream(sr):
                sr.wait_event(self._bin_done[b])
                if self.graphs[b] is not None and ev is None:
                    self.graphs[b].replay()
                    self._note(b, "clean")
                else:
                    self._render(b, ev, stages=_lib.STAGE_RENDER)
                if self._render_done[b] is None:
                    self._render_done[b] = torch.cuda.Event()
                self._render_done[b].record(sr)
            return b
        g = self.graphs[b] if ev is None else None
        with torch.cuda.stream(self.streams[b % self.n_streams]):
            if g is not None:
                g.replay()
            else:
                self._render(b, ev, stages=self._half())
        return b

    def _half(self) -> int:
        return {"render-only": _lib.STAGE_RENDER | _lib.STAGE_KEEP_BINS, "bin-only": _lib.STAGE_BIN}.get(self.schedule, 0)

    def sync(self) -> None:
        for s in self.streams:
            s.synchronize()

    def poison(self) -> None:
        """Overwrite every output slab with NaN bit patterns (and wait): a later verify() then proves that the frames
        submitted in between really wrote their slabs -- a replay that silently did nothing would leave the poison."""
        self.sync()
        for t in self.slabs:
            t.view(torch.int32).fill_(-1)
        torch.cuda.synchronize(self.device)
        self._since_poison = 0

    def verify(self) -> int:
        """Every slab rendered into (since the last poison(), if any) equals a fresh eager render of the same rows,
        bit for bit.  Returns the number of slabs compared; raises on the first mismatch."""
        self.sync()
        ref = torch.empty_like(self.slabs[0])
        image, depth = slab_views(ref, self.width)
        renderer.render_buffers(self.buf, self.cam, rows=self.rows, mode=self.mode, out=(image, depth, None))
        torch.cuda.synchronize(self.device)
        done = self.count if self._since_poison is None else self._since_poison
        n = min(self.n, done)
        for b in [(self.count - 1 - i) % self.n for i in range(n)]:
            if not torch.equal(self.slabs[b].view(torch.int32), ref.view(torch.int32)):
                bad = int((self.slabs[b].view(torch.int32) != ref.view(torch.int32)).sum())
                raise RuntimeError(f"frame slab {b} ({'graph replay' if self.graphs[b] is not None else 'eager'} on "
                                   f"stream {b}) differs from the eager render in {bad} words")
        return n
