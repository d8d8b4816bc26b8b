"""The hip backend's mirror of the reference backend interface: ``render(scene, **params) -> dict``.

Same call and same scene dict as ``diffrend.numpy.renderer.render`` (numpy/renderer.py:204-272) and
``diffrend.torch.renderer.render`` (torch/renderer.py:136-355); the work is done by hand-written
gfx950 kernels behind the C ABI of include/srh.h.  PyTorch is used for device memory and streams only.

Layers
  flatten_scene()   expanded scene dict (lists / ndarrays / tensors) -> SceneBuffers: contiguous fp32 /
                    int32 device arrays in the reference's layouts and concatenation order, plus the
                    ctypes descriptors libsrh consumes.  Done once per scene; stays resident in HBM.
  render_buffers()  one frame from resident buffers and a camera (what bench.py times).
  render()          the drop-in: flatten + render_buffers + reference-shaped result dict.

There is no CPU fallback: without a GPU or without libsrh.so these functions raise.
"""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .scene import PRIM_CODE, _OBJ_FIELDS, unit_up

# keyword arguments of the torch backend's render() that callers pass routinely (torch/renderer.py:152-168, 233-245,
# 291, 326-327); the hip backend accepts them so call sites need no edits.  What each does here:
#   tiled, tile_size        memory knobs of the reference's (pixels x primitives) arrays: no effect on any output
#   backface_culling        the reference only LABELS primitives (torch/utils.py:515-536) and never reads the labels:
#                           every output is unchanged (oracle/check_ref_kwargs.py ran the reference both ways)
#   vis_stat                True raises in the reference ('Removed Support for vis_stat', :235) and here
#   norm_depth_image_only   `image` becomes the normalised depth of :245-249 (see render())
#   shadow                  shadow rays (:291-314)
_TORCH_ONLY_KWARGS = {"tiled", "tile_size", "backface_culling", "norm_depth_image_only", "vis_stat", "shadow"}


def _require_gpu(device: torch.device) -> None:
    if device.type != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("the hip backend needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback -- use the reference's numpy backend instead")


def _as_tensor(x, dtype: torch.dtype, device: torch.device, keep_graph: bool = False) -> torch.Tensor:
    """Contiguous tensor of ``x`` in ``dtype``: on ``device`` if ``x`` already lives there or takes part in autograd
    (with ``keep_graph`` a tensor that requires grad stays attached to the graph -- casts and copies are
    differentiable -- so gradients reach the caller's leaf); otherwise still in host memory, for ``_upload`` to send
    with everything else in one transfer."""
    if isinstance(x, torch.Tensor):
        t = x if (keep_graph and x.requires_grad) else x.detach()
        if t.is_cuda or t.requires_grad:
            return t.to(device=device, dtype=dtype).contiguous()
        x = t.numpy()
    # host leaves are converted and packed with numpy (plain single-threaded copies): torch's CPU operators go through
    # its OpenMP pool, which on a box with fewer cores than threads costs milliseconds per frame while the GPU runs
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=_NP_DTYPE[dtype]))


_NP_DTYPE = {torch.float32: np.float32, torch.int32: np.int32}
_STAGING: Dict[torch.device, Tuple[torch.Tensor, torch.cuda.Event]] = {}
_STAGING_LOCK = threading.Lock()
_UPLOAD_ALIGN = 256


def _upload(tensors: Dict[str, torch.Tensor], device: torch.device) -> None:
    """Move every host tensor of ``tensors`` to the device in ONE host-to-device copy: the leaves are packed into a
    pinned staging buffer (kept per device, guarded by an event so that a new frame's packing waits for the previous
    frame's copy) and the device side is carved into typed views.  A scene is ~10 small arrays; sent one by one from
    pageable memory each costs a synchronous copy of ~0.6 ms, which was all of render(scene)'s time."""
    host = [(k, t) for k, t in tensors.items() if not t.is_cuda]
    if not host:
        return
    offsets, total = [], 0
    for _, t in host:
        offsets.append(total)
        total += -(-t.numel() * t.element_size() // _UPLOAD_ALIGN) * _UPLOAD_ALIGN
    total = max(total, _UPLOAD_ALIGN)
    with _STAGING_LOCK:
        _upload_locked(tensors, device, host, offsets, total)


def _upload_locked(tensors, device, host, offsets, total) -> None:
    entry = _STAGING.get(device)
    if entry is not None:
        entry[1].synchronize()
    if entry is None or entry[0].numel() < total:
        entry = (torch.empty(max(total, 1 << 20), dtype=torch.uint8).pin_memory(), torch.cuda.Event())
        _STAGING[device] = entry
    staging, done = entry
    staging_np = staging.numpy()
    for (_, t), off in zip(host, offsets):
        n = t.numel() * t.element_size()
        if n:
            staging_np[off:off + n] = t.numpy().reshape(-1).view(np.uint8)
    packed = torch.empty(total, dtype=torch.uint8, device=device)
    packed.copy_(staging[:total], non_blocking=True)
    done.record(torch.cuda.current_stream(device))
    for (k, t), off in zip(host, offsets):
        n = t.numel() * t.element_size()
        tensors[k] = packed[off:off + n].view(t.dtype).reshape(t.shape)


def _host_view(x) -> Optional[np.ndarray]:
    """numpy view for host-side validation; None for device tensors (not worth a sync)."""
    if isinstance(x, torch.Tensor):
        return None if x.is_cuda else x.detach().numpy()
    return np.asarray(x)


@dataclass
class SceneBuffers:
    """A scene resident in HBM in the layouts of include/srh.h."""
    device: torch.device
    kinds: List[str]
    counts: List[int]
    tensors: Dict[str, torch.Tensor]          # "<kind>.<field>", "lights.pos", ... (keeps memory alive)
    objects: _lib.SrhObjects
    lights: _lib.SrhLights
    materials: _lib.SrhMaterials
    gamma: Optional[float]
    workspace: Optional[torch.Tensor] = None   # per-frame scratch, sized for the largest frame seen so far
    workspace_frame: Tuple[int, int] = (0, 0)
    total: int = 0
    shadow_workspace: Optional[torch.Tensor] = None   # scratch of the accelerated shadow pass (light views)

    def ensure_workspace(self, width: int, height: int) -> torch.Tensor:
        """Device scratch for libsrh (primitive records + tile bins) at ``width x height``."""
        if self.workspace is None or width > self.workspace_frame[0] or height > self.workspace_frame[1]:
            lib = _lib.load()
            w, h = max(width, self.workspace_frame[0]), max(height, self.workspace_frame[1])
            need = lib.srh_workspace_bytes(C.byref(self.objects), w, h)
            if need == 0:
                raise _lib.SrhError(-1, lib.srh_last_error().decode())
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
            self.workspace_frame = (w, h)
        return self.workspace

    def ensure_shadow_workspace(self, width: int, height: int) -> torch.Tensor:
        """Scratch of the accelerated shadow pass: room for the light views (tile bins in every light's screen space)
        behind the primary frame's scratch."""
        lib = _lib.load()
        need = lib.srh_shadow_workspace_bytes(C.byref(self.objects), width, height, self.lights.n_lights)
        if need == 0:
            raise _lib.SrhError(-2, lib.srh_last_error().decode())
        if self.shadow_workspace is None or self.shadow_workspace.numel() < need:
            self.shadow_workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self.shadow_workspace

    def new_workspace(self, width: int, height: int) -> torch.Tensor:
        """An additional scratch buffer (one per frame in flight when frames are pipelined over several streams)."""
        lib = _lib.load()
        need = lib.srh_workspace_bytes(C.byref(self.objects), width, height)
        if need == 0:
            raise _lib.SrhError(-1, lib.srh_last_error().decode())
        return torch.empty(need, dtype=torch.uint8, device=self.device)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.tensors.values())


def _check_w(name: str, arr: Optional[np.ndarray], want: float) -> None:
    if arr is None or arr.size == 0:
        return
    w = arr[..., 3]
    if not np.all(w == want):
        raise ValueError(f"{name}: homogeneous w must be {want:g} for every row (the reference's convention, "
                         f"docs/scene_description.md:3-5); found {np.unique(w)[:4]}")


def flatten_scene(scene: Dict[str, Any], device="cuda", validate: bool = True, keep_graph: bool = False) -> SceneBuffers:
    """Upload an expanded scene.  Object batches keep scene['objects'] dict order, which defines the
    global primitive numbering (numpy/renderer.py:172-201).  The caller's scene is not modified.
    ``keep_graph`` keeps tensors that require grad attached to autograd (see ``render``)."""
    device = torch.device(device)
    _require_gpu(device)
    if device.type == "cuda" and device.index is None:
        # "cuda" means the current device; tensors report "cuda:N", and the out-buffer checks compare devices
        device = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.load()
    objs = scene["objects"]
    if not objs:
        raise ValueError("scene['objects'] is empty")
    if len(objs) > _lib.MAX_SEGMENTS:
        raise ValueError(f"at most {_lib.MAX_SEGMENTS} object batches")
    f32, i32 = torch.float32, torch.int32
    tensors: Dict[str, torch.Tensor] = {}
    kinds: List[str] = []
    counts: List[int] = []
    ob = _lib.SrhObjects()
    n_mat = int(np.asarray(_shape_of(scene["materials"]["albedo"]))[0])
    for s, (kind, grp) in enumerate(objs.items()):
        if kind not in PRIM_CODE:
            raise ValueError(f"unknown object type {kind!r}; expanded scenes hold disk / plane / sphere / "
                             f"triangle (use surf_renderer_amd.scene.load_scene for JSON 'obj' lists)")
        seg = ob.seg[s]
        seg.type = PRIM_CODE[kind]
        count = None
        for name in _OBJ_FIELDS[kind]:
            t = _as_tensor(grp[name], f32, device, keep_graph)
            if name == "radius":
                t = t.reshape(-1)
            elif name == "face":
                t = t.reshape(-1, 3, 4)
            else:
                t = t.reshape(-1, 4)
            if count is None:
                count = t.shape[0]
            elif t.shape[0] != count:
                raise ValueError(f"{kind}.{name}: {t.shape[0]} rows, expected {count}")
            if validate:
                host = _host_view(grp[name])
                if name in ("pos", "face"):
                    _check_w(f"{kind}.{name}", None if host is None else host.reshape(-1, 4), 1.0)
                elif name == "normal":
                    _check_w(f"{kind}.{name}", None if host is None else host.reshape(-1, 4), 0.0)
            tensors[f"{kind}.{name}"] = t
        mi_host = _host_view(grp["material_idx"])
        if validate and mi_host is not None and mi_host.size:
            if mi_host.min() < 0 or mi_host.max() >= n_mat:
                raise IndexError(f"{kind}.material_idx out of range for {n_mat} materials")
        mi = _as_tensor(grp["material_idx"], i32, device).reshape(-1)
        if mi.shape[0] != count:
            raise ValueError(f"{kind}.material_idx: {mi.shape[0]} entries, expected {count}")
        if count == 0:
            raise ValueError(f"{kind}: empty batch")
        tensors[f"{kind}.material_idx"] = mi
        seg.count = count
        kinds.append(kind)
        counts.append(count)
    ob.n_segments = len(kinds)

    lights = scene["lights"]
    lpos = _as_tensor(lights["pos"], f32, device, keep_graph).reshape(-1, 4)
    lidx = _as_tensor(lights["color_idx"], i32, device).reshape(-1)
    colors = _as_tensor(scene["colors"], f32, device, keep_graph).reshape(-1, 3)
    albedo = _as_tensor(scene["materials"]["albedo"], f32, device, keep_graph).reshape(-1, 3)
    if lpos.shape[0] != lidx.shape[0]:
        raise ValueError("lights.pos and lights.color_idx disagree on the number of lights")
    if lpos.shape[0] > _lib.MAX_LIGHTS:
        raise ValueError(f"at most {_lib.MAX_LIGHTS} lights")
    if validate:
        _check_w("lights.pos", _host_view(lights["pos"]), 1.0)
        ci = _host_view(lights["color_idx"])
        if ci is not None and ci.size and (ci.min() < 0 or ci.max() >= colors.shape[0]):
            raise IndexError("lights.color_idx out of range for the colour table")
    tensors.update({"lights.pos": lpos, "lights.color_idx": lidx, "colors": colors, "materials.albedo": albedo})
    # inputs of the torch backend's shading model only (ignored by the numpy one, numpy/renderer.py:234-255)
    if "attenuation" in lights:
        att = _as_tensor(lights["attenuation"], f32, device, keep_graph).reshape(-1, 3)
        if att.shape[0] != lpos.shape[0]:
            raise ValueError("lights.attenuation must have one (kc, kl, kq) row per light")
        tensors["lights.attenuation"] = att
    if "ambient" in lights:
        tensors["lights.ambient"] = _as_tensor(lights["ambient"], f32, device, keep_graph).reshape(3)
    if "coeffs" in scene["materials"]:
        cfs = _as_tensor(scene["materials"]["coeffs"], f32, device, keep_graph).reshape(-1, 3)
        if cfs.shape[0] != albedo.shape[0]:
            raise ValueError("materials.coeffs must have one row per material")
        tensors["materials.coeffs"] = cfs

    _upload(tensors, device)
    for s, kind in enumerate(kinds):
        for name in _OBJ_FIELDS[kind] + ("material_idx",):
            setattr(ob.seg[s], name, tensors[f"{kind}.{name}"].data_ptr())
    ls = _lib.SrhLights(n_lights=lpos.shape[0], n_colors=colors.shape[0], pos=tensors["lights.pos"].data_ptr(),
                        color_idx=tensors["lights.color_idx"].data_ptr(), colors=tensors["colors"].data_ptr())
    ms = _lib.SrhMaterials(n_materials=albedo.shape[0], albedo=tensors["materials.albedo"].data_ptr())
    if "lights.attenuation" in tensors:
        ls.attenuation = tensors["lights.attenuation"].data_ptr()
    if "lights.ambient" in tensors:
        ls.ambient = tensors["lights.ambient"].data_ptr()
    if "materials.coeffs" in tensors:
        ms.coeffs = tensors["materials.coeffs"].data_ptr()

    gamma = None
    if "tonemap" in scene:
        tm = scene["tonemap"]
        if tm.get("type", "gamma") != "gamma":
            raise ValueError(f"tonemap type {tm.get('type')!r}: only 'gamma' exists (numpy/renderer.py:140-142)")
        g = tm["gamma"]
        gamma = float(g.detach().cpu().reshape(-1)[0]) if isinstance(g, torch.Tensor) else float(np.ravel(g)[0])

    return SceneBuffers(device=device, kinds=kinds, counts=counts, tensors=tensors, objects=ob, lights=ls,
                        materials=ms, gamma=gamma, total=sum(counts))


def _shape_of(x):
    if isinstance(x, torch.Tensor):
        return tuple(x.shape)
    return np.asarray(x).shape


def camera_struct(camera: Dict[str, Any], shading: str = "numpy") -> _lib.SrhCamera:
    """scene['camera'] -> SrhCamera.  List-typed ``at`` / ``up`` take the reference's float32 detour
    (numpy/ops.py:95-100, quirk Q11), which for the numpy backend's semantics includes normalising ``up`` in
    float32; arrays and tensors are taken at full precision."""
    def vec(val, f32_if_list: bool):
        if isinstance(val, torch.Tensor):
            return val.detach().cpu().double().numpy().reshape(-1)
        if f32_if_list and isinstance(val, (list, tuple)):
            return np.asarray(val, dtype=np.float32).astype(np.float64).reshape(-1)
        return np.asarray(val, dtype=np.float64).reshape(-1)

    def scalar(val) -> float:
        if isinstance(val, torch.Tensor):
            return float(val.detach().cpu().reshape(-1)[0])
        return float(np.ravel(val)[0])

    cam = _lib.SrhCamera()
    eye, at, up = vec(camera["eye"], False), vec(camera["at"], True), vec(camera["up"], True)
    if shading == "torch":
        # the torch backend holds all three as float32 tensors (make_torch_var, torch/render.py:81-100)
        eye, at, up = (v.astype(np.float32).astype(np.float64) for v in (eye, at, up))
    if up.size == 3:
        up = np.append(up, 0.0)
    if eye.size != 4 or at.size != 4 or up.size != 4:
        raise ValueError("camera.eye / camera.at must be homogeneous 4-vectors, camera.up a 3- or 4-vector")
    if shading == "numpy" and isinstance(camera["up"], (list, tuple)):
        # the reference normalises a list-typed up in float32 (numpy/ops.py:99,109): hand over the finished y axis
        with np.errstate(all="ignore"):
            unit = unit_up(camera["up"], up)
        if np.all(np.isfinite(unit)):
            up = np.append(unit, 0.0)
            cam.up_is_unit = 1
    cam.eye[:] = eye.tolist()
    cam.at[:] = at.tolist()
    cam.up[:] = up.tolist()
    cam.fovy = scalar(camera["fovy"])
    cam.focal_length = scalar(camera["focal_length"])
    cam.near_clip = scalar(camera["near"])
    cam.far_clip = scalar(camera["far"])
    vp = [int(v) for v in np.ravel(_host_view(camera["viewport"]) if not isinstance(camera["viewport"], torch.Tensor)
                                   else camera["viewport"].cpu().numpy())]
    cam.viewport[:] = vp
    proj = str(camera.get("proj_type", "perspective"))
    if proj in ("ortho", "orthographic"):
        cam.ortho = 1                    # torch/utils.py:461: only the torch backend's semantics have it
    elif proj not in ("persp", "perspective"):
        raise ValueError(f"camera.proj_type {proj!r}: expected 'perspective' or 'ortho'")
    return cam


def frame_size(cam: _lib.SrhCamera) -> Tuple[int, int]:
    return cam.viewport[2] - cam.viewport[0], cam.viewport[3] - cam.viewport[1]


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# What a workspace's bin counters hold is known only to whoever used it last, so it is noted ON the tensor object (the
# note dies with it; a new tensor over recycled memory starts unknown): ("clean", layout) after a binned frame whose
# render kernel left every counter at zero, ("binned", layout) between a frame's two stages.  A clean workspace needs
# no clearing launch (SrhParams.counters_clean): a frame is then two kernels.
def _ws_state(ws: torch.Tensor):
    return getattr(ws, "_srh_state", None)


def _ws_note(ws: torch.Tensor, state) -> None:
    try:
        ws._srh_state = state
    except AttributeError:                                  # a tensor type that takes no attributes: always unknown
        pass


def _layout_key(buf: "SceneBuffers", width: int, height: int, what="frame"):
    ob = buf.objects
    return (what, tuple((ob.seg[s].type, ob.seg[s].count) for s in range(ob.n_segments)), int(width), int(height))


def render_buffers(buf: SceneBuffers, cam: _lib.SrhCamera, rows: Optional[Tuple[int, int]] = None,
                   mode: str = "auto", out: Optional[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]] = None,
                   want_nearest: bool = True, events: Optional[_lib.EventPair] = None,
                   workspace: Optional[torch.Tensor] = None, shading: str = "numpy", double_sided: bool = False,
                   use_quartic: bool = False, aux: Optional[Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]] = None,
                   waves_per_tile: int = 0, stages: int = 0):
    """One frame (or the row slab ``rows=(r0, r1)`` of it) from resident buffers.  Everything is
    enqueued on the current stream of ``buf.device``; nothing synchronises.  ``out`` may supply
    preallocated (image (h,W,3) f32, depth (h,W) f32, nearest (h,W) i32 or None).  ``workspace`` overrides the
    buffers' own scratch: frames in flight on different streams each need their own (``new_workspace``).
    ``stages`` (``_lib.STAGE_BIN`` / ``_lib.STAGE_RENDER``, 0 = both) splits a binned frame into its binning kernels and its
    render kernel, for a caller that runs them on two streams (``pipeline.FramePipeline``).
    ``shading='torch'`` selects the torch backend's semantics (Phong with attenuation / specular / ambient,
    ``double_sided``, ``use_quartic``, orthonormal camera, far+1 background); ``aux=(normal, pos)`` are optional
    dense (h,W,3) f32 outputs."""
    lib = _lib.load()
    width, height = frame_size(cam)
    r0, r1 = (0, height) if rows is None else (int(rows[0]), int(rows[1]))
    h = r1 - r0
    if out is None:
        image = torch.empty((max(h, 0), width, 3), dtype=torch.float32, device=buf.device)
        depth = torch.empty((max(h, 0), width), dtype=torch.float32, device=buf.device)
        nearest = torch.empty((max(h, 0), width), dtype=torch.int32, device=buf.device) if want_nearest else None
    else:
        image, depth, nearest = out
        for t, shape, dt in ((image, (h, width, 3), torch.float32), (depth, (h, width), torch.float32),
                             (nearest, (h, width), torch.int32)):
            if t is None:
                continue
            inner = tuple(t.stride()[1:]) == ((3, 1) if len(shape) == 3 else (1,))
            if tuple(t.shape) != shape or t.dtype != dt or not inner or t.device != buf.device:
                raise ValueError(f"out buffer mismatch: want {dt} {shape} with dense rows on {buf.device}, "
                                 f"got {t.dtype} {tuple(t.shape)} strides {t.stride()} on {t.device}")
    params = _lib.SrhParams(row0=r0, row1=r1, mode=_lib.MODES[mode],
                            tonemap_gamma=0 if buf.gamma is None else 1,
                            gamma=1.0 if buf.gamma is None else buf.gamma,
                            shading=_lib.SHADING[shading], double_sided=int(bool(double_sided)),
                            use_quartic=int(bool(use_quartic)), waves_per_tile=int(waves_per_tile),
                            normal_out=aux[0].data_ptr() if aux and aux[0] is not None else None,
                            pos_out=aux[1].data_ptr() if aux and aux[1] is not None else None,
                            image_row_stride=image.stride(0) if h > 1 else 0,
                            depth_row_stride=depth.stride(0) if h > 1 else 0,
                            nearest_row_stride=nearest.stride(0) if (nearest is not None and h > 1) else 0,
                            ev_start=events.start if events else None, ev_stop=events.stop if events else None,
                            stages=int(stages))
    if workspace is None:
        workspace = buf.ensure_workspace(width, height)
    binned = mode in ("auto", "binned") and not cam.ortho
    after = None
    if binned:
        key = _layout_key(buf, width, height)
        halves = int(stages) & (_lib.STAGE_BIN | _lib.STAGE_RENDER) or (_lib.STAGE_BIN | _lib.STAGE_RENDER)
        state = _ws_state(workspace)
        if halves & _lib.STAGE_BIN:
            params.counters_clean = int(state == ("clean", key))
        elif state != ("binned", key):
            raise ValueError("stages=STAGE_RENDER needs the bins of a stages=STAGE_BIN call with the same scene and "
                             "frame size in this workspace (its last use left it " +
                             ("without any" if state is None else f"{state[0]}") + ")")
        after = ("binned", key) if (not halves & _lib.STAGE_RENDER or int(stages) & _lib.STAGE_KEEP_BINS) else ("clean", key)
        _ws_note(workspace, None)                           # unknown until the call has been accepted
    with torch.cuda.device(buf.device):
        rc = lib.srh_render_fwd(C.byref(cam), C.byref(buf.objects), C.byref(buf.lights), C.byref(buf.materials),
                                C.byref(params), workspace.data_ptr(), workspace.numel(),
                                image.data_ptr(), depth.data_ptr(),
                                nearest.data_ptr() if nearest is not None else None, _stream_ptr(buf.device))
    _lib.check(rc)
    if binned:
        _ws_note(workspace, after)
    return image, depth, nearest


def bin_statistics(buf: SceneBuffers, cam: _lib.SrhCamera, rows: Optional[Tuple[int, int]] = None) -> Dict[str, Any]:
    """What one binned frame really tests (measurement; synchronises).  Runs the frame's binning stage alone on a scratch
    workspace of its own and reads the list lengths back: ``entries`` (batches, tile rows, tile columns) = candidates of
    every 16 x 16-pixel tile's bins, ``wide`` (batches,) = primitives on the frame-wide lists that every tile tests,
    ``executed_pair_tests`` = sum over tiles of (bin entries + frame-wide entries) x 256 pixels -- the (pixel, primitive)
    pairs that go through the fp32 reject test, against ``algorithmic_pair_tests`` = primitives x pixels that the
    reference evaluates -- and ``tile_row_cost``, the per-tile-row sums ``dist.cost_weighted_slabs`` partitions."""
    lib = _lib.load()
    width, height = frame_size(cam)
    r0, r1 = (0, height) if rows is None else (int(rows[0]), int(rows[1]))
    ws = buf.new_workspace(width, height)
    image = torch.empty((r1 - r0, width, 3), dtype=torch.float32, device=buf.device)
    depth = torch.empty((r1 - r0, width), dtype=torch.float32, device=buf.device)
    render_buffers(buf, cam, rows=(r0, r1), mode="binned", out=(image, depth, None), workspace=ws, stages=_lib.STAGE_BIN)
    off, tx, ty, pad, cap = C.c_size_t(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(lib.srh_bin_counters(C.byref(buf.objects), width, height, r0, r1, C.byref(off), C.byref(tx), C.byref(ty),
                                    C.byref(pad), C.byref(cap)))
    nseg = buf.objects.n_segments
    words = ws[off.value:off.value + 4 * (64 + nseg * pad.value)].view(torch.int32).cpu().numpy().astype(np.int64)
    which = int(words[9]) & 1
    wide = np.minimum(words[4 * which:4 * which + nseg], np.asarray(buf.counts, dtype=np.int64))
    bins = words[64:64 + nseg * pad.value].reshape(nseg, pad.value)[:, :tx.value * ty.value]
    entries = np.minimum(bins, cap.value).reshape(nseg, ty.value, tx.value)
    per_tile = entries.sum(axis=0) + int(wide.sum())
    return {"entries": entries, "wide": wide, "bin_capacity": cap.value,
            "executed_pair_tests": int(per_tile.sum()) * 256,
            "algorithmic_pair_tests": int(buf.total) * (r1 - r0) * width,
            "tile_row_cost": per_tile.sum(axis=1)}


def generate_rays(camera: Dict[str, Any], device="cuda", rows: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """``ray_dir`` as the reference returns it: (4, N) unit directions, row-major over the image
    (numpy/renderer.py:145-169)."""
    device = torch.device(device)
    _require_gpu(device)
    lib = _lib.load()
    cam = camera_struct(camera)
    width, height = frame_size(cam)
    r0, r1 = (0, height) if rows is None else rows
    out = torch.empty((4, max(r1 - r0, 0) * width), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.srh_generate_rays(C.byref(cam), r0, r1, out.data_ptr(), _stream_ptr(device)))
    return out


_TORCH_SHADING_KEYS = ("materials.coeffs", "lights.attenuation", "lights.ambient")


def _float_keys(buf: SceneBuffers, shading: str = "numpy") -> List[str]:
    """Keys of buf.tensors that are differentiable inputs, in a fixed order (the torch shading model adds its own
    inputs where the scene has them)."""
    keys = [f"{kind}.{name}" for kind in buf.kinds for name in _OBJ_FIELDS[kind]]
    keys += ["lights.pos", "colors", "materials.albedo"]
    if shading == "torch":
        keys += [k for k in _TORCH_SHADING_KEYS if k in buf.tensors]
    return keys


def shadow_pass(buf: SceneBuffers, cam: _lib.SrhCamera, rows, image: torch.Tensor, depth: torch.Tensor,
                nearest: torch.Tensor, double_sided: bool = False, use_quartic: bool = False,
                all_pairs: bool = False) -> torch.Tensor:
    """The torch backend's ``shadow=True`` (torch/renderer.py:291-314) over a frame rendered with
    ``shading='torch'``: re-shades ``image`` in place with per-light visibility from shadow rays and returns the
    (rows, W) int64 visibility bit field (bit l = light l visible).  Candidates come from tile bins in each light's
    screen space; ``all_pairs=True`` runs the reference's O(pixels x lights x primitives) loop instead (same result)."""
    lib = _lib.load()
    width, height = frame_size(cam)
    r0, r1 = (0, height) if rows is None else (int(rows[0]), int(rows[1]))
    vis = torch.empty((r1 - r0, width), dtype=torch.int64, device=buf.device)
    params = _lib.SrhParams(row0=r0, row1=r1, mode=_lib.MODES["exact" if all_pairs else "auto"],
                            tonemap_gamma=0 if buf.gamma is None else 1,
                            gamma=1.0 if buf.gamma is None else buf.gamma, shading=_lib.SHADING["torch"],
                            double_sided=int(bool(double_sided)), use_quartic=int(bool(use_quartic)))
    if all_pairs:
        workspace = buf.ensure_workspace(width, height)
    else:
        workspace = buf.ensure_shadow_workspace(width, height)
    with torch.cuda.device(buf.device):
        _lib.check(lib.srh_shadow_shade(C.byref(cam), C.byref(buf.objects), C.byref(buf.lights), C.byref(buf.materials),
                                        C.byref(params), workspace.data_ptr(), workspace.numel(), nearest.data_ptr(),
                                        depth.data_ptr(), image.data_ptr(), vis.data_ptr(), _stream_ptr(buf.device)))
    return vis


class _RenderFunction(torch.autograd.Function):
    """render_buffers with the analytic backward of libsrh (srh_render_bwd).  Gradient semantics are those of
    autograd through the reference's torch backend (SURVEY.md section 8, row a-B): selection and masks are piecewise
    constant; a disc's radius and a triangle's vertices 1, 2 receive zero gradient."""

    @staticmethod
    def forward(ctx, buf, cam, rows, mode, shade, *inputs):
        # shade = (shading, double_sided, use_quartic, shadow)
        image, depth, nearest = render_buffers(buf, cam, rows=rows, mode=mode, shading=shade[0],
                                               double_sided=shade[1], use_quartic=shade[2])
        vis = shadow_pass(buf, cam, rows, image, depth, nearest, shade[1], shade[2]) if shade[3] else None
        ctx.buf, ctx.cam, ctx.rows, ctx.mode, ctx.shade = buf, cam, rows, mode, shade
        ctx.has_vis = vis is not None
        if vis is not None:
            ctx.save_for_backward(depth, nearest, vis)
        else:
            ctx.save_for_backward(depth, nearest)
        ctx.mark_non_differentiable(nearest)
        return image, depth, nearest

    @staticmethod
    def backward(ctx, g_image, g_depth, _g_nearest):
        if ctx.has_vis:
            depth, nearest, vis = ctx.saved_tensors
        else:
            (depth, nearest), vis = ctx.saved_tensors, None
        keys = _float_keys(ctx.buf, ctx.shade[0])
        grads = _render_backward(ctx.buf, ctx.cam, ctx.rows, ctx.mode, ctx.shade, depth, nearest, vis, g_image, g_depth,
                                 ctx.needs_input_grad[5:])
        return (None, None, None, None, None) + tuple(grads.get(k) for k in keys)


def _render_backward(buf: SceneBuffers, cam: _lib.SrhCamera, rows, mode: str, shade, depth: torch.Tensor,
                     nearest: torch.Tensor, vis: Optional[torch.Tensor], g_image: Optional[torch.Tensor],
                     g_depth: Optional[torch.Tensor], need: Sequence[bool]) -> Dict[str, torch.Tensor]:
    """srh_render_bwd: gradients of the inputs named by ``_float_keys`` (those with ``need``) for the upstream gradients
    of image and depth, from the winners the forward pass saved.  Everything is enqueued on the current stream."""
    lib = _lib.load()
    width, height = frame_size(cam)
    r0, r1 = (0, height) if rows is None else (int(rows[0]), int(rows[1]))
    keys = _float_keys(buf, shade[0])
    grads: Dict[str, torch.Tensor] = {}
    sg = _lib.SrhGrads()
    for key, want in zip(keys, need):
        if not want:
            continue
        g = torch.zeros_like(buf.tensors[key])
        grads[key] = g
        if key == "lights.pos":
            sg.lights_pos = g.data_ptr()
        elif key == "colors":
            sg.colors = g.data_ptr()
        elif key == "materials.albedo":
            sg.albedo = g.data_ptr()
        elif key == "materials.coeffs":
            sg.coeffs = g.data_ptr()
        elif key == "lights.attenuation":
            sg.attenuation = g.data_ptr()
        elif key == "lights.ambient":
            sg.ambient = g.data_ptr()
        else:
            kind, name = key.split(".")
            s = buf.kinds.index(kind)
            if name == "radius" and kind == "disk":
                continue                      # identically zero (numpy/renderer.py:88: the radius only feeds a mask)
            getattr(sg, name)[s] = g.data_ptr()
    g_image = g_image.to(torch.float32).contiguous() if g_image is not None else \
        torch.zeros((r1 - r0, width, 3), dtype=torch.float32, device=buf.device)
    g_depth = g_depth.to(torch.float32).contiguous() if g_depth is not None else None
    params = _lib.SrhParams(row0=r0, row1=r1, mode=_lib.MODES[mode],
                            tonemap_gamma=0 if buf.gamma is None else 1,
                            gamma=1.0 if buf.gamma is None else buf.gamma,
                            shading=_lib.SHADING[shade[0]], double_sided=int(bool(shade[1])),
                            use_quartic=int(bool(shade[2])),
                            visibility=vis.data_ptr() if vis is not None else None)
    workspace = buf.ensure_workspace(width, height)
    with torch.cuda.device(buf.device):
        rc = lib.srh_render_bwd(C.byref(cam), C.byref(buf.objects), C.byref(buf.lights), C.byref(buf.materials),
                                C.byref(params), workspace.data_ptr(), workspace.numel(),
                                g_image.data_ptr(), g_depth.data_ptr() if g_depth is not None else None,
                                nearest.data_ptr(), depth.data_ptr(), C.byref(sg), _stream_ptr(buf.device))
    _lib.check(rc)
    return grads


_OVERRIDE_FIELDS = {"lights.pos": ("lights", "pos"), "lights.color_idx": ("lights", "color_idx"),
                    "colors": ("lights", "colors"), "lights.attenuation": ("lights", "attenuation"),
                    "lights.ambient": ("lights", "ambient"), "materials.albedo": ("materials", "albedo"),
                    "materials.coeffs": ("materials", "coeffs")}


class ViewScenes:
    """One scene per view, as srh_render_views takes it (SrhParams.per_view): arrays of SrhObjects / SrhLights /
    SrhMaterials that equal the base scene's except for the pointers a view overrides.  ``overrides[v]`` maps flat leaf
    names (``"disk.pos"``, ``"disk.normal"``, ``"lights.pos"``, ``"colors"``, ``"materials.albedo"``, ...: the keys of
    ``SceneBuffers.tensors``) to arrays or tensors of the base leaf's shape -- what the reference's batch loop assigns
    per element before each ``render()`` (diffrend/torch/GAN/gan.py:325-378: ``disk.pos``, ``disk.normal``,
    ``lights.pos``).  float32 contiguous tensors on the device are used in place; everything else is converted."""

    def __init__(self, buf: SceneBuffers, overrides: Sequence[Dict[str, Any]]):
        n = len(overrides)
        self.n, self.mask, self.keep = n, 0, []
        self.objects = (_lib.SrhObjects * n)(*[_lib.SrhObjects.from_buffer_copy(buf.objects) for _ in range(n)])
        self.lights = (_lib.SrhLights * n)(*[_lib.SrhLights.from_buffer_copy(buf.lights) for _ in range(n)])
        self.materials = (_lib.SrhMaterials * n)(*[_lib.SrhMaterials.from_buffer_copy(buf.materials) for _ in range(n)])
        for v, ov in enumerate(overrides):
            for key, val in (ov or {}).items():
                base = buf.tensors.get(key)
                if base is None:
                    raise KeyError(f"view {v}: {key!r} is not a leaf of this scene (leaves: {sorted(buf.tensors)})")
                t = _as_tensor(val, base.dtype, buf.device)
                if not t.is_cuda:
                    t = t.to(buf.device)
                t = t.reshape(base.shape) if t.numel() == base.numel() else t
                if tuple(t.shape) != tuple(base.shape):
                    raise ValueError(f"view {v}: {key} has shape {tuple(t.shape)}, the scene's leaf {tuple(base.shape)}")
                self.keep.append(t)
                if key in _OVERRIDE_FIELDS:
                    which, field = _OVERRIDE_FIELDS[key]
                    setattr((self.lights if which == "lights" else self.materials)[v], field, t.data_ptr())
                    self.mask |= _lib.VIEWS_LIGHTS if which == "lights" else _lib.VIEWS_MATERIALS
                else:
                    kind, field = key.split(".")
                    setattr(self.objects[v].seg[buf.kinds.index(kind)], field, t.data_ptr())
                    self.mask |= _lib.VIEWS_OBJECTS

    def view(self, buf: SceneBuffers, v: int) -> SceneBuffers:
        """The scene of view v as resident buffers of its own (for the per-view passes: shadows)."""
        import copy
        one = copy.copy(buf)
        one.objects, one.lights, one.materials = self.objects[v], self.lights[v], self.materials[v]
        return one


def render_views_buffers(buf: SceneBuffers, cams: Sequence[_lib.SrhCamera], images: torch.Tensor, depths: torch.Tensor,
                         nearests: Optional[torch.Tensor] = None, rows: Optional[Tuple[int, int]] = None,
                         workspace: Optional[torch.Tensor] = None, image_row_stride: int = 0,
                         depth_row_stride: int = 0, view_row0: Optional[Sequence[int]] = None,
                         scenes: Optional[ViewScenes] = None, **shading_kw) -> torch.Tensor:
    """Low-level form of ``render_views``: resident scene buffers, camera structs, caller-provided stacked outputs
    (view v starts v * rows * row_stride elements after view 0) and an optional row slab; with ``view_row0`` view v
    renders rows [view_row0[v], view_row0[v] + rows[1] - rows[0]) instead; ``scenes`` gives every view its own
    geometry / lights / materials (``ViewScenes``).  One library call, every pipeline kernel launched once for the
    whole batch.  Returns the workspace (pass it back in to reuse it)."""
    lib = _lib.load()
    width, height = frame_size(cams[0])
    r0, r1 = (0, height) if rows is None else (int(rows[0]), int(rows[1]))
    n = len(cams)
    shading = shading_kw.get("shading", "numpy")
    params = _lib.SrhParams(row0=r0, row1=r1, mode=_lib.MODES["auto"],
                            tonemap_gamma=0 if buf.gamma is None else 1,
                            gamma=1.0 if buf.gamma is None else buf.gamma,
                            shading=_lib.SHADING[shading], double_sided=int(bool(shading_kw.get("double_sided", False))),
                            use_quartic=int(bool(shading_kw.get("use_quartic", False))),
                            waves_per_tile=int(shading_kw.get("waves_per_tile", 0)),
                            image_row_stride=int(image_row_stride), depth_row_stride=int(depth_row_stride))
    row0_arr = None
    if view_row0 is not None:
        if len(view_row0) != n:
            raise ValueError("view_row0 needs one entry per view")
        row0_arr = (C.c_int32 * n)(*[int(r) for r in view_row0])
        params.view_row0 = C.cast(row0_arr, C.c_void_p)
    nbytes = lib.srh_workspace_bytes_views(C.byref(buf.objects), width, height, n)
    if nbytes == 0:
        raise _lib.SrhError(-2, lib.srh_last_error().decode())
    if workspace is None or workspace.numel() < nbytes:
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=buf.device)
    arr = (_lib.SrhCamera * n)(*cams)
    binned = not cams[0].ortho
    key = _layout_key(buf, width, height, ("views", n))
    if binned:
        params.counters_clean = int(_ws_state(workspace) == ("clean", key))
        _ws_note(workspace, None)
    ob, ls, ms = C.byref(buf.objects), C.byref(buf.lights), C.byref(buf.materials)
    if scenes is not None:
        if scenes.n != n:
            raise ValueError(f"{scenes.n} per-view scenes for {n} cameras")
        params.per_view = scenes.mask
        if scenes.mask & _lib.VIEWS_OBJECTS:
            ob = scenes.objects
        if scenes.mask & _lib.VIEWS_LIGHTS:
            ls = scenes.lights
        if scenes.mask & _lib.VIEWS_MATERIALS:
            ms = scenes.materials
    with torch.cuda.device(buf.device):
        _lib.check(lib.srh_render_views(n, arr, ob, ls, ms,
                                        C.byref(params), workspace.data_ptr(), workspace.numel(), images.data_ptr(),
                                        depths.data_ptr(), nearests.data_ptr() if nearests is not None else None,
                                        _stream_ptr(buf.device)))
    if binned:
        _ws_note(workspace, ("clean", key))
    return workspace


def render_views(scene: Dict[str, Any], cameras: Sequence[Dict[str, Any]], device="cuda", mode: str = "auto",
                 streams: int = 4, want_nearest: bool = True, batch: int = 256,
                 overrides: Optional[Sequence[Dict[str, Any]]] = None, **shading_kw) -> Dict[str, torch.Tensor]:
    """Many views per call: the batch axis of the reference's real callers (one ``render()`` per view in a
    Python loop, diffrend/torch/GAN/gan.py:325-378, torch/batch_render.py:36-53).  The scene is uploaded once;
    ``overrides[v]`` replaces leaves of it for view v (``{"disk.pos": ..., "disk.normal": ..., "lights.pos": ...}``: what
    the GAN's loop assigns per batch element -- see ``ViewScenes``), so a batch may hold a different splat set and light
    per view.  In the default binned mode the views go to the library ``batch`` at a time (``srh_render_views``): every
    kernel of the frame pipeline is launched once per batch with the view as a grid dimension, so small views neither
    pay three launches each nor leave the GPU idle.  Other modes, or ``batch=0``, issue one call per view round-robin
    over ``streams`` HIP streams.  All cameras must share one viewport size.  ``shadow=True`` (with ``shading='torch'``)
    runs the shadow-ray pass on every view after its batch (torch/batch_render.py:59,104-106 renders that way by
    default); ``visibility`` (B,H,W) int64 is then returned too.  Returns stacked tensors ``image`` (B,H,W,3), ``depth``
    (B,H,W) and ``nearest`` (B,H,W) int32; ``shading`` / ``double_sided`` / ``use_quartic`` as in ``render``.
    Forward only."""
    unknown = set(shading_kw) - {"shading", "double_sided", "use_quartic", "waves_per_tile", "shadow"}
    if unknown:
        raise TypeError(f"render_views() got unexpected keyword arguments {sorted(unknown)}")
    shadow = bool(shading_kw.pop("shadow", False))
    if shadow and shading_kw.get("shading", "numpy") != "torch":
        raise ValueError("shadow rays exist only in the torch backend's semantics: shading='torch'")
    device = torch.device(device)
    buf = flatten_scene(scene, device)
    cams = [camera_struct(c, shading_kw.get("shading", "numpy")) for c in cameras]
    if not cams:
        raise ValueError("no cameras")
    width, height = frame_size(cams[0])
    if any(frame_size(c) != (width, height) for c in cams):
        raise ValueError("all cameras of a batch must have the same viewport size")
    n = len(cams)
    if overrides is not None and len(overrides) != n:
        raise ValueError(f"{len(overrides)} overrides for {n} cameras")
    want_nearest = want_nearest or shadow                   # the shadow pass starts from the winners
    image = torch.empty((n, height, width, 3), dtype=torch.float32, device=device)
    depth = torch.empty((n, height, width), dtype=torch.float32, device=device)
    nearest = torch.empty((n, height, width), dtype=torch.int32, device=device) if want_nearest else None
    out = {"image": image, "depth": depth}
    if want_nearest:
        out["nearest"] = nearest
    every = ViewScenes(buf, overrides) if overrides is not None else None

    def scene_of(v: int) -> SceneBuffers:
        return every.view(buf, v) if every is not None else buf

    def shadows(first: int, count: int) -> None:
        if not shadow:
            return
        if "visibility" not in out:
            out["visibility"] = torch.empty((n, height, width), dtype=torch.int64, device=device)
        buf.ensure_shadow_workspace(width, height)          # the per-view copies below share it
        for v in range(first, first + count):
            out["visibility"][v] = shadow_pass(scene_of(v), cams[v], None, image[v], depth[v], nearest[v],
                                               double_sided=bool(shading_kw.get("double_sided", False)),
                                               use_quartic=bool(shading_kw.get("use_quartic", False)))

    if every is not None and not (mode in ("auto", "binned") and int(batch) > 0):
        # per-view scenes outside the batched call: one frame per view from that view's buffers
        for v in range(n):
            render_buffers(scene_of(v), cams[v], mode=mode, out=(image[v], depth[v], nearest[v] if want_nearest else None),
                           **shading_kw)
        shadows(0, n)
        return out
    if mode in ("auto", "binned") and int(batch) > 0:
        lib = _lib.load()
        shading = shading_kw.get("shading", "numpy")
        params = _lib.SrhParams(row0=0, row1=height, mode=_lib.MODES[mode],
                                tonemap_gamma=0 if buf.gamma is None else 1,
                                gamma=1.0 if buf.gamma is None else buf.gamma,
                                shading=_lib.SHADING[shading], double_sided=int(bool(shading_kw.get("double_sided", False))),
                                use_quartic=int(bool(shading_kw.get("use_quartic", False))),
                                waves_per_tile=int(shading_kw.get("waves_per_tile", 0)))
        step = max(1, min(int(batch), n))
        nbytes = lib.srh_workspace_bytes_views(C.byref(buf.objects), width, height, step)
        if nbytes == 0:
            raise _lib.SrhError(-2, lib.srh_last_error().decode())
        workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            for i in range(0, n, step):
                m = min(step, n - i)
                arr = (_lib.SrhCamera * m)(*cams[i:i + m])
                ob, ls, ms = C.byref(buf.objects), C.byref(buf.lights), C.byref(buf.materials)
                params.per_view = 0
                if every is not None:                       # element i of each per-view array is this batch's view 0
                    params.per_view = every.mask
                    if every.mask & _lib.VIEWS_OBJECTS:
                        ob = C.byref(every.objects[i])
                    if every.mask & _lib.VIEWS_LIGHTS:
                        ls = C.byref(every.lights[i])
                    if every.mask & _lib.VIEWS_MATERIALS:
                        ms = C.byref(every.materials[i])
                _lib.check(lib.srh_render_views(m, arr, ob, ls, ms,
                                                C.byref(params), workspace.data_ptr(), workspace.numel(),
                                                image[i].data_ptr(), depth[i].data_ptr(),
                                                nearest[i].data_ptr() if want_nearest else None, _stream_ptr(device)))
                shadows(i, m)
        return out
    n_streams = max(1, min(int(streams), n))
    pool = [torch.cuda.Stream(device) for _ in range(n_streams)]
    scratch = [buf.new_workspace(width, height) for _ in range(n_streams)]
    current = torch.cuda.current_stream(device)
    for st in pool:
        st.wait_stream(current)                 # uploads and allocations above happen-before the views
    for i, cam in enumerate(cams):
        k = i % n_streams
        with torch.cuda.stream(pool[k]):
            render_buffers(buf, cam, mode=mode, out=(image[i], depth[i], nearest[i] if want_nearest else None),
                           workspace=scratch[k], **shading_kw)
    for st in pool:
        current.wait_stream(st)
    for t in (image, depth, nearest, *scratch, *buf.tensors.values()):
        if t is not None:
            t.record_stream(current)
    shadows(0, n)
    return out


def _source_leaves(scene: Dict[str, Any]) -> Dict[str, Any]:
    """The caller's own leaf objects under the keys flatten_scene files them under."""
    out: Dict[str, Any] = {}
    for kind, grp in scene["objects"].items():
        for name in _OBJ_FIELDS.get(kind, ()):
            if name in grp:
                out[f"{kind}.{name}"] = grp[name]
    out["lights.pos"] = scene["lights"]["pos"]
    out["colors"] = scene["colors"]
    out["materials.albedo"] = scene["materials"]["albedo"]
    for key, (grp, name) in {"lights.attenuation": ("lights", "attenuation"), "lights.ambient": ("lights", "ambient"),
                             "materials.coeffs": ("materials", "coeffs")}.items():
        if name in scene[grp]:
            out[key] = scene[grp][name]
    return out


class ResidentScene:
    """A scene flattened ONCE and rendered many times -- the shape of an optimisation loop
    (diffrend/torch/test_optimization.py: render, loss, backward, optimiser step, repeat).  ``render(scene)`` pays the
    scene conversion, validation and descriptor building on every call (about as long as the GPU work for a mesh of a
    few thousand triangles); here that happens in the constructor.  Leaves that are contiguous float32 tensors on the
    render device are used IN PLACE: an optimiser that updates them in place (torch.optim does) is seen by the next
    ``render()`` without any copy, and their ``.grad`` is filled by the analytic HIP backward.  ``nearest`` comes back
    as the kernel writes it (int32; ``render(scene)`` widens it to the reference's int64 with one more pass).

        rs = ResidentScene(scene, shading='torch')        # leaves with requires_grad=True stay attached
        for _ in range(steps):
            opt.zero_grad(); loss(rs.render()['image']).backward(); opt.step()
    """

    def __init__(self, scene: Dict[str, Any], device="cuda", shading: str = "numpy", mode: str = "auto",
                 double_sided: bool = False, use_quartic: bool = False, validate: bool = True):
        if shading not in _lib.SHADING:
            raise ValueError(f"shading must be 'numpy' or 'torch', got {shading!r}")
        self.device = torch.device(device)
        self.buf = flatten_scene(scene, self.device, validate=validate, keep_graph=True)
        self.shading, self.mode = shading, mode
        self._camera = scene["camera"]
        self.cam = camera_struct(scene["camera"], shading)
        if self.cam.ortho and shading != "torch":
            raise ValueError("orthographic projection exists only in the torch backend's semantics: shading='torch'")
        self.shade = (shading, bool(double_sided), bool(use_quartic), False)
        self.inputs = [self.buf.tensors[k] for k in _float_keys(self.buf, shading)]
        self.differentiable = any(t.requires_grad for t in self.inputs)
        # "In place" has to be true for every leaf that is being optimised: a float64, CPU or non-contiguous leaf is
        # COPIED once by flatten_scene (the copy stays attached to autograd, so its gradients still reach the leaf and
        # the optimiser keeps stepping it) -- and every later render() would draw the first iteration's values.
        stale = []
        self.leaves: Dict[str, torch.Tensor] = {}           # the caller's own differentiable leaves, by flat key
        for key, leaf in _source_leaves(scene).items():
            if isinstance(leaf, torch.Tensor) and leaf.requires_grad and key in self.buf.tensors:
                self.leaves[key] = leaf
                t = self.buf.tensors[key]
                if t.data_ptr() != leaf.data_ptr() or t.dtype != leaf.dtype or t.device != leaf.device:
                    stale.append(f"{key} ({str(leaf.dtype).replace('torch.', '')} on {leaf.device}"
                                 f"{'' if leaf.is_contiguous() else ', not contiguous'})")
        if stale:
            raise ValueError("ResidentScene uses differentiable leaves in place, and these would be copied once and "
                             "then never refreshed: " + ", ".join(stale) + f".  Give contiguous float32 tensors on "
                             f"{self.device}, or call render(scene) per iteration (it converts on every call).")

    def set_camera(self, camera: Dict[str, Any]) -> None:
        self._camera = camera
        self.cam = camera_struct(camera, self.shading)

    def render(self, rows: Optional[Tuple[int, int]] = None) -> "RenderResult":
        if self.differentiable and torch.is_grad_enabled():
            image, depth, nearest = _RenderFunction.apply(self.buf, self.cam, rows, self.mode, self.shade, *self.inputs)
        else:
            image, depth, nearest = render_buffers(self.buf, self.cam, rows=rows, mode=self.mode, shading=self.shading,
                                                   double_sided=self.shade[1], use_quartic=self.shade[2])
        return RenderResult(self._camera, self.device, image=image, depth=depth, nearest=nearest)

    def capture_step(self, loss_fn, warmup: int = 3) -> "CapturedStep":
        """One optimisation step's GPU work -- render, ``loss_fn(result)``, backward -- captured as ONE hipGraph and
        replayed per iteration: the Python side of an iteration (autograd bookkeeping, descriptor structs, ~20 kernel
        launches) then costs one graph launch.  See ``CapturedStep``."""
        return CapturedStep(self, loss_fn, warmup)


class CapturedStep:
    """``step = rs.capture_step(loss_fn)``; then per iteration ``loss = step.replay(); optimiser.step()``.

    The whole-step capture recipe of torch.cuda.graphs: a few eager iterations on a side stream, then one iteration
    recorded into a graph -- the library's forward, ``loss_fn`` on the rendered image and depth, torch's own backward of
    the loss down to image and depth, and the library's backward from there to the leaves, called directly.  (Letting
    the autograd engine run the renderer's autograd.Function inside a capture ends in a segmentation fault in
    hipStreamEndCapture on ROCm 7.2 -- tools/diag_capture.py; each of the pieces used here captures fine.)  The
    leaves' ``.grad`` tensors are static: every replay overwrites them (they do not accumulate), ``loss`` and
    ``result`` are static tensors too.
    What a replay reads is device memory only -- the leaves (an optimiser's in-place step is seen by the next replay)
    and whatever tensors ``loss_fn`` closes over (update a target with ``copy_``) -- while the camera and everything
    else on the host was frozen at capture time."""

    def __init__(self, rs: ResidentScene, loss_fn, warmup: int = 3):
        if not rs.differentiable:
            raise ValueError("capture_step needs at least one leaf that requires grad")
        self.rs = rs
        keys = _float_keys(rs.buf, rs.shading)
        # the CALLER's leaves (rs.inputs are reshaped views of them: autograd reaches the leaves through the views, a
        # hand-made .grad has to be put on the leaves themselves)
        self.keys = [k for k in keys if k in rs.leaves]
        self.leaves = [rs.leaves[k] for k in self.keys]
        current = torch.cuda.current_stream(rs.device)
        side = torch.cuda.Stream(rs.device)
        side.wait_stream(current)
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):        # module load, allocator, the scratch's bin counters
                for t in self.leaves:
                    t.grad = None
                loss_fn(rs.render()).backward()
        current.wait_stream(side)
        torch.cuda.synchronize(rs.device)
        for t in self.leaves:
            t.grad = None
        self.graph = torch.cuda.CUDAGraph()
        need = [k in rs.leaves for k in keys]
        with torch.cuda.graph(self.graph):
            with torch.no_grad():
                image, depth, nearest = render_buffers(rs.buf, rs.cam, mode=rs.mode, shading=rs.shading,
                                                       double_sided=rs.shade[1], use_quartic=rs.shade[2])
            img, dep = image.requires_grad_(), depth.requires_grad_()
            self.result = RenderResult(rs._camera, rs.device, image=img, depth=dep, nearest=nearest)
            self.loss = loss_fn(self.result)
            g_img, g_dep = torch.autograd.grad(self.loss, [img, dep], allow_unused=True)
            got = _render_backward(rs.buf, rs.cam, None, rs.mode, rs.shade, depth, nearest, None, g_img, g_dep, need)
        # the static gradient tensors every replay writes, in the order of self.leaves and in the leaves' own shapes
        self.grads = [(got[k] if k in got else torch.zeros_like(rs.buf.tensors[k])).view(leaf.shape)
                      for k, leaf in zip(self.keys, self.leaves)]
        for t, g in zip(self.leaves, self.grads):
            t.grad = g

    def replay(self) -> torch.Tensor:
        self.graph.replay()
        for t, g in zip(self.leaves, self.grads):          # whatever happened to .grad in between (zero_grad(set_to_none))
            t.grad = g
        return self.loss


def _norm_depth_image(depth: torch.Tensor, far: float) -> torch.Tensor:
    """`norm_depth_image_only` of the torch backend (torch/renderer.py:245-249), on the far + 1 background depth:
    background pixels take the minimum depth, then (d - min) / (max - min) -- the same tensor expression, so it is
    differentiable through `depth` like the reference's.  The reference's own path raises before it gets here
    (oracle/check_ref_kwargs.py), so this output is restated from the source, not pinned by a reference fixture."""
    min_depth = depth.min()
    image = torch.where(depth >= far, min_depth, depth)
    return (image - min_depth) / (depth.max() - min_depth)


class RenderResult(dict):
    """The reference's result dict.  ``image`` (H,W,3), ``depth`` (H,W) and ``nearest`` (H,W) are always
    present; ``ray_dir`` (4,N) is produced on first access.  The three O(M*N) entries of the numpy
    backend (``ray_dist``, ``obj_dist``, ``valid_pixels``) do not exist in a streaming renderer."""

    def __init__(self, camera, device, *args, **kw):
        super().__init__(*args, **kw)
        self._camera, self._device = camera, device

    def numpy(self, *keys: str) -> Dict[str, np.ndarray]:
        """Host ndarrays of the named entries (default image, depth, nearest) -- what the reference's numpy backend
        returns.  The copies go through pinned host memory from torch's caching host allocator, asynchronously and
        with one wait at the end: ~10x the rate of ``.cpu()`` on a 2048 x 2048 frame.  The arrays own their memory."""
        keys = keys or ("image", "depth", "nearest")
        stream = torch.cuda.current_stream(self._device)
        host = {}
        for k in keys:
            t = self[k].detach()
            host[k] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            host[k].copy_(t, non_blocking=True)
        stream.synchronize()
        return {k: h.numpy() for k, h in host.items()}

    def __missing__(self, key):
        if key == "ray_dir":
            val = generate_rays(self._camera, self._device)
            self[key] = val
            return val
        if key in ("ray_dist", "obj_dist", "valid_pixels"):
            raise KeyError(f"{key!r}: the hip backend never materialises (primitives x pixels) arrays")
        raise KeyError(key)


def render(scene: Dict[str, Any], **params) -> RenderResult:
    """Drop-in for the reference backends' ``render(scene)``.

    Returns torch tensors on the render device: ``image`` (H,W,3) float32, ``depth`` (H,W) float32 with
    +inf where nothing is hit, ``nearest`` (H,W) int64 (0 where nothing is hit), matching
    ``diffrend.numpy.renderer.render`` within fp32 rounding of the stored outputs.

    Keyword arguments: ``device`` ('cuda'), ``mode`` ('auto' | 'exact' | 'fast' | 'binned'), ``rows`` ((r0, r1) slab),
    ``waves_per_tile`` (0 | 1 | 4: launch shape of the binned kernel, a tuning knob with identical results),
    ``validate`` (host-side index / w checks), ``shading`` ('numpy' | 'torch').  With ``shading='torch'`` the call
    follows ``diffrend.torch.renderer.render`` instead (Phong shading with lights.attenuation / lights.ambient /
    materials.coeffs, ``double_sided``, ``use_quartic``, orthonormal camera basis, far+1 background, extra outputs
    ``normal`` and ``pos``; ``shadow=True`` adds the all-pairs shadow-ray pass and a ``light_visibility`` bit field;
    ``camera.proj_type = 'ortho'``; ``norm_depth_image_only=True`` returns the normalised depth as ``image``).  The
    torch backend's remaining kwargs are accepted where they change no output of the reference (``tiled``,
    ``tile_size``, ``backface_culling`` -- see ``_TORCH_ONLY_KWARGS``); ``vis_stat=True`` raises as it does there.
    """
    unknown = set(params) - _TORCH_ONLY_KWARGS - {"device", "mode", "rows", "validate", "shading", "double_sided",
                                                  "use_quartic", "waves_per_tile"}
    if unknown:
        raise TypeError(f"render() got unexpected keyword arguments {sorted(unknown)}")
    device = torch.device(params.get("device", "cuda"))
    buf = flatten_scene(scene, device, validate=params.get("validate", True), keep_graph=torch.is_grad_enabled())
    shading = params.get("shading", "numpy")
    cam = camera_struct(scene["camera"], shading)
    rows, mode = params.get("rows"), params.get("mode", "auto")
    if shading not in _lib.SHADING:
        raise ValueError(f"shading must be 'numpy' or 'torch', got {shading!r}")
    shadow = bool(params.get("shadow", False))
    if shadow and shading != "torch":
        raise ValueError("shadow rays exist only in the torch backend's semantics: shading='torch'")
    if params.get("vis_stat", False):
        raise RuntimeError("Removed Support for vis_stat")          # torch/renderer.py:235, same message
    norm_depth = bool(params.get("norm_depth_image_only", False))
    if norm_depth and shading != "torch":
        raise ValueError("norm_depth_image_only exists only in the torch backend's semantics: shading='torch'")
    inputs = [buf.tensors[k] for k in _float_keys(buf, shading)]
    if cam.ortho:
        if shading != "torch":
            raise ValueError("orthographic projection exists only in the torch backend's semantics: shading='torch'")
    if shading == "torch":
        # the torch backend's semantics
        shade = ("torch", bool(params.get("double_sided", False)), bool(params.get("use_quartic", False)), shadow)
        if torch.is_grad_enabled() and any(t.requires_grad for t in inputs):
            # differentiable call (no normal / pos outputs on this path)
            image, depth, nearest = _RenderFunction.apply(buf, cam, rows, mode, shade, *inputs)
            if norm_depth:
                image = _norm_depth_image(depth, cam.far_clip)
            return RenderResult(scene["camera"], device, image=image, depth=depth, nearest=nearest.to(torch.int64))
        width, height = frame_size(cam)
        r0, r1 = (0, height) if rows is None else rows
        normal = torch.empty((r1 - r0, width, 3), dtype=torch.float32, device=device)
        pos = torch.empty((r1 - r0, width, 3), dtype=torch.float32, device=device)
        image, depth, nearest = render_buffers(buf, cam, rows=rows, mode=mode, shading="torch",
                                               double_sided=params.get("double_sided", False),
                                               use_quartic=params.get("use_quartic", False), aux=(normal, pos),
                                               waves_per_tile=params.get("waves_per_tile", 0))
        if norm_depth:          # torch/renderer.py:245-260 returns before the fragment stage: no normal / pos, no shadows
            return RenderResult(scene["camera"], device, image=_norm_depth_image(depth, cam.far_clip), depth=depth,
                                nearest=nearest.to(torch.int64))
        extra = {}
        if shadow:
            extra["light_visibility"] = shadow_pass(buf, cam, rows, image, depth, nearest, shade[1], shade[2])
        return RenderResult(scene["camera"], device, image=image, depth=depth, nearest=nearest.to(torch.int64),
                            normal=normal, pos=pos, **extra)
    if torch.is_grad_enabled() and any(t.requires_grad for t in inputs):
        # differentiable call: image and depth carry a grad_fn backed by the analytic HIP backward
        image, depth, nearest = _RenderFunction.apply(buf, cam, rows, mode, ("numpy", False, False, False), *inputs)
    else:
        image, depth, nearest = render_buffers(buf, cam, rows=rows, mode=mode,
                                               waves_per_tile=params.get("waves_per_tile", 0))
    return RenderResult(scene["camera"], device, image=image, depth=depth, nearest=nearest.to(torch.int64))
