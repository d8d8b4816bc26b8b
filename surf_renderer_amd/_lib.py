"""ctypes binding of libsrh.so (the C ABI in include/srh.h).  Fails loudly when the library is
missing or was built for a different ABI: there is no CPU fallback behind the hip backend."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

ABI_VERSION = 10
MAX_SEGMENTS = 4
MAX_LIGHTS = 64

MODE_AUTO, MODE_EXACT, MODE_FAST, MODE_BINNED = 0, 1, 2, 3
MODES = {"auto": MODE_AUTO, "exact": MODE_EXACT, "fast": MODE_FAST, "binned": MODE_BINNED}
SHADING = {"numpy": 0, "torch": 1}

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class SrhCamera(C.Structure):
    _fields_ = [("eye", C.c_double * 4), ("at", C.c_double * 4), ("up", C.c_double * 4),
                ("fovy", C.c_double), ("focal_length", C.c_double),
                ("near_clip", C.c_double), ("far_clip", C.c_double),
                ("viewport", C.c_int32 * 4), ("ortho", C.c_int32), ("up_is_unit", C.c_int32)]


class SrhSegment(C.Structure):
    _fields_ = [("type", C.c_int32), ("count", C.c_int32),
                ("pos", C.c_void_p), ("normal", C.c_void_p), ("radius", C.c_void_p),
                ("face", C.c_void_p), ("material_idx", C.c_void_p)]


class SrhObjects(C.Structure):
    _fields_ = [("n_segments", C.c_int32), ("seg", SrhSegment * MAX_SEGMENTS)]


class SrhLights(C.Structure):
    _fields_ = [("n_lights", C.c_int32), ("n_colors", C.c_int32),
                ("pos", C.c_void_p), ("color_idx", C.c_void_p), ("colors", C.c_void_p),
                ("attenuation", C.c_void_p), ("ambient", C.c_void_p)]


class SrhMaterials(C.Structure):
    _fields_ = [("n_materials", C.c_int32), ("albedo", C.c_void_p), ("coeffs", C.c_void_p)]


class SrhParams(C.Structure):
    _fields_ = [("row0", C.c_int32), ("row1", C.c_int32), ("mode", C.c_int32),
                ("tonemap_gamma", C.c_int32), ("gamma", C.c_double),
                ("shading", C.c_int32), ("double_sided", C.c_int32), ("use_quartic", C.c_int32), ("waves_per_tile", C.c_int32),
                ("normal_out", C.c_void_p), ("pos_out", C.c_void_p),
                ("image_row_stride", C.c_int64), ("depth_row_stride", C.c_int64),
                ("nearest_row_stride", C.c_int64),
                ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p), ("visibility", C.c_void_p),
                ("view_row0", C.c_void_p), ("stages", C.c_int32), ("counters_clean", C.c_int32),
                ("per_view", C.c_int32), ("reserved1", C.c_int32)]

STAGE_BIN, STAGE_RENDER, STAGE_KEEP_BINS = 1, 2, 4
VIEWS_OBJECTS, VIEWS_LIGHTS, VIEWS_MATERIALS = 1, 2, 4


class SrhGrads(C.Structure):
    _fields_ = [("pos", C.c_void_p * MAX_SEGMENTS), ("normal", C.c_void_p * MAX_SEGMENTS),
                ("radius", C.c_void_p * MAX_SEGMENTS), ("face", C.c_void_p * MAX_SEGMENTS),
                ("lights_pos", C.c_void_p), ("colors", C.c_void_p), ("albedo", C.c_void_p),
                ("coeffs", C.c_void_p), ("attenuation", C.c_void_p), ("ambient", C.c_void_p)]


EXPORTS = ("srh_abi_version", "srh_last_error", "srh_workspace_bytes", "srh_generate_rays", "srh_render_fwd",
           "srh_render_bwd", "srh_workspace_bytes_views", "srh_render_views", "srh_shadow_shade",
           "srh_shadow_workspace_bytes", "srh_bin_counters",
           "srh_event_create", "srh_event_destroy", "srh_event_elapsed_ms")

_lib: Optional[C.CDLL] = None


class SrhError(RuntimeError):
    """A libsrh entry point returned non-zero."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libsrh error {code}: {message}")
        self.code = code


def lib_path() -> str:
    """In-tree libsrh.so; SRH_LIB points experiments (kernel A/B builds) at another build of the same ABI."""
    return os.environ.get("SRH_LIB") or _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """dlopen libsrh.so (building it with hipcc first if it is absent and a toolchain exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError(f"{path} is missing; run `python -m surf_renderer_amd.build`")
        _build.build_lib()
    # torch first: the device buffers this library is handed come from torch's HIP runtime, and the process must hold
    # ONE runtime -- dlopening libsrh.so before torch pulls in the system's libamdhip64, torch then brings its own, and
    # every launch ends in "no ROCm-capable device is detected" (seen with build() followed by smoke() in one process)
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RuntimeError(f"{path} does not export {name}")
    lib.srh_abi_version.restype = C.c_int
    lib.srh_last_error.restype = C.c_char_p
    lib.srh_workspace_bytes.restype = C.c_size_t
    lib.srh_workspace_bytes.argtypes = [C.POINTER(SrhObjects), C.c_int32, C.c_int32]
    lib.srh_generate_rays.restype = C.c_int
    lib.srh_generate_rays.argtypes = [C.POINTER(SrhCamera), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.srh_render_fwd.restype = C.c_int
    lib.srh_render_fwd.argtypes = [C.POINTER(SrhCamera), C.POINTER(SrhObjects), C.POINTER(SrhLights),
                                   C.POINTER(SrhMaterials), C.POINTER(SrhParams), C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.srh_render_bwd.restype = C.c_int
    lib.srh_render_bwd.argtypes = [C.POINTER(SrhCamera), C.POINTER(SrhObjects), C.POINTER(SrhLights),
                                   C.POINTER(SrhMaterials), C.POINTER(SrhParams), C.c_void_p, C.c_size_t,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SrhGrads), C.c_void_p]
    lib.srh_workspace_bytes_views.restype = C.c_size_t
    lib.srh_workspace_bytes_views.argtypes = [C.POINTER(SrhObjects), C.c_int32, C.c_int32, C.c_int32]
    lib.srh_render_views.restype = C.c_int
    lib.srh_render_views.argtypes = [C.c_int32, C.POINTER(SrhCamera), C.POINTER(SrhObjects), C.POINTER(SrhLights),
                                     C.POINTER(SrhMaterials), C.POINTER(SrhParams), C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.srh_shadow_workspace_bytes.restype = C.c_size_t
    lib.srh_shadow_workspace_bytes.argtypes = [C.POINTER(SrhObjects), C.c_int32, C.c_int32, C.c_int32]
    lib.srh_shadow_shade.restype = C.c_int
    lib.srh_shadow_shade.argtypes = [C.POINTER(SrhCamera), C.POINTER(SrhObjects), C.POINTER(SrhLights),
                                     C.POINTER(SrhMaterials), C.POINTER(SrhParams), C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.srh_bin_counters.restype = C.c_int
    lib.srh_bin_counters.argtypes = [C.POINTER(SrhObjects), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.srh_event_create.restype = C.c_int
    lib.srh_event_create.argtypes = [C.POINTER(C.c_void_p)]
    lib.srh_event_destroy.restype = C.c_int
    lib.srh_event_destroy.argtypes = [C.c_void_p]
    lib.srh_event_elapsed_ms.restype = C.c_int
    lib.srh_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    got = lib.srh_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {got}, this package expects {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != 0:
        raise SrhError(code, load().srh_last_error().decode("utf-8", "replace"))


class EventPair:
    """Two timing-enabled HIP events recorded by libsrh around a frame's dominant kernel."""

    def __init__(self):
        lib = load()
        self.start, self.stop = C.c_void_p(), C.c_void_p()
        check(lib.srh_event_create(C.byref(self.start)))
        check(lib.srh_event_create(C.byref(self.stop)))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        check(load().srh_event_elapsed_ms(self.start, self.stop, C.byref(ms)))
        return float(ms.value)

    def close(self) -> None:
        lib = load()
        for ev in (self.start, self.stop):
            if ev:
                lib.srh_event_destroy(ev)
        self.start = self.stop = C.c_void_p()
