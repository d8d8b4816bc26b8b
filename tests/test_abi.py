"""The C-ABI shared library: it builds for gfx950, loads, and exports every symbol include/srh.h declares.
No compute is launched here (there is no GPU in the build container); argument validation that happens
before any HIP call is exercised."""
import ctypes as C
import os
import re

import pytest

from surf_renderer_amd import _lib, build

HEADER = os.path.join(build.INCLUDE, "srh.h")


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return _lib.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("srh_abi_version", "srh_last_error", "srh_workspace_bytes", "srh_generate_rays", "srh_render_fwd"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"libsrh.so does not export {name}"
    assert set(_lib.EXPORTS) <= set(declared_symbols())


def test_abi_version_matches_header(lib):
    text = open(HEADER).read()
    want = int(re.search(r"#define\s+SRH_ABI_VERSION\s+(\d+)", text).group(1))
    assert lib.srh_abi_version() == want == _lib.ABI_VERSION


def test_struct_layouts_match_the_header():
    # natural C layout of the declarations in srh.h (LP64)
    assert C.sizeof(_lib.SrhCamera) == 3 * 32 + 4 * 8 + 16 + 8
    assert C.sizeof(_lib.SrhSegment) == 8 + 5 * 8
    assert C.sizeof(_lib.SrhObjects) == 8 + _lib.MAX_SEGMENTS * C.sizeof(_lib.SrhSegment)
    assert C.sizeof(_lib.SrhLights) == 8 + 5 * 8
    assert C.sizeof(_lib.SrhMaterials) == 24
    assert C.sizeof(_lib.SrhParams) == 16 + 8 + 16 + 2 * 8 + 3 * 8 + 4 * 8 + 8 + 8        # ... stages, counters_clean | per_view, reserved1


def test_workspace_query_validates_its_input(lib):
    ob = _lib.SrhObjects()
    ob.n_segments = 0
    assert lib.srh_workspace_bytes(C.byref(ob), 64, 64) == 0
    assert b"n_segments" in lib.srh_last_error()
    ob.n_segments = 1
    ob.seg[0].type = 7
    ob.seg[0].count = 3
    assert lib.srh_workspace_bytes(C.byref(ob), 64, 64) == 0
    assert b"type" in lib.srh_last_error()
    ob.seg[0].type = 0
    assert lib.srh_workspace_bytes(C.byref(ob), 64, 64) == 0      # NULL arrays
    dummy = (C.c_float * 16)()
    idx = (C.c_int32 * 4)()
    addr = C.addressof(dummy)
    ob.seg[0].pos = ob.seg[0].normal = ob.seg[0].radius = addr
    ob.seg[0].material_idx = C.addressof(idx)
    small = lib.srh_workspace_bytes(C.byref(ob), 64, 64)
    big = lib.srh_workspace_bytes(C.byref(ob), 2048, 2048)
    assert 0 < small < big and small % 256 == 0
    assert lib.srh_workspace_bytes(C.byref(ob), 0, 64) == 0
    # the tile lists use 32-bit offsets with up to 64 entries per primitive: more than 2^26 primitives is refused
    ob.n_segments = 2
    ob.seg[1] = ob.seg[0]
    ob.seg[0].count = ob.seg[1].count = (1 << 25) + 1
    assert lib.srh_workspace_bytes(C.byref(ob), 64, 64) == 0
    assert b"too many primitives" in lib.srh_last_error()
    ob.seg[1].count = (1 << 25) - 2
    assert lib.srh_workspace_bytes(C.byref(ob), 64, 64) > (1 << 34)


def test_render_rejects_bad_arguments_before_touching_the_gpu(lib):
    cam = _lib.SrhCamera()
    rc = lib.srh_render_fwd(C.byref(cam), None, None, None, None, None, 0, None, None, None, None)
    assert rc != 0 and lib.srh_last_error()                          # empty viewport
    cam.viewport[:] = [0, 0, 8, 8]
    cam.eye[:] = [0, 0, 1, 1]
    cam.at[:] = [0, 0, 1, 1]                                         # eye == at
    cam.up[:] = [0, 1, 0, 0]
    rc = lib.srh_generate_rays(C.byref(cam), 0, 8, None, None)
    assert rc == -5 and b"degenerate" in lib.srh_last_error()
    cam.at[:] = [0, 0, 0, 1]
    assert lib.srh_generate_rays(C.byref(cam), 3, 2, None, None) == -2      # empty row range
    assert lib.srh_generate_rays(C.byref(cam), 0, 8, None, None) == -1      # NULL output


def test_hip_frontend_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from surf_renderer_amd import render, synthetic
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        render(synthetic.demo_scene(8, 8))
