"""GPU parity: the hip backend, called through the C ABI, against (a) the reference's own outputs in
tests/golden and (b) the CPU oracle on seeded scenes.

Stated tolerance (fp32 outputs of an fp64 decision path):
  nearest  identical
  depth    |got - want| <= 1.2e-7 * |want|   (one fp32 ulp), +inf in the same places
  image    |got - want| <= 2e-7 + 2e-6 * |want|, NaN in the same places
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_cases
from oracle import np_oracle
from oracle.golden_io import load_case

pytestmark = pytest.mark.gpu

DEPTH_RTOL = 1.2e-7
IMAGE_RTOL, IMAGE_ATOL = 2e-6, 2e-7


def _render(scene, **kw):
    from surf_renderer_amd import render
    res = render(scene, device="cuda:0", **kw)
    torch.cuda.synchronize()
    return {k: res[k].cpu().numpy() for k in ("image", "depth", "nearest")}


def assert_parity(got, want, max_nearest_mismatch=0.0):
    same = got["nearest"] == want["nearest"]
    frac = 1.0 - same.mean()
    assert frac <= max_nearest_mismatch, f"nearest differs on {frac:.4%} of pixels"
    np.testing.assert_array_equal(np.isinf(got["depth"][same]), np.isinf(want["depth"][same]))
    fin = same & np.isfinite(want["depth"])
    np.testing.assert_allclose(got["depth"][fin], want["depth"][fin], rtol=DEPTH_RTOL, atol=0)
    np.testing.assert_allclose(got["image"][same], want["image"][same], rtol=IMAGE_RTOL, atol=IMAGE_ATOL,
                               equal_nan=True)


@pytest.mark.parametrize("mode", ["exact", "fast", "binned"])
@pytest.mark.parametrize("case", golden_cases())
def test_golden(case, mode):
    scene, want, _ = load_case(os.path.join(GOLDEN_DIR, case + ".npz"))
    got = _render(scene, mode=mode)
    assert got["nearest"].dtype == np.int64 and got["image"].dtype == np.float32
    assert_parity(got, want)


def test_row_slab_equals_full_frame():
    scene, want, _ = load_case(os.path.join(GOLDEN_DIR, "g2_demo_planes_64x48.npz"))
    full = _render(scene)
    part = _render(scene, rows=(5, 31))
    for k in ("image", "depth", "nearest"):
        np.testing.assert_array_equal(part[k], full[k][5:31])


def test_ray_dir_matches_oracle():
    from surf_renderer_amd import generate_rays
    scene, _, _ = load_case(os.path.join(GOLDEN_DIR, "g8j_array_camera_reordered.npz"))
    got = generate_rays(scene["camera"], device="cuda:0").cpu().numpy()
    _, want, _, _ = np_oracle.generate_rays(scene["camera"])
    np.testing.assert_allclose(got, want, rtol=0, atol=6e-8)


def test_caller_scene_not_modified():
    scene, _, _ = load_case(os.path.join(GOLDEN_DIR, "g1_demo_64x48.npz"))
    import copy
    before = copy.deepcopy(scene)
    _render(scene)
    for kind in scene["objects"]:
        for k, v in scene["objects"][kind].items():
            np.testing.assert_array_equal(v, before["objects"][kind][k])


def _modes_identical(scene, modes=("exact", "fast", "binned")):
    ref = _render(scene, mode=modes[0])
    for mode in modes[1:]:
        # the binned kernel has two launch shapes (one or four waves per tile); the default picks by tile count
        for wpt in ((0, 1, 4) if mode == "binned" else (0,)):
            got = _render(scene, mode=mode, waves_per_tile=wpt)
            for k in ("nearest", "depth", "image"):
                np.testing.assert_array_equal(got[k], ref[k], err_msg=f"{mode} (waves_per_tile {wpt}) vs {modes[0]}: {k}")
    return ref


@pytest.mark.parametrize("builder", ["disk_cloud", "bunny_splat", "bunny_mesh", "mixed", "halfbox"])
def test_modes_are_bit_identical(builder):
    """The reject tests and the tile binning may only skip pairs that are misses: every mode must produce
    the same bits as the all-pairs fp64 mode."""
    from surf_renderer_amd import synthetic
    scene = {
        "disk_cloud": lambda: synthetic.disk_cloud_scene(20000, 640, 360, radius=0.03, seed=5),
        "bunny_splat": lambda: synthetic.bunny_splat_scene(256, 256),
        "bunny_mesh": lambda: synthetic.bunny_mesh_scene(256, 192),
        "mixed": lambda: synthetic.demo_scene(333, 250, with_planes=True),
        "halfbox": lambda: synthetic.json_scene("halfbox_sphere_cube.json", 320, 240),
    }[builder]()
    ref = _modes_identical(scene)
    assert np.isfinite(ref["depth"]).mean() > 0.05


def test_modes_identical_random_cameras():
    """Discs seen edge-on, behind the camera, straddling the image border, from inside the cloud."""
    from surf_renderer_amd import synthetic
    rng = np.random.RandomState(11)
    for trial in range(6):
        scene = synthetic.disk_cloud_scene(3000, 200, 150, radius=float(rng.uniform(0.01, 0.3)), seed=100 + trial)
        eye = rng.uniform(-1.5, 1.5, size=3)
        scene["camera"]["eye"] = [float(eye[0]), float(eye[1]), float(eye[2]), 1.0]
        scene["camera"]["at"] = [float(v) for v in rng.uniform(-0.3, 0.3, size=3)] + [1.0]
        scene["camera"]["fovy"] = float(np.deg2rad(rng.uniform(20, 110)))
        scene["camera"]["near"] = float(rng.choice([1e-3, 0.1, 0.5]))
        _modes_identical(scene)


def test_oracle_parity_mid_size():
    """CPU oracle vs hip on a seeded scene larger than the goldens (fp64 oracle, a few seconds)."""
    from surf_renderer_amd import synthetic
    from surf_renderer_amd.scene import scene_to_numpy
    scene = synthetic.disk_cloud_scene(1500, 160, 120, radius=0.06, seed=3)
    want = np_oracle.render(scene_to_numpy(scene, round_fp32=True))
    assert_parity(_render(scene), want)


def test_reject_margins_stress():
    """Many seeded disc clouds at a resolution where thousands of pixel centres fall within a hair of an ellipse
    edge: the fp32 reject tests must never drop a pair the fp64 path accepts (binned == exact, bit for bit)."""
    from surf_renderer_amd import synthetic
    rng = np.random.RandomState(2024)
    for trial in range(12):
        n = int(rng.choice([500, 4000, 12000]))
        scene = synthetic.disk_cloud_scene(n, 512, 384, radius=float(rng.choice([0.004, 0.02, 0.07])), seed=300 + trial)
        scene["camera"]["eye"] = [float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(1.5, 6)), 1.0]
        _modes_identical(scene, modes=("exact", "binned"))


# ---- BASELINE-size checks: size-independent properties + the CPU oracle on a few rows ------------------------------
def _oracle_rows_check(scene, got, rows):
    from surf_renderer_amd.scene import scene_to_numpy
    sc = scene_to_numpy(scene, round_fp32=True)
    for r0, r1 in rows:
        want = np_oracle.render(sc, rows=(r0, r1), tile=512)
        part = {k: got[k][r0:r1] for k in ("image", "depth", "nearest")}
        assert_parity(part, want)


def test_config2_bunny_splat_512():
    from surf_renderer_amd import synthetic
    scene = synthetic.bunny_splat_scene(512, 512)
    got = _modes_identical(scene, modes=("binned", "fast"))
    assert 0.2 < np.isfinite(got["depth"]).mean() < 0.4
    _oracle_rows_check(scene, got, [(255, 257)])


def test_config3_mixed_and_halfbox_1024():
    from surf_renderer_amd import synthetic
    for scene in (synthetic.json_scene("halfbox_sphere_cube.json", 1024, 1024),
                  synthetic.demo_scene(1024, 1024, with_planes=True)):
        got = _modes_identical(scene, modes=("binned", "fast"))
        _oracle_rows_check(scene, got, [(300, 302), (700, 701)])


def test_config4_bunny_mesh_1024_forward():
    from surf_renderer_amd import synthetic
    scene = synthetic.bunny_mesh_scene(1024, 1024)
    got = _modes_identical(scene, modes=("binned", "fast"))
    _oracle_rows_check(scene, got, [(512, 513)])


def test_config5_full_size_modes_identical_and_row_slabs():
    """100 000 discs at 2048 x 2048: the binned frame equals the all-pairs fp64 frame bit for bit, row slabs of any
    split reassemble to the same frame, and two renders of the same frame are identical (atomics only reorder
    bin contents, never the result)."""
    from surf_renderer_amd import renderer, synthetic
    scene = synthetic.disk_cloud_scene()
    buf = renderer.flatten_scene(scene, "cuda:0")
    cam = renderer.camera_struct(scene["camera"])

    def frame(mode, rows=None):
        img, dep, near = renderer.render_buffers(buf, cam, rows=rows, mode=mode)
        torch.cuda.synchronize()
        return img.cpu().numpy(), dep.cpu().numpy(), near.cpu().numpy()

    ref = frame("exact")
    a = frame("binned")
    b = frame("binned")
    for x, y, z in zip(ref, a, b):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(y, z)
    assert 0.5 < np.isfinite(ref[1]).mean() < 0.7
    parts = [frame("binned", rows=r) for r in ((0, 683), (683, 1366), (1366, 2048))]
    for k in range(3):
        np.testing.assert_array_equal(np.concatenate([p[k] for p in parts]), ref[k])
    _oracle_rows_check(scene, {"image": a[0], "depth": a[1], "nearest": a[2].astype(np.int64)}, [(1024, 1025)])


def test_long_tile_lists_saturate_the_ordinal():
    """More than 4095 primitives in one tile: the packed keys' ordinal saturates and such pixels must resolve on
    the wave-parallel slow path -- still bit-identical to the all-pairs mode."""
    from surf_renderer_amd import synthetic
    scene = synthetic.disk_cloud_scene(30000, 48, 32, radius=0.05, seed=17)
    ref = _modes_identical(scene, modes=("exact", "binned"))
    assert np.isfinite(ref["depth"]).mean() > 0.3
    # many large primitives (every one lands in the frame-wide list) plus near <= 0 (no depth pre-test at all)
    scene = synthetic.disk_cloud_scene(300, 96, 64, radius=0.9, seed=18)
    _modes_identical(scene, modes=("exact", "binned"))
    scene["camera"]["near"] = 0.0
    _modes_identical(scene, modes=("exact", "binned"))


def test_crowded_bins_overflow_to_the_frame_wide_list():
    """One-pass binning gives every bin a fixed number of slots.  Crowd a few tiles of a frame with many tiles far
    beyond that capacity (here 64 slots, ~1500 discs over a 3x3-tile patch): the discs that find their bin full go to
    the frame-wide list that every tile tests, and the frame stays bit-identical to the all-pairs mode."""
    from surf_renderer_amd import synthetic
    for (w, h) in ((512, 512), (1024, 1024)):            # four waves per tile / one wave per tile
        scene = synthetic.disk_cloud_scene(1500, w, h, radius=0.01, seed=23)
        d = scene["objects"]["disk"]
        rng = np.random.RandomState(5)
        d["pos"][:, :2] = rng.uniform(-0.03, 0.03, size=(1500, 2)).astype(np.float32)      # a patch at the image centre
        d["pos"][:1000, 2] = rng.uniform(-0.5, 0.5, size=1000).astype(np.float32)
        # plus a sprinkling over the whole frame, so ordinary bins and crowded ones meet in one frame
        d["pos"][1000:, :3] = rng.uniform(-1.0, 1.0, size=(500, 3)).astype(np.float32)
        ref = _modes_identical(scene, modes=("exact", "binned"))
        assert np.isfinite(ref["depth"]).any()
        if w == 512:                                      # the same through the batched entry point (per-view bins)
            from surf_renderer_amd import render_views
            cams = [dict(scene["camera"], eye=[0.3 * k, 0.0, 4.0, 1.0]) for k in range(3)]
            batch = render_views(scene, cams, device="cuda:0")
            torch.cuda.synchronize()
            for i, cam in enumerate(cams):
                single = _render({**scene, "camera": cam}, mode="exact")
                for k in ("nearest", "depth", "image"):
                    np.testing.assert_array_equal(batch[k][i].cpu().numpy(), single[k], err_msg=f"crowded view {i}: {k}")


def test_render_views_equals_per_view_render():
    """Batched multi-view rendering (several streams, shared scene) returns exactly what render() returns per view."""
    from surf_renderer_amd import render, render_views, synthetic
    scene = synthetic.bunny_splat_scene(96, 80)
    rng = np.random.RandomState(5)
    cams = []
    for _ in range(7):
        cam = dict(scene["camera"])
        eye = rng.normal(size=3)
        eye = 10.0 * eye / np.linalg.norm(eye)
        cam["eye"] = [float(eye[0]), float(eye[1]), float(eye[2]), 1.0]
        cams.append(cam)
    singles = [render({**scene, "camera": cam}, device="cuda:0") for cam in cams]
    # one library call per batch of views (batches of 3 -> 3 + 3 + 1, and all 7 at once), and the stream-pool path
    for kw in ({"batch": 3}, {"batch": 256}, {"batch": 0, "streams": 3}, {"mode": "fast", "streams": 2}):
        batch = render_views(scene, cams, device="cuda:0", **kw)
        torch.cuda.synchronize()
        assert batch["image"].shape == (7, 80, 96, 3)
        for i, single in enumerate(singles):
            for k in ("image", "depth", "nearest"):
                np.testing.assert_array_equal(batch[k][i].cpu().numpy(), single[k].cpu().numpy(), err_msg=f"{kw} view {i} {k}")
    assert np.isfinite(batch["depth"].cpu().numpy()).mean() > 0.05
    # torch shading, a mixed scene with frame-wide primitives (planes), larger views
    scene2 = synthetic.demo_scene(200, 144, with_planes=True)
    scene2["lights"]["attenuation"] = np.array([[1, 0, 0]] * 4, dtype=np.float32)
    scene2["lights"]["ambient"] = np.array([0.02, 0.02, 0.02], dtype=np.float32)
    scene2["materials"]["coeffs"] = np.array([[0.8, 0.2, 6.0]] * len(scene2["materials"]["albedo"]), dtype=np.float32)
    cams2 = []
    for k in range(5):
        cam = dict(scene2["camera"])
        cam["eye"] = [float(3.0 * np.cos(k)), 1.0 + 0.3 * k, float(10.0 + np.sin(k)), 1.0]
        cams2.append(cam)
    batch = render_views(scene2, cams2, device="cuda:0", shading="torch", double_sided=True)
    for i, cam in enumerate(cams2):
        single = render({**scene2, "camera": cam}, device="cuda:0", shading="torch", double_sided=True)
        for k in ("image", "depth", "nearest"):
            np.testing.assert_array_equal(batch[k][i].cpu().numpy(), single[k].cpu().numpy(), err_msg=f"torch view {i} {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["spheres", "coplanar_discs", "spheres_behind_eye"])
def test_undecided_pixel_paths_are_bit_identical(kind):
    """Scenes whose candidates cannot be ranked by the fp32 bound (overlapping spheres, coplanar overlapping splats, the
    numpy backend's t = 1.0 sentinel for spheres behind the eye): most pixels go through the re-sweep / slow paths,
    and must still equal the all-pairs fp64 mode bit for bit."""
    from surf_renderer_amd import synthetic
    n = 3000
    scene = synthetic.disk_cloud_scene(n, 256, 256, radius=0.08, seed=11)
    rng = np.random.RandomState(3)
    if kind == "coplanar_discs":
        pos = np.concatenate([rng.uniform(-1, 1, (n, 2)), 1e-4 * rng.normal(size=(n, 1)), np.ones((n, 1))], 1)
        scene["objects"]["disk"]["pos"] = pos.astype(np.float32)
        scene["objects"]["disk"]["normal"] = np.tile(np.array([[0, 0, 1, 0]], np.float32), (n, 1))
    else:
        d = scene["objects"].pop("disk")
        pos = d["pos"].copy()
        if kind == "spheres_behind_eye":
            pos[: n // 2, 2] += 6.0                    # eye is at z = 4: half of the spheres sit behind it
        scene["objects"]["sphere"] = {"pos": pos, "radius": d["radius"], "material_idx": d["material_idx"]}
    ref = _modes_identical(scene, modes=("exact", "binned"))
    assert np.isfinite(ref["depth"]).mean() > 0.3


@pytest.mark.gpu
def test_saturated_list_positions_are_bit_identical():
    """More than 4095 primitives in one tile's lists: key fields saturate and such pixels are resolved by the
    re-sweep; the result must not change."""
    from surf_renderer_amd import synthetic
    scene = synthetic.disk_cloud_scene(6000, 64, 64, radius=0.9, seed=2)      # every disc covers most of the 16 tiles
    ref = _modes_identical(scene, modes=("exact", "binned"))
    assert np.isfinite(ref["depth"]).mean() > 0.9


@pytest.mark.gpu
def test_views_batch_of_row_slabs_into_interleaved_buffers():
    """What a multi-GPU rank does per batch: its row slab of several frames, one library call, written into the
    (frames, rows, 4W) send buffer of the collection ([W x rgb | W x depth] per row) -- equal to per-frame calls."""
    from surf_renderer_amd import renderer, synthetic
    scene = synthetic.disk_cloud_scene(4000, 320, 256, radius=0.05, seed=8)
    buf = renderer.flatten_scene(scene, "cuda:0")
    cams = []
    for k in range(3):
        cam = dict(scene["camera"])
        cam["eye"] = [0.3 * k, -0.2 * k, 4.0, 1.0]
        cams.append(renderer.camera_struct(cam))
    W, r0, r1 = 320, 64, 160
    h = r1 - r0
    send = torch.zeros((3, h, 4 * W), dtype=torch.float32, device="cuda:0")
    img = send.as_strided((3, h, W, 3), (h * 4 * W, 4 * W, 3, 1), 0)
    dep = send.as_strided((3, h, W), (h * 4 * W, 4 * W, 1), 3 * W)
    renderer.render_views_buffers(buf, cams, img, dep, rows=(r0, r1), image_row_stride=4 * W, depth_row_stride=4 * W)
    torch.cuda.synchronize()
    for k, cam in enumerate(cams):
        i1, d1, _ = renderer.render_buffers(buf, cam, rows=(r0, r1))
        np.testing.assert_array_equal(img[k].cpu().numpy(), i1.cpu().numpy())
        np.testing.assert_array_equal(dep[k].cpu().numpy(), d1.cpu().numpy())
    assert np.isfinite(dep.cpu().numpy()).mean() > 0.2
    # views with their own first rows (a rank's two half-slabs of each frame of a batch)
    hh = 48
    row0 = [32, 176, 32, 176, 32, 176]
    send2 = torch.zeros((6, hh, 4 * W), dtype=torch.float32, device="cuda:0")
    img2 = send2.as_strided((6, hh, W, 3), (hh * 4 * W, 4 * W, 3, 1), 0)
    dep2 = send2.as_strided((6, hh, W), (hh * 4 * W, 4 * W, 1), 3 * W)
    renderer.render_views_buffers(buf, [c for c in cams for _ in range(2)], img2, dep2, rows=(0, hh), view_row0=row0,
                                  image_row_stride=4 * W, depth_row_stride=4 * W)
    torch.cuda.synchronize()
    for v in range(6):
        i1, d1, _ = renderer.render_buffers(buf, cams[v // 2], rows=(row0[v], row0[v] + hh))
        np.testing.assert_array_equal(img2[v].cpu().numpy(), i1.cpu().numpy())
        np.testing.assert_array_equal(dep2[v].cpu().numpy(), d1.cpu().numpy())


def _random_scene(rng):
    """A random mixed scene for the fuzz below: 1-4 primitive types, 1-3000 primitives each with log-uniform sizes from
    sub-pixel to screen-filling, random camera (also inside the cloud), near in {0, 0.01, 0.1, 1}."""
    f32 = lambda a: np.asarray(a, dtype=np.float32)          # noqa: E731
    W, H = int(rng.choice([48, 64, 97, 128, 200])), int(rng.choice([48, 64, 80, 128, 160]))
    eye = rng.normal(size=3)
    eye = eye / np.linalg.norm(eye) * rng.choice([0.3, 1.0, 2.5, 4.0, 8.0])
    cam = {"viewport": [0, 0, W, H], "fovy": float(np.deg2rad(rng.choice([20, 45, 70, 110]))),
           "focal_length": float(rng.choice([0.5, 1.0, 3.0])), "eye": [*map(float, eye), 1.0],
           "at": [*map(float, rng.normal(size=3) * 0.2), 1.0], "up": [*map(float, rng.normal(size=3)), 0.0],
           "near": float(rng.choice([0.0, 0.01, 0.1, 1.0])), "far": float(rng.choice([5.0, 50.0, 1000.0]))}
    objs = {}
    for k in list(rng.permutation(["disk", "triangle", "sphere", "plane"]))[: rng.randint(1, 5)]:
        n = int(rng.choice([1, 5, 60, 700, 3000])) if k != "plane" else int(rng.choice([1, 2, 3]))
        pos = np.concatenate([rng.uniform(-1.5, 1.5, (n, 3)), np.ones((n, 1))], 1)
        nrm = np.concatenate([rng.normal(size=(n, 3)), np.zeros((n, 1))], 1)
        mat = rng.randint(0, 3, n)
        if k == "disk":
            objs[k] = {"pos": f32(pos), "normal": f32(nrm), "material_idx": mat,
                       "radius": f32(np.exp(rng.uniform(np.log(0.003), np.log(1.5), n)))}
        elif k == "sphere":
            objs[k] = {"pos": f32(pos), "radius": f32(np.exp(rng.uniform(np.log(0.005), np.log(0.8), n))), "material_idx": mat}
        elif k == "plane":
            pos[:, :3] *= 2.0
            objs[k] = {"pos": f32(pos), "normal": f32(nrm), "material_idx": mat}
        else:
            c = rng.uniform(-1.5, 1.5, (n, 1, 3))
            v = c + rng.normal(size=(n, 3, 3)) * np.exp(rng.uniform(np.log(0.01), np.log(0.8), (n, 1, 1)))
            fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]) * rng.choice([-1, 1], (n, 1))
            objs[k] = {"face": f32(np.concatenate([v, np.ones((n, 3, 1))], 2)),
                       "normal": f32(np.concatenate([fn, np.zeros((n, 1))], 1)), "material_idx": mat}
    return {"camera": cam, "lights": {"pos": f32([[3, 4, 5, 1], [-4, 2, 3, 1]]), "color_idx": np.array([1, 2])},
            "colors": f32([[0, 0, 0], [.8, .5, .4], [.3, .6, .9]]),
            "materials": {"albedo": f32([[.5, .5, .5], [.9, .3, .2], [.2, .7, .4]])},
            "objects": objs, "tonemap": {"type": "gamma", "gamma": 0.8}}


@pytest.mark.gpu
def test_fuzz_binned_equals_all_pairs():
    """300 random scenes: the binned mode (both launch shapes) against the all-pairs fp64 mode, bit for bit.  This is the
    test that caught candidates without a depth estimate being dropped after a hit at t = 0 (near <= 0, many spheres)."""
    rng = np.random.RandomState(20240607)
    for it in range(300):
        scene = _random_scene(rng)
        ref = _render(scene, mode="exact")
        for wpt in (1, 4):
            got = _render(scene, mode="binned", waves_per_tile=wpt)
            for k in ("nearest", "depth", "image"):
                ok = np.array_equal(got[k], ref[k], equal_nan=True)
                assert ok, f"scene {it} (waves_per_tile {wpt}): {k} differs on {(got[k] != ref[k]).sum()} values"


@pytest.mark.gpu
def test_sphere_cloud_with_near_zero():
    """near = 0: a sphere whose line is missed gives the valid t = 0 (numpy backend, Q2), so every pixel is a tie at
    t = 0 between hundreds of spheres, and the lowest index must win -- through keys without a depth estimate."""
    from surf_renderer_amd import synthetic
    scene = synthetic.disk_cloud_scene(1500, 128, 96, radius=0.05, seed=12)
    d = scene["objects"].pop("disk")
    scene["objects"]["sphere"] = {"pos": d["pos"], "radius": d["radius"], "material_idx": d["material_idx"]}
    scene["camera"]["near"] = 0.0
    _modes_identical(scene, modes=("exact", "binned"))


@pytest.mark.gpu
def test_fuzz_slabs_torch_shading_and_views():
    """150 random scenes through the other entry points: torch shading exact vs binned, a random row slab against the
    rows of the full frame (the slab path culls primitives by their bounding ball -- this fuzz caught spheres being
    culled although near <= 0 makes a missed sphere a valid hit at t = 0 on every pixel), and a batch of views against
    per-view renders."""
    from surf_renderer_amd import render, render_views
    rng = np.random.RandomState(77)

    def same(a, b, what):
        for k in ("nearest", "depth", "image"):
            x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
            assert np.array_equal(x, y, equal_nan=True), f"{what}: {k} differs on {(x != y).sum()} values"

    for it in range(150):
        sc = _random_scene(rng)
        H = sc["camera"]["viewport"][3]
        sc["lights"]["attenuation"] = np.array([[1, 0, 0], [0.5, 0.1, 0.01]], dtype=np.float32)
        sc["lights"]["ambient"] = np.array([0.01, 0.02, 0.01], dtype=np.float32)
        sc["materials"]["coeffs"] = np.array([[1, 0, 0], [0.7, 0.3, 5], [0.5, 0.5, 20]], dtype=np.float32)
        ds = bool(rng.randint(2))
        same(render(sc, device="cuda:0", mode="exact", shading="torch", double_sided=ds),
             render(sc, device="cuda:0", mode="binned", shading="torch", double_sided=ds,
                    waves_per_tile=int(rng.choice([1, 4]))), f"scene {it} torch shading")
        full = render(sc, device="cuda:0")
        r0 = int(rng.randint(0, H - 1))
        r1 = int(rng.randint(r0 + 1, H + 1))
        same({k: full[k][r0:r1] for k in ("nearest", "depth", "image")}, render(sc, device="cuda:0", rows=(r0, r1)),
             f"scene {it} rows {r0}:{r1}")
        if it % 5 == 0:
            cams = []
            for _ in range(3):
                e = rng.normal(size=3)
                e = e / np.linalg.norm(e) * 3.0
                cams.append(dict(sc["camera"], eye=[*map(float, e), 1.0]))
            vb = render_views(sc, cams, device="cuda:0")
            for i, cam in enumerate(cams):
                one = render({**sc, "camera": cam}, device="cuda:0")
                same({"image": vb["image"][i], "depth": vb["depth"][i], "nearest": vb["nearest"][i].to(torch.int64)},
                     one, f"scene {it} view {i}")


@pytest.mark.gpu
def test_fuzz_against_the_oracles():
    """60 small random scenes (list-typed, non-unit, non-fp32 camera vectors; all primitive types; near down to 0)
    against the CPU oracles at the stated tolerance: numpy semantics, then Phong / attenuation / ambient semantics
    in both projections.  This is the test that caught the float32 normalisation of a list-typed `up` (Q11)."""
    from surf_renderer_amd import render
    from surf_renderer_amd.scene import scene_to_numpy
    from oracle import np_oracle_tch
    rng = np.random.RandomState(77)
    done = 0
    while done < 60:
        scene = _random_scene(rng)
        W, H = scene["camera"]["viewport"][2:]
        if W * H > 64 * 80 or sum(len(g["material_idx"]) for g in scene["objects"].values()) > 800:
            continue
        done += 1
        assert_parity(_render(scene), np_oracle.render(scene_to_numpy(scene, round_fp32=True)))
        scene["lights"]["attenuation"] = np.array([[1, 0, 0], [0.5, 0.1, 0.01]], dtype=np.float32)
        scene["lights"]["ambient"] = np.array([0.01, 0.02, 0.01], dtype=np.float32)
        scene["materials"]["coeffs"] = np.array([[1, 0, 0], [0.7, 0.3, 5], [0.5, 0.5, 20]], dtype=np.float32)
        scene["camera"]["near"] = max(scene["camera"]["near"], 0.01)
        for proj in ("perspective", "ortho"):
            scene["camera"]["proj_type"] = proj
            ds = bool(rng.randint(2))
            sc = scene_to_numpy(scene, round_fp32=True)
            sc["camera"]["proj_type"] = proj
            want = np_oracle_tch.render(sc, double_sided=ds)
            res = render(scene, device="cuda:0", shading="torch", double_sided=ds)
            same = res["nearest"].cpu().numpy() == want["nearest"]
            assert same.all(), f"scene {done} ({proj}): nearest differs on {(~same).sum()} pixels"
            np.testing.assert_allclose(res["image"].cpu().numpy(), want["image"], rtol=IMAGE_RTOL, atol=IMAGE_ATOL)


@pytest.mark.gpu
def test_host_scene_upload_and_pinned_readback():
    """render(scene) with host leaves of mixed containers (float64 / float32 ndarrays, CPU tensors, lists) sends them
    in one packed transfer: same frame as with leaves already on the device; and RenderResult.numpy() returns what
    .cpu() does.  Two frames back to back share the staging buffer without the second overwriting the first."""
    from surf_renderer_amd import render, synthetic
    a = synthetic.demo_scene(96, 64, with_planes=True)
    b = synthetic.disk_cloud_scene(500, 96, 64, radius=0.1, seed=2)
    mixed = copy_scene = __import__("copy").deepcopy(a)
    mixed["objects"]["disk"]["pos"] = torch.tensor(np.asarray(a["objects"]["disk"]["pos"], dtype=np.float64))
    mixed["objects"]["sphere"]["radius"] = [float(r) for r in np.ravel(a["objects"]["sphere"]["radius"])]
    mixed["colors"] = np.asarray(a["colors"], dtype=np.float32)
    on_device = __import__("copy").deepcopy(a)
    for grp in on_device["objects"].values():
        for k in list(grp):
            grp[k] = torch.tensor(np.asarray(grp[k]), device="cuda:0")
    ra, rb = render(mixed, device="cuda:0"), render(b, device="cuda:0")      # b's upload must not disturb a's frame
    rd = render(on_device, device="cuda:0")
    for k in ("image", "depth", "nearest"):
        assert torch.equal(ra[k], rd[k]), k
    host = ra.numpy()
    for k in ("image", "depth", "nearest"):
        np.testing.assert_array_equal(host[k], ra[k].cpu().numpy())
    assert_parity({k: v for k, v in rb.numpy().items()}, _render(b))
    assert set(ra.numpy("depth")) == {"depth"}


@pytest.mark.gpu
def test_scale_millions_of_primitives_and_an_8k_frame():
    """Sizes well beyond BASELINE's: 4 M sub-pixel discs (tile lists of ~4 k entries everywhere) and an 8192 x 8192
    frame (262 144 tiles): the binned pipeline against the all-pairs mode, bit for bit."""
    from surf_renderer_amd import render, synthetic
    for scene in (synthetic.disk_cloud_scene(4_000_000, 512, 512, radius=0.003, seed=1),
                  synthetic.disk_cloud_scene(20_000, 8192, 8192, radius=0.03, seed=2)):
        a = render(scene, device="cuda:0", mode="binned")
        b = render(scene, device="cuda:0", mode="fast")
        assert 0.2 < torch.isfinite(a["depth"]).float().mean().item() < 0.99
        for k in ("image", "depth", "nearest"):
            assert torch.equal(a[k], b[k]), k
        del a, b


_SPECIAL = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-30, -1e-30, 1e20, -1e20, 1e38, 3e-39, 1.0, -1.0], dtype=np.float32)


def _poison(rng, scene, frac):
    """Overwrite one coordinate of a fraction of the primitives (never w) with nan / inf / huge / denormal / zero."""
    def hit(a, cols):
        a = np.array(a, dtype=np.float32, copy=True)
        flat = a.reshape(a.shape[0], -1)
        for r in np.nonzero(rng.uniform(size=flat.shape[0]) < frac)[0]:
            flat[r, rng.choice(cols)] = rng.choice(_SPECIAL)
        return flat.reshape(a.shape)
    for grp in scene["objects"].values():
        if "pos" in grp:
            grp["pos"] = hit(grp["pos"], [0, 1, 2])
        if "normal" in grp:
            grp["normal"] = hit(grp["normal"], [0, 1, 2])
        if "radius" in grp:
            grp["radius"] = hit(grp["radius"].reshape(-1, 1), [0]).reshape(-1)
        if "face" in grp:
            grp["face"] = hit(grp["face"], [0, 1, 2, 4, 5, 6, 8, 9, 10])


@pytest.mark.gpu
def test_fuzz_non_finite_and_degenerate_primitives():
    """Random scenes in which 2-50 % of the primitives carry a nan / inf / 1e38 / denormal / zero coordinate, radius or
    normal component: every mode gives the all-pairs fp64 result bit for bit (nan-aware), and small ones match the
    numpy oracle -- whose comparisons drop a nan hit exactly as `near <= t` does in the reference (numpy/renderer.py:219)."""
    from surf_renderer_amd.scene import scene_to_numpy
    # SRH_FUZZ_NONFINITE_SEED / _SCENES: one-off campaigns with other seeds (profiles/r02_fuzz_campaign.txt)
    rng = np.random.RandomState(int(os.environ.get("SRH_FUZZ_NONFINITE_SEED", "71")))
    for it in range(int(os.environ.get("SRH_FUZZ_NONFINITE_SCENES", "60"))):
        scene = _random_scene(rng)
        _poison(rng, scene, float(rng.choice([0.02, 0.2, 0.5])))
        ref = _render(scene, mode="exact")
        for mode, wpt in (("fast", 0), ("binned", 1), ("binned", 4)):
            got = _render(scene, mode=mode, waves_per_tile=wpt)
            for k in ("nearest", "depth", "image"):
                assert np.array_equal(got[k], ref[k], equal_nan=True), f"scene {it} {mode}/{wpt}: {k} differs"
        W, H = scene["camera"]["viewport"][2:]
        if W * H <= 64 * 80 and sum(len(g["material_idx"]) for g in scene["objects"].values()) <= 800:
            with np.errstate(all="ignore"):
                # dots="ordered": the reference's np.dot contractions as explicit sums in the order the GPU path
                # documents.  A poisoned primitive (a vertex at -inf beside a normal component of 1e20, say) can leave a
                # pixel hanging on the last bit of such a sum, and the BLAS behind np.dot rounds it in an order of its
                # own (seed 2203, scene 2109: one pixel, the two oracle variants disagree with EACH OTHER there) -- so
                # the BLAS variant may differ from the GPU only where it differs from the ordered one
                want = np_oracle.render(scene_to_numpy(scene, round_fp32=True), dots="ordered")
                blas = np_oracle.render(scene_to_numpy(scene, round_fp32=True))
                np.testing.assert_array_equal(ref["nearest"], want["nearest"])
                split = blas["nearest"] != want["nearest"]
                assert ((ref["nearest"] == blas["nearest"]) | split).all() and split.mean() < 0.01
                d = ref["depth"].astype(np.float64)
                assert np.all(np.isclose(d, want["depth"], rtol=DEPTH_RTOL, atol=0) | (d == want["depth"]))
                img = ref["image"].astype(np.float64)
                ok = np.isclose(img, want["image"], rtol=IMAGE_RTOL, atol=IMAGE_ATOL, equal_nan=True) | \
                    ((np.abs(want["image"]) > 3e38) & np.isinf(img))          # beyond float32 on our side
                assert ok.all(), f"scene {it}: image differs from the oracle on {(~ok).sum()} values"


@pytest.mark.gpu
def test_disc_whose_conic_coefficients_cancel_is_not_rejected():
    """Found by the non-finite fuzz with seed 2002 (scene 2425): a disc with centre y = -1e20 and radius -1e20 passes
    through the scene (|oc|^2 and r^2 agree to the last bit, the reference's fp64 test says hit), but the fp64
    coefficients of its image conic were rounding noise that looked like a small well-conditioned ellipse, and every
    accelerated mode rejected 166 pixels of it.  The reject record now checks how much of the coefficients survived
    the cancellation (srh_reject.h: conic_trusted) and stays "always a candidate" otherwise."""
    f32 = lambda a: np.asarray(a, dtype=np.float32)          # noqa: E731
    scene = {
        "camera": {"viewport": [0, 0, 200, 80], "fovy": 0.3490658503988659, "focal_length": 0.5,
                   "eye": [-0.22518337330819965, -0.17873926241833957, -0.08570136787524409, 1.0],
                   "at": [-0.052733267564547964, -0.40446551969026406, -0.02202172392123179, 1.0],
                   "up": [0.8508139646210928, 0.6642066983782233, 0.4076471168705725, 0.0], "near": 1.0, "far": 50.0},
        "lights": {"pos": f32([[3, 4, 5, 1], [-4, 2, 3, 1]]), "color_idx": np.array([1, 2])},
        "colors": f32([[0, 0, 0], [.8, .5, .4], [.3, .6, .9]]),
        "materials": {"albedo": f32([[.5, .5, .5], [.9, .3, .2]])},
        "objects": {"disk": {
            "pos": f32([[1.0083192586898804, -0.6783924698829651, -0.0375119112432003, 1.0],
                        [0.29687345027923584, -1.0000000200408773e+20, 0.36714065074920654, 1.0]]),
            "normal": f32([[-0.18393109738826752, 0.6064280867576599, 1.0000000031710769e-30, 0.0],
                           [-0.05883683264255524, 3.000000645916e-39, -0.07081481069326401, 0.0]]),
            "radius": f32([1.0000000200408773e+20, -1.0000000200408773e+20]),
            "material_idx": np.array([0, 1])}},
        "tonemap": {"type": "gamma", "gamma": 0.8}}
    ref = _render(scene, mode="exact")
    assert (ref["nearest"] == 1).sum() > 100                 # the huge disc is what most pixels see
    for mode, wpt in (("fast", 0), ("binned", 1), ("binned", 4)):
        got = _render(scene, mode=mode, waves_per_tile=wpt)
        for k in ("nearest", "depth", "image"):
            assert np.array_equal(got[k], ref[k], equal_nan=True), f"{mode}/{wpt}: {k} differs"


@pytest.mark.gpu
def test_render_views_refuses_what_it_cannot_do():
    from surf_renderer_amd import render_views, synthetic
    scene = synthetic.demo_scene(32, 24)
    cams = [scene["camera"]]
    with pytest.raises(ValueError, match="shadow"):                              # shadow rays belong to the torch semantics
        render_views(scene, cams, device="cuda:0", shadow=True)
    with pytest.raises(KeyError, match="not a leaf"):
        render_views(scene, cams, device="cuda:0", overrides=[{"disk.colour": np.zeros(3)}])
    with pytest.raises(ValueError, match="shape"):
        render_views(scene, cams, device="cuda:0", overrides=[{"lights.pos": np.zeros((7, 4), dtype=np.float32)}])
    with pytest.raises(TypeError, match="unexpected"):
        render_views(scene, cams, device="cuda:0", tile_size=4096)
    with pytest.raises(Exception, match="one projection per call"):           # perspective and orthographic views mixed
        render_views(scene, [scene["camera"], dict(scene["camera"], proj_type="ortho")], device="cuda:0", shading="torch")
    with pytest.raises(Exception, match="SRH_SHADING_TORCH"):                  # the numpy semantics have no ortho camera
        render_views(scene, [dict(scene["camera"], proj_type="ortho")], device="cuda:0")
    assert render_views(scene, cams, device="cuda:0", shadow=False)["image"].shape == (1, 24, 32, 3)


def test_render_views_from_two_threads_on_one_device():
    """srh_render_views keeps a per-device staging ring behind a mutex: two threads rendering different batches on
    their own streams at the same time each get exactly what per-view render() gives (more calls than ring slots,
    so slots are reused while the other thread is submitting)."""
    import threading
    from surf_renderer_amd import render, render_views, synthetic
    scene = synthetic.bunny_splat_scene(64, 48)
    rng = np.random.RandomState(11)

    def cams(n):
        out = []
        for _ in range(n):
            cam = dict(scene["camera"])
            eye = rng.normal(size=3)
            eye = 10.0 * eye / np.linalg.norm(eye)
            cam["eye"] = [float(eye[0]), float(eye[1]), float(eye[2]), 1.0]
            out.append(cam)
        return out

    jobs = [cams(11), cams(9)]
    want = [[render({**scene, "camera": c}, device="cuda:0") for c in job] for job in jobs]
    torch.cuda.synchronize()
    got, errors = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream("cuda:0")):
                res = None
                for _ in range(6):                      # 6 x ceil(n / 2) library calls per thread, 4 ring slots
                    res = render_views(scene, jobs[i], device="cuda:0", batch=2)
                torch.cuda.current_stream().synchronize()
                got[i] = res
        except Exception as exc:                        # noqa: BLE001 -- reported below, in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for v, single in enumerate(want[i]):
            for k in ("image", "depth", "nearest"):
                np.testing.assert_array_equal(got[i][k][v].cpu().numpy(), single[k].cpu().numpy(),
                                              err_msg=f"thread {i} view {v} {k}")


def test_render_views_refuses_stream_capture():
    from surf_renderer_amd import _lib, renderer, synthetic
    scene = synthetic.bunny_splat_scene(32, 32)
    buf = renderer.flatten_scene(scene, "cuda:0")
    cam = renderer.camera_struct(scene["camera"])
    img = torch.empty((2, 32, 32, 3), device="cuda:0")
    dep = torch.empty((2, 32, 32), device="cuda:0")
    ws = renderer.render_views_buffers(buf, [cam, cam], img, dep)       # sizes the workspace, warms the ring
    buf.ensure_workspace(32, 32)                                        # nothing may be allocated under capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(torch.cuda.Stream("cuda:0")):
        g.capture_begin(capture_error_mode="thread_local")
        try:
            with pytest.raises(_lib.SrhError, match="stream-captured"):
                renderer.render_views_buffers(buf, [cam, cam], img, dep, workspace=ws)
            renderer.render_buffers(buf, cam, out=(img[0], dep[0], None))      # srh_render_fwd may be captured
        finally:
            g.capture_end()
    g.replay()
    torch.cuda.synchronize()
    want, wdep, _ = renderer.render_buffers(buf, cam)
    torch.cuda.synchronize()
    assert torch.equal(img[0], want) and torch.equal(dep[0], wdep)



@pytest.mark.gpu
@pytest.mark.parametrize("shading,shadow", [("numpy", False), ("torch", False), ("torch", True)])
def test_views_with_their_own_geometry_lights_and_shadows(shading, shadow):
    """The batch loop of the reference's GAN assigns, per element, a new splat set (disk.pos / disk.normal), a new eye and
    a new position of light 0 before each render() (diffrend/torch/GAN/gan.py:325-378), and batch_render.py renders with
    shadow=True by default (:59,104-106).  One render_views call with per-view overrides (SrhParams.per_view) gives, view
    for view, exactly what render() gives for that element's scene -- image, depth, winners, visibility bits."""
    from surf_renderer_amd import render, render_views, synthetic
    rng = np.random.RandomState(5)
    base = synthetic.disk_cloud_scene(900, 96, 80, radius=0.08, seed=3)
    base["lights"]["attenuation"] = np.array([[1, 0, 0]] * len(base["lights"]["pos"]), dtype=np.float32)
    n = 5
    cams, overrides, scenes = [], [], []
    for v in range(n):
        e = rng.normal(size=3)
        e = e / np.linalg.norm(e) * 4.0
        cams.append(dict(base["camera"], eye=[*map(float, e), 1.0]))
        pos = np.concatenate([rng.uniform(-1, 1, (900, 3)), np.ones((900, 1))], 1).astype(np.float32)
        nrm = np.concatenate([rng.normal(size=(900, 3)), np.zeros((900, 1))], 1).astype(np.float32)
        lpos = np.array(base["lights"]["pos"], dtype=np.float32, copy=True)
        lpos[0, :3] = rng.uniform(-6, 6, 3)
        ov = {"disk.pos": torch.tensor(pos, device="cuda:0") if v % 2 else pos, "disk.normal": nrm, "lights.pos": lpos}
        if v == 3:
            ov = {"lights.pos": lpos}                       # a view that keeps the base geometry
        overrides.append(ov)
        sc = dict(base, camera=cams[v], lights=dict(base["lights"], pos=lpos),
                  objects={"disk": dict(base["objects"]["disk"], **({"pos": pos, "normal": nrm} if v != 3 else {}))})
        scenes.append(sc)
    kw = dict(shading=shading, **({"shadow": True} if shadow else {}))
    got = render_views(base, cams, device="cuda:0", overrides=overrides, **kw)
    for batch in (2, 0):                                     # two-view batches; one library call per view
        again = render_views(base, cams, device="cuda:0", overrides=overrides, batch=batch, **kw)
        for k in got:
            assert torch.equal(again[k], got[k]), f"batch={batch}: {k} differs"
    for v in range(n):
        one = render(scenes[v], device="cuda:0", **kw)
        assert torch.equal(got["image"][v], one["image"]), f"view {v}: image"
        assert torch.equal(got["depth"][v], one["depth"]), f"view {v}: depth"
        assert torch.equal(got["nearest"][v].to(torch.int64), one["nearest"]), f"view {v}: nearest"
        if shadow:
            assert torch.equal(got["visibility"][v], one["light_visibility"]), f"view {v}: visibility"
    assert not torch.equal(got["image"][0], got["image"][1])



def test_bare_cuda_device_means_the_current_device():
    """flatten_scene(device="cuda") -- torch reports tensors on "cuda:N", and every out-buffer check compares devices: a
    scene flattened on the bare name used to refuse its own preallocated outputs (renderer.bin_statistics raised)."""
    from surf_renderer_amd import renderer, synthetic
    scene = synthetic.disk_cloud_scene(500, 96, 64)
    buf = renderer.flatten_scene(scene, device="cuda")
    assert buf.device.index == torch.cuda.current_device()
    cam = renderer.camera_struct(scene["camera"])
    st = renderer.bin_statistics(buf, cam)
    assert st["executed_pair_tests"] > 0
    image = torch.empty((64, 96, 3), dtype=torch.float32, device="cuda")
    depth = torch.empty((64, 96), dtype=torch.float32, device="cuda")
    renderer.render_buffers(buf, cam, out=(image, depth, None))
    ref = renderer.render(scene, device="cuda:0")
    torch.cuda.synchronize()
    assert torch.equal(image, ref["image"]) and torch.equal(depth, ref["depth"])


def test_key_field_width_follows_the_tile_list_length():
    """The packed keys give the list position bits(n + 1) low bits for a tile with n entries (srh_binned.h:
    ord_mask_for), at most 12 -- so n just below, at and above every power of two up to the saturation point must all
    resolve like the all-pairs mode: n coincident-in-screen discs over ONE tile, depths a few 1e-4 apart so that
    neighbouring keys differ only in the bits the field width decides about, plus ties (equal depths: lowest index)."""
    rng = np.random.RandomState(77)
    for n in (1, 2, 3, 6, 7, 8, 14, 15, 16, 30, 31, 32, 62, 63, 64, 126, 127, 128, 254, 255, 256, 510, 511, 512,
              1022, 1023, 1024, 2046, 2047, 2048, 4093, 4094, 4095, 4096, 4200):
        pos = np.zeros((n, 4), dtype=np.float32)
        pos[:, 0] = rng.uniform(-0.01, 0.01, n)
        pos[:, 1] = rng.uniform(-0.01, 0.01, n)
        z = rng.uniform(-0.2, 0.2, n)
        z[rng.randint(0, n, max(1, n // 3))] = z[0]                     # ties with disc 0
        z += rng.randint(-2, 3, n) * 1e-4                               # near ties
        pos[:, 2] = z
        pos[:, 3] = 1.0
        nrm = np.zeros((n, 4), dtype=np.float32)
        nrm[:, :3] = rng.normal(size=(n, 3)) * 0.15 + np.array([0.0, 0.0, 1.0])
        scene = {
            "camera": {"proj_type": "perspective", "viewport": [0, 0, 32, 16], "fovy": 0.3, "focal_length": 1.0,
                       "eye": [0.0, 0.0, 4.0, 1.0], "up": [0.0, 1.0, 0.0, 0.0], "at": [0.0, 0.0, 0.0, 1.0], "near": 0.1, "far": 100.0},
            "lights": {"pos": np.array([[3, 4, 5, 1]], dtype=np.float32), "color_idx": np.array([1])},
            "colors": np.array([[0, 0, 0], [.8, .7, .6]], dtype=np.float32),
            "materials": {"albedo": np.array([[.6, .6, .6]], dtype=np.float32)},
            "tonemap": {"type": "gamma", "gamma": 0.8},
            "objects": {"disk": {"pos": pos, "normal": nrm, "radius": rng.uniform(0.1, 0.5, n).astype(np.float32),
                                 "material_idx": np.zeros(n, dtype=np.int64)}},
        }
        ref = _modes_identical(scene, modes=("exact", "binned"))
        assert np.isfinite(ref["depth"]).any(), n
