"""GPU: the analytic HIP backward (srh_render_bwd through the autograd.Function of render()) against the gradient
oracle (oracle/torch_oracle.py, fp64 autograd, pinned to the reference torch backend's autograd).

Stated tolerance: per input array  |got - want| <= 2e-4 * max|want| + 1e-6  (fp64 per-pixel chain rule, fp32 atomic
accumulation over up to ~1e6 pixels)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import torch_oracle
from oracle.golden_io import load_case, unpack_scene

pytestmark = pytest.mark.gpu

LEAF_PATHS = {"lights.pos": ("lights", "pos"), "colors": ("colors",), "materials.albedo": ("materials", "albedo")}


def _hip_gradients(scene, g_img, g_dep):
    """Scene with torch leaves on the GPU -> render() -> backward; returns ({key: grad ndarray}, forward dict)."""
    from surf_renderer_amd import render
    import copy
    sc = copy.deepcopy(scene)
    leaves = {}

    def leaf(arr):
        return torch.tensor(np.asarray(arr, dtype=np.float32), device="cuda:0", requires_grad=True)

    for kind, grp in sc["objects"].items():
        for name in torch_oracle.LEAF_KEYS[kind]:
            grp[name] = leaves[f"{kind}.{name}"] = leaf(grp[name])
    sc["lights"]["pos"] = leaves["lights.pos"] = leaf(sc["lights"]["pos"])
    sc["colors"] = leaves["colors"] = leaf(sc["colors"])
    sc["materials"]["albedo"] = leaves["materials.albedo"] = leaf(sc["materials"]["albedo"])
    res = render(sc, device="cuda:0")
    assert res["image"].requires_grad and res["depth"].requires_grad
    hit = torch.isfinite(res["depth"].detach())
    loss = torch.sum(res["image"] * torch.as_tensor(g_img, dtype=torch.float32, device="cuda:0"))
    if g_dep is not None:
        gd = torch.as_tensor(g_dep, dtype=torch.float32, device="cuda:0")
        loss = loss + torch.sum(torch.where(hit, res["depth"] * gd, torch.zeros_like(gd)))
    loss.backward()
    torch.cuda.synchronize()
    fwd = {"nearest": res["nearest"].cpu().numpy(), "depth": res["depth"].detach().cpu().numpy().astype(np.float64)}
    return {k: v.grad.cpu().numpy().astype(np.float64) for k, v in leaves.items()}, fwd


def _check(scene, seed=0, with_depth=True):
    vp = scene["camera"]["viewport"]
    h, w = vp[3] - vp[1], vp[2] - vp[0]
    rng = np.random.RandomState(seed)
    g_img = rng.uniform(-1, 1, size=(h, w, 3)).astype(np.float32).astype(np.float64)
    g_dep = rng.uniform(-1, 1, size=(h, w)).astype(np.float32).astype(np.float64) if with_depth else None
    got, fwd = _hip_gradients(scene, g_img, g_dep)
    want = torch_oracle.gradients(scene, g_img, g_dep, ref=fwd)
    for key, w_arr in want.items():
        scale = np.abs(w_arr).max()
        np.testing.assert_allclose(got[key], w_arr, rtol=0, atol=2e-4 * scale + 1e-6, err_msg=key)
    return got, want


@pytest.mark.parametrize("case", ["g1_demo_64x48", "g2_demo_planes_64x48", "g8j_array_camera_reordered",
                                  "g8a_sphere_behind_camera", "g8f_camera_inside_sphere"])
def test_backward_matches_gradient_oracle(case):
    scene, _, _ = load_case(os.path.join(GOLDEN_DIR, case + ".npz"))
    if case == "g8f_camera_inside_sphere":
        pytest.skip("the centre pixel's normal is 0/0 by construction (reference quirk): gradients are nan there")
    got, want = _check(scene, seed=3)
    assert any(np.abs(v).max() > 0 for v in want.values())
    if "disk.radius" in got:
        assert np.all(got["disk.radius"] == 0)
    if "triangle.face" in got:
        assert np.all(got["triangle.face"][:, 1:, :] == 0) and np.all(got["triangle.face"][..., 3] == 0)


def test_backward_against_reference_autograd_fixture():
    npz = np.load(os.path.join(GOLDEN_DIR, "g9_torch_autograd.npz"), allow_pickle=False)
    scene = unpack_scene(npz)
    got, _ = _hip_gradients(scene, npz["grad_in/image"].astype(np.float64), npz["grad_in/depth"].astype(np.float64))
    for key in npz.files:
        if key.startswith("grad/"):
            name = key[5:]
            want = npz[key].astype(np.float64)
            g = got[name]
            if name == "lights.pos":
                g, want = g[:, :3], want[:, :3]
            np.testing.assert_allclose(g, want, atol=2e-3 * max(np.abs(want).max(), 1e-6), err_msg=name)


def test_backward_config4_bunny_mesh():
    """BASELINE config 4 (bunny.obj, d image / d vertex and d image / d normal) at a resolution the CPU gradient
    oracle handles in seconds, plus the no-tonemap and image-only variants."""
    from surf_renderer_amd import synthetic
    scene = synthetic.bunny_mesh_scene(160, 128)
    got, want = _check(scene, seed=7)
    assert np.abs(want["triangle.face"][:, 0, :3]).max() > 0 and np.abs(want["triangle.normal"]).max() > 0
    scene.pop("tonemap")
    _check(scene, seed=8, with_depth=False)


def test_backward_config4_at_its_own_size():
    """BASELINE config 4 as named: bunny.obj at 1024 x 1024, d image / d vertex and d image / d normal against the
    gradient oracle (O(pixels x lights) on the CPU once the forward pass supplies the winners: ~20 s)."""
    from surf_renderer_amd import synthetic
    scene = synthetic.bunny_mesh_scene(1024, 1024)
    got, want = _check(scene, seed=11)
    assert np.abs(want["triangle.face"][:, 0, :3]).max() > 0 and np.abs(want["triangle.normal"]).max() > 0
    assert np.all(got["triangle.face"][:, 1:, :] == 0)


def test_backward_plane_filling_a_large_frame():
    """One plane covers every pixel of a 2048 x 1024 frame: all four waves of every workgroup are one run of the same
    winner (the workgroup-merge path of scatter_primitive_grads, srh_backward.h) and two million per-pixel terms are
    accumulated into a handful of fp32 sums -- against the fp64 oracle, and twice, for the run-to-run spread of the
    atomic accumulation order."""
    f32 = lambda a: np.asarray(a, dtype=np.float32)          # noqa: E731
    scene = {"camera": {"viewport": [0, 0, 2048, 1024], "fovy": float(np.deg2rad(50.0)), "focal_length": 1.0,
                        "eye": [0.3, 2.0, 6.0, 1.0], "at": [0.0, 0.0, 0.0, 1.0], "up": [0.0, 1.0, 0.0, 0.0],
                        "near": 0.1, "far": 1000.0},
             "lights": {"pos": f32([[3, 6, 5, 1], [-4, 5, 3, 1], [0, 8, -2, 1]]), "color_idx": np.array([1, 2, 3])},
             "colors": f32([[0, 0, 0], [.8, .5, .4], [.3, .6, .9], [.5, .9, .3]]),
             "materials": {"albedo": f32([[.7, .6, .5]])},
             "objects": {"plane": {"pos": f32([[0, 0, -30, 1]]), "normal": f32([[0.05, 0.3, 1.0, 0]]),
                                   "material_idx": np.array([0])}},
             "tonemap": {"type": "gamma", "gamma": 0.8}}
    got, want = _check(scene, seed=5)
    assert np.abs(want["plane.pos"]).max() > 0 and np.abs(want["plane.normal"]).max() > 0
    a, _ = _hip_gradients(scene, np.ones((1024, 2048, 3)), None)
    b, _ = _hip_gradients(scene, np.ones((1024, 2048, 3)), None)
    for key in a:
        spread = np.abs(a[key] - b[key]).max()
        assert spread <= 2e-5 * max(np.abs(a[key]).max(), 1e-30), f"{key}: run-to-run spread {spread}"


def test_backward_disk_cloud_and_no_grad_inputs():
    from surf_renderer_amd import render, synthetic
    scene = synthetic.disk_cloud_scene(1500, 160, 120, radius=0.06, seed=3)
    _check(scene, seed=1)
    # only some inputs require grad: the others get none, and inference mode takes the plain path
    sc = synthetic.disk_cloud_scene(300, 64, 48, radius=0.1, seed=4)
    sc["materials"]["albedo"] = torch.tensor(np.asarray(sc["materials"]["albedo"], dtype=np.float32), requires_grad=True)
    res = render(sc, device="cuda:0")
    res["image"].sum().backward()
    assert sc["materials"]["albedo"].grad is not None and sc["materials"]["albedo"].grad.abs().max() > 0
    with torch.no_grad():
        assert not render(sc, device="cuda:0")["image"].requires_grad


def _torch_shading_scene(name):
    """Scenes with the torch backend's extra inputs: the reference-pinned gradient fixture (all four primitive
    types, specular materials, ambient, three attenuation laws), and a disc cloud lit from both sides."""
    import json
    if name.startswith(("g10", "g11")):
        npz = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        from oracle.golden_io import unpack_scene
        return unpack_scene(npz), json.loads(str(npz["kwargs"]))
    from surf_renderer_amd import synthetic
    from surf_renderer_amd.scene import scene_to_numpy
    sc = scene_to_numpy(synthetic.disk_cloud_scene(400, 64, 48, radius=0.12, seed=4), round_fp32=True)
    nrm = sc["objects"]["disk"]["normal"].copy()
    nrm[::2] *= -1.0
    sc["objects"]["disk"]["normal"] = nrm
    sc["lights"]["attenuation"] = np.array([[1, 0, 0], [0.5, 0.1, 0.01], [1, 0, 0.02], [0.3, 0.2, 0]], dtype=np.float64)
    sc["lights"]["ambient"] = np.array([0.02, 0.01, 0.03])
    sc["materials"]["coeffs"] = np.array([[0.8, 0.2, 5.0]])
    return sc, {"double_sided": True}


def _leaf_scene_tch(scene):
    """Copy of the scene whose differentiable arrays (incl. coeffs / attenuation / ambient) are GPU leaves."""
    import copy
    sc = copy.deepcopy(scene)
    leaves = {}

    def leaf(arr):
        return torch.tensor(np.asarray(arr, dtype=np.float32), device="cuda:0", requires_grad=True)

    for kind, grp in sc["objects"].items():
        for name in torch_oracle.LEAF_KEYS[kind]:
            grp[name] = leaves[f"{kind}.{name}"] = leaf(grp[name])
    sc["lights"]["pos"] = leaves["lights.pos"] = leaf(sc["lights"]["pos"])
    sc["lights"]["attenuation"] = leaves["lights.attenuation"] = leaf(sc["lights"]["attenuation"])
    sc["lights"]["ambient"] = leaves["lights.ambient"] = leaf(sc["lights"]["ambient"])
    sc["colors"] = leaves["colors"] = leaf(sc["colors"])
    sc["materials"]["albedo"] = leaves["materials.albedo"] = leaf(sc["materials"]["albedo"])
    sc["materials"]["coeffs"] = leaves["materials.coeffs"] = leaf(sc["materials"]["coeffs"])
    return sc, leaves


@pytest.mark.parametrize("name", ["g10_torch_autograd_phong", "g10_torch_autograd_phong_ds_quartic", "cloud_ds",
                                  "g11_torch_autograd_ortho"])
def test_torch_shading_backward_matches_gradient_oracle(name):
    """render(scene, shading='torch') under autograd: the analytic HIP backward of the Phong model against the fp64
    gradient oracle (itself pinned by the reference's torch autograd, tests/test_torch_oracle.py)."""
    from oracle import np_oracle_tch
    scene, kw = _torch_shading_scene(name)
    ref = np_oracle_tch.render(scene, **kw)
    H, W = ref["depth"].shape
    rng = np.random.RandomState(5)
    g_img = rng.uniform(-1, 1, size=(H, W, 3))
    g_dep = rng.uniform(-1, 1, size=(H, W))

    from surf_renderer_amd import render
    leaf_scene, leaves = _leaf_scene_tch(scene)
    res = render(leaf_scene, device="cuda:0", shading="torch", **kw)
    same = res["nearest"].cpu().numpy() == ref["nearest"]
    hit = ref["depth"] <= scene["camera"]["far"]
    assert (same | ~hit).mean() > 0.999
    far = float(scene["camera"]["far"])
    dep = res["depth"]
    loss = torch.sum(res["image"] * torch.as_tensor(g_img, dtype=torch.float32, device=dep.device)) + \
        torch.sum(torch.where(dep <= far, dep * torch.as_tensor(g_dep, dtype=torch.float32, device=dep.device),
                              torch.zeros_like(dep)))
    loss.backward()
    want = torch_oracle.gradients_tch(scene, g_img, g_dep, ref=ref, **kw)
    checked = 0
    for key, t in leaves.items():
        got = t.grad.cpu().numpy().astype(np.float64) if t.grad is not None else np.zeros(tuple(t.shape))
        w = want[key].reshape(got.shape)
        scale = max(np.abs(w).max(), 1e-9)
        np.testing.assert_allclose(got, w, atol=2e-4 * scale + 1e-6, err_msg=key)
        checked += 1
    assert checked >= 9


def test_torch_shading_backward_with_shadows():
    """shadow=True under autograd: visibility is a constant 0 / 1 factor per light (the reference computes it with
    comparisons), so the backward is the Phong backward with that factor -- against the gradient oracle fed the same
    visibility."""
    from oracle import np_oracle_tch
    from surf_renderer_amd import render
    scene, kw = _torch_shading_scene("g10_torch_autograd_phong")
    ref = np_oracle_tch.render(scene, shadow=True, **kw)
    H, W = ref["depth"].shape
    rng = np.random.RandomState(9)
    g_img = rng.uniform(-1, 1, size=(H, W, 3))
    leaf_scene, leaves = _leaf_scene_tch(scene)
    res = render(leaf_scene, device="cuda:0", shading="torch", shadow=True, **kw)
    torch.sum(res["image"] * torch.as_tensor(g_img, dtype=torch.float32, device="cuda:0")).backward()
    want = torch_oracle.gradients_tch(scene, g_img, None, ref=ref, visibility=ref["visibility"], **kw)
    plain = torch_oracle.gradients_tch(scene, g_img, None, ref=ref, **kw)
    assert np.abs(want["materials.albedo"] - plain["materials.albedo"]).max() > 1e-3    # shadows matter here
    for key, t in leaves.items():
        got = t.grad.cpu().numpy().astype(np.float64) if t.grad is not None else np.zeros(tuple(t.shape))
        w = want[key].reshape(got.shape)
        # a handful of grazing shadow rays may be decided differently: 0.5 % of the largest entry
        np.testing.assert_allclose(got, w, atol=5e-3 * max(np.abs(w).max(), 1e-9) + 1e-6, err_msg=key)


def test_fuzz_backward_against_the_gradient_oracles():
    """40 small random mixed scenes (all primitive types, cameras inside the cloud, sub-pixel to screen-filling
    primitives): the numpy-semantics backward, then the Phong backward with random double_sided / use_quartic /
    shadow, each against the fp64 gradient oracle.  Tolerance 5e-4 of the largest entry per array (random scenes
    hold grazing hits whose fp32 atomic sums cancel harder than the curated scenes above)."""
    from oracle import np_oracle_tch
    from surf_renderer_amd import render
    from surf_renderer_amd.scene import scene_to_numpy
    from test_hip_parity import _random_scene
    # SRH_FUZZ_BWD_SEED / SRH_FUZZ_BWD_SCENES: one-off campaigns with other seeds (profiles/r02_fuzz_campaign.txt)
    rng = np.random.RandomState(int(os.environ.get("SRH_FUZZ_BWD_SEED", "41")))
    n_scenes = int(os.environ.get("SRH_FUZZ_BWD_SCENES", "40"))

    def compare(tag, got, want, tol):
        for key, w in want.items():
            assert np.all(np.isfinite(w)), (tag, key)
            g = got[key].reshape(w.shape)
            np.testing.assert_allclose(g, w, rtol=0, atol=tol * max(np.abs(w).max(), 1e-9) + 1e-6, err_msg=f"{tag} {key}")

    done = 0
    while done < n_scenes:
        scene = _random_scene(rng)
        W, H = scene["camera"]["viewport"][2:]
        if W * H > 64 * 80 or sum(len(g["material_idx"]) for g in scene["objects"].values()) > 400:
            continue
        done += 1
        scene["camera"]["near"] = max(scene["camera"]["near"], 0.01)      # near <= 0: missed spheres hit with a 0/0 normal
        sc = scene_to_numpy(scene, round_fp32=True)
        g_img = rng.uniform(-1, 1, size=(H, W, 3)).astype(np.float32).astype(np.float64)
        g_dep = rng.uniform(-1, 1, size=(H, W)).astype(np.float32).astype(np.float64)
        got, fwd = _hip_gradients(sc, g_img, g_dep)
        compare(f"scene {done} numpy", got, torch_oracle.gradients(sc, g_img, g_dep, ref=fwd), 5e-4)

        sc["lights"]["attenuation"] = np.array([[1, 0, 0], [0.5, 0.1, 0.01]])
        sc["lights"]["ambient"] = np.array([0.01, 0.02, 0.01])
        sc["materials"]["coeffs"] = np.array([[1, 0, 0], [0.7, 0.3, 5], [0.5, 0.5, 20]])
        kw = {"double_sided": bool(rng.randint(2)), "use_quartic": bool(rng.randint(2))}
        shadow = bool(rng.randint(2))
        ref = np_oracle_tch.render(sc, shadow=shadow, **kw)
        leaf_scene, leaves = _leaf_scene_tch(sc)
        res = render(leaf_scene, device="cuda:0", shading="torch", shadow=shadow, **kw)
        assert np.array_equal(res["nearest"].cpu().numpy(), ref["nearest"])
        dep, far = res["depth"], float(sc["camera"]["far"])
        loss = torch.sum(res["image"] * torch.as_tensor(g_img, dtype=torch.float32, device=dep.device)) + \
            torch.sum(torch.where(dep <= far, dep * torch.as_tensor(g_dep, dtype=torch.float32, device=dep.device),
                                  torch.zeros_like(dep)))
        loss.backward()
        want = torch_oracle.gradients_tch(sc, g_img, g_dep, ref=ref, visibility=ref["visibility"] if shadow else None, **kw)
        got = {k: (t.grad.cpu().numpy().astype(np.float64) if t.grad is not None else np.zeros(tuple(t.shape)))
               for k, t in leaves.items()}
        compare(f"scene {done} torch {kw} shadow={shadow}", got, want, 5e-3 if shadow else 5e-4)


def test_resident_scene_matches_render_and_sees_in_place_updates():
    """ResidentScene flattens once: same outputs and gradients as render(scene), and an in-place update of a leaf
    (what torch.optim does) is what the next render() sees."""
    import torch
    from surf_renderer_amd import ResidentScene, render, synthetic
    scene = synthetic.demo_scene(72, 56, with_planes=True)
    pos = torch.tensor(np.asarray(scene["objects"]["disk"]["pos"], dtype=np.float32), device="cuda:0", requires_grad=True)
    nrm = torch.tensor(np.asarray(scene["objects"]["disk"]["normal"], dtype=np.float32), device="cuda:0", requires_grad=True)
    scene["objects"]["disk"] = dict(scene["objects"]["disk"], pos=pos, normal=nrm)

    def loss(res):
        return (res["image"] ** 2).sum() + 0.1 * res["depth"].clamp(max=50.0).sum()

    a = render(scene, device="cuda:0")
    loss(a).backward()
    want = (pos.grad.clone(), nrm.grad.clone())
    pos.grad = None
    nrm.grad = None
    rs = ResidentScene(scene, device="cuda:0")
    b = rs.render()
    assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"])
    assert torch.equal(a["nearest"], b["nearest"].to(torch.int64))
    loss(b).backward()
    torch.testing.assert_close(pos.grad, want[0], rtol=1e-4, atol=1e-6 * float(want[0].abs().max()))
    torch.testing.assert_close(nrm.grad, want[1], rtol=1e-4, atol=1e-6 * float(want[1].abs().max()))
    with torch.no_grad():
        pos[:, 2] += 0.25                                # in place, as an optimiser step
    c = rs.render()
    d = render(scene, device="cuda:0")
    assert torch.equal(c["image"], d["image"]) and not torch.equal(c["image"], a["image"])


@pytest.mark.parametrize("how", ["float64", "cpu", "strided"])
def test_resident_scene_refuses_leaves_it_would_have_to_copy(how):
    """A differentiable leaf that is float64, on the CPU or not contiguous cannot be used in place: flatten_scene would
    copy it once, the optimiser would keep stepping the leaf, and every later render() would draw the first iteration's
    values.  ResidentScene says so instead of going stale; render(scene), which converts per call, still takes it."""
    import torch
    from surf_renderer_amd import ResidentScene, render, synthetic
    scene = synthetic.demo_scene(48, 40, with_planes=True)
    base = np.asarray(scene["objects"]["disk"]["pos"], dtype=np.float32)
    if how == "float64":
        pos = torch.tensor(base.astype(np.float64), device="cuda:0", requires_grad=True)
    elif how == "cpu":
        pos = torch.tensor(base, requires_grad=True)
    else:
        wide = torch.zeros((base.shape[0], 8), device="cuda:0")
        wide[:, :4] = torch.tensor(base, device="cuda:0")
        pos = wide[:, :4].detach().requires_grad_(True)
        assert not pos.is_contiguous()
    scene["objects"]["disk"] = dict(scene["objects"]["disk"], pos=pos)
    with pytest.raises(ValueError, match="disk.pos"):
        ResidentScene(scene, device="cuda:0")
    res = render(scene, device="cuda:0")
    (res["image"] ** 2).sum().backward()
    assert pos.grad is not None and bool(torch.isfinite(pos.grad).all())


def test_captured_step_equals_the_eager_iteration():
    """ResidentScene.capture_step: render + loss + backward as one hipGraph.  Loss and gradients of a replay equal the
    eager iteration's, and a replay after an in-place update of the leaves (an optimiser step) follows it."""
    import torch
    from surf_renderer_amd import ResidentScene, synthetic
    scene = synthetic.bunny_mesh_scene(160, 128)
    tri = scene["objects"]["triangle"]
    face = torch.tensor(np.asarray(tri["face"], dtype=np.float32), device="cuda:0", requires_grad=True)
    normal = torch.tensor(np.asarray(tri["normal"], dtype=np.float32), device="cuda:0", requires_grad=True)
    scene["objects"]["triangle"] = dict(tri, face=face, normal=normal)
    rs = ResidentScene(scene, device="cuda:0")
    target = torch.rand((128, 160, 3), device="cuda:0")

    def loss_fn(res):
        return ((res["image"] - target) ** 2).sum() + 0.01 * res["depth"].clamp(max=50.0).sum()

    def eager():
        face.grad = None
        normal.grad = None
        loss = loss_fn(rs.render())
        loss.backward()
        return loss.detach().clone(), face.grad.clone(), normal.grad.clone()

    step = rs.capture_step(loss_fn)
    for it in range(3):
        want = eager()
        face.grad = None                                 # as optimiser.zero_grad(set_to_none=True) leaves them
        normal.grad = None
        got_loss = step.replay().clone()
        torch.cuda.synchronize()
        assert face.grad is step.grads[0] and normal.grad is step.grads[1]
        got = (got_loss, face.grad.clone(), normal.grad.clone())
        torch.testing.assert_close(got[0], want[0], rtol=1e-5, atol=0)
        for g, w in zip(got[1:], want[1:]):
            torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-6 * float(w.abs().max()))
        with torch.no_grad():                            # an optimiser step, in place
            face[:, 0, :3] += 0.002 * torch.randn_like(face[:, 0, :3])
            target.copy_(torch.rand_like(target))

