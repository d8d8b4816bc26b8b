"""The gradient oracle (oracle/torch_oracle.py): its forward against the reference's golden outputs, and its
autograd against gradients the reference's own torch backend produced (tests/golden/g9_torch_autograd.npz,
oracle/gen_golden_grad.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import torch_oracle
from oracle.golden_io import load_case, unpack_scene


@pytest.mark.parametrize("case", ["g1_demo_64x48", "g2_demo_planes_64x48", "g7_disk_cloud_3000_r08_64x64",
                                  "g8a_sphere_behind_camera", "g8g_near_zero_sphere_miss"])
def test_forward_matches_reference_output(case):
    scene, want, _ = load_case(os.path.join(GOLDEN_DIR, case + ".npz"))
    leaves = torch_oracle.make_leaves(scene, requires_grad=False)
    image, depth, _ = torch_oracle.render(scene, leaves, ref=want)
    np.testing.assert_allclose(depth.numpy(), want["depth"], rtol=1e-12)
    np.testing.assert_allclose(image.numpy(), want["image"], rtol=1e-10, atol=1e-13)


def test_autograd_matches_reference_torch_backend():
    npz = np.load(os.path.join(GOLDEN_DIR, "g9_torch_autograd.npz"), allow_pickle=False)
    scene = unpack_scene(npz)
    # forwards agree in this regime (the fixture's forward is float32)
    leaves = torch_oracle.make_leaves(scene, requires_grad=False)
    image, depth, hit = torch_oracle.render(scene, leaves)
    assert bool(hit.all())
    np.testing.assert_allclose(image.numpy(), npz["ref/image"], atol=2e-5)
    np.testing.assert_allclose(depth.numpy(), npz["ref/depth"], rtol=2e-5)
    grads = torch_oracle.gradients(scene, npz["grad_in/image"].astype(np.float64), npz["grad_in/depth"].astype(np.float64))
    checked = 0
    for key in npz.files:
        if not key.startswith("grad/"):
            continue
        name = key[5:]
        want = npz[key].astype(np.float64)
        got = grads[name]
        if name == "lights.pos":
            got, want = got[:, :3], want[:, :3]
        scale = max(np.abs(want).max(), 1e-6)
        np.testing.assert_allclose(got, want, atol=2e-3 * scale, err_msg=name)
        checked += 1
    assert checked == 10
    assert np.all(grads["disk.radius"] == 0)                       # Q: a disc's radius has no gradient
    assert np.all(grads["triangle.face"][:, 1:, :] == 0)           # only vertex 0 of a triangle is differentiated
    assert np.abs(grads["triangle.face"][:, 0, :3]).max() > 0


def test_gradients_are_consistent_with_finite_differences():
    scene, want, _ = load_case(os.path.join(GOLDEN_DIR, "g1_demo_64x48.npz"))
    rng = np.random.RandomState(0)
    g_img = rng.uniform(-1, 1, size=want["image"].shape)
    grads = torch_oracle.gradients(scene, g_img, ref=want)

    def loss(sc):
        leaves = torch_oracle.make_leaves(sc, requires_grad=False)
        image, _, _ = torch_oracle.render(sc, leaves, ref=want)
        return float(torch.sum(image * torch.as_tensor(g_img)))

    import copy
    for key, idx in (("disk.pos", (0, 2)), ("disk.normal", (2, 0)), ("sphere.radius", (0,)), ("sphere.pos", (1, 1)),
                     ("triangle.face", (0, 0, 2)), ("materials.albedo", (4, 1)), ("lights.pos", (1, 0))):
        eps = 1e-6
        vals = []
        for sign in (+1, -1):
            sc = copy.deepcopy(scene)
            parts = key.split(".")
            arr = sc["objects"][parts[0]][parts[1]] if parts[0] in sc["objects"] else \
                (sc["lights"]["pos"] if key == "lights.pos" else sc["materials"]["albedo"])
            arr[idx] += sign * eps
            vals.append(loss(sc))
        fd = (vals[0] - vals[1]) / (2 * eps)
        np.testing.assert_allclose(grads[key][idx], fd, rtol=2e-5, atol=1e-7, err_msg=f"{key}{idx}")


@pytest.mark.parametrize("case", ["g10_torch_autograd_phong", "g10_torch_autograd_phong_ds_quartic", "g11_torch_autograd_ortho"])
def test_phong_autograd_matches_reference_torch_backend(case):
    """Torch-backend semantics (attenuation, specular, ambient, per-light relu, double_sided, use_quartic): the fp64
    gradient oracle against what the reference's torch backend produced under autograd (float32).  The reference's
    sphere gradients are NaN (sqrt under a mask, torch/utils.py:238-279), so the sphere leaves are not compared."""
    import json
    npz = np.load(os.path.join(GOLDEN_DIR, case + ".npz"), allow_pickle=False)
    scene = unpack_scene(npz)
    kw = json.loads(str(npz["kwargs"]))
    leaves = torch_oracle.make_leaves_tch(scene, requires_grad=False)
    image, depth, hit = torch_oracle.render_tch(scene, leaves, **kw)
    assert bool(hit.all())
    same = np.asarray(npz["ref/nearest"]) == np.asarray(__import__("oracle.np_oracle_tch", fromlist=["x"]).render(scene, **kw)["nearest"])
    assert same.mean() > 0.995
    np.testing.assert_allclose(image.numpy()[same], npz["ref/image"][same], atol=3e-4)
    np.testing.assert_allclose(depth.numpy()[same], npz["ref/depth"][same], rtol=2e-5)
    grads = torch_oracle.gradients_tch(scene, npz["grad_in/image"].astype(np.float64),
                                       npz["grad_in/depth"].astype(np.float64), **kw)
    checked = 0
    for key in npz.files:
        if not key.startswith("grad/") or key.startswith("grad/sphere."):
            continue
        name = key[5:]
        want = npz[key].astype(np.float64)
        got = grads[name]
        if name == "lights.pos":
            got, want = got[:, :3], want[:, :3]
        if name in ("plane.pos", "disk.pos"):
            got, want = got[:, :3], want[:, :3]
        scale = max(np.abs(want).max(), 1e-6)
        # float32 reference, a handful of silhouette pixels decided differently: 1 % of the largest entry
        np.testing.assert_allclose(got, want, atol=1e-2 * scale, err_msg=name)
        checked += 1
    assert checked == 13
    assert np.isnan(npz["grad/sphere.pos"]).any()                  # documents why spheres are left out
    assert np.all(np.isfinite(grads["sphere.pos"])) and np.abs(grads["sphere.pos"]).max() > 0


def test_phong_gradients_are_consistent_with_finite_differences():
    """The torch-semantics oracle's autograd against central differences of its own forward, in particular for the
    sphere leaves, which the reference fixture cannot pin (its sphere gradients are NaN)."""
    import copy
    from oracle import np_oracle_tch
    npz = np.load(os.path.join(GOLDEN_DIR, "g10_torch_autograd_phong.npz"), allow_pickle=False)
    scene = unpack_scene(npz)
    ref = np_oracle_tch.render(scene)
    rng = np.random.RandomState(1)
    g_img = rng.uniform(-1, 1, size=ref["image"].shape)
    grads = torch_oracle.gradients_tch(scene, g_img, ref=ref)

    def loss(sc):
        leaves = torch_oracle.make_leaves_tch(sc, requires_grad=False)
        image, _, _ = torch_oracle.render_tch(sc, leaves, ref=ref)
        return float(torch.sum(image * torch.as_tensor(g_img)))

    def target(sc, key):
        a, b = key.split(".")
        return sc["objects"][a][b] if a in sc["objects"] else sc[a][b]

    for key, idx in (("sphere.pos", (0, 1)), ("sphere.pos", (1, 2)), ("sphere.radius", (0,)), ("sphere.radius", (1,)),
                     ("materials.coeffs", (1, 0)), ("materials.coeffs", (2, 1)), ("materials.coeffs", (1, 2)),
                     ("lights.attenuation", (1, 1)), ("lights.attenuation", (2, 2)), ("lights.ambient", (1,)),
                     ("disk.normal", (0, 1)), ("plane.pos", (0, 2)), ("lights.pos", (0, 0))):
        eps = 1e-6
        vals = []
        for sign in (+1, -1):
            sc = copy.deepcopy(scene)
            target(sc, key)[idx] += sign * eps
            vals.append(loss(sc))
        fd = (vals[0] - vals[1]) / (2 * eps)
        np.testing.assert_allclose(grads[key][idx], fd, rtol=5e-5, atol=2e-7, err_msg=f"{key}{idx}")
