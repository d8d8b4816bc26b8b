"""The asynchronous image writer (surf_renderer_amd.frame_writer) that stands for the writer process of
diffrend/torch/batch_render.py:36-53,165-179: N frames in -> N image + N depth PNGs out, with the reference's
conversions; queue back-pressure; sentinel shutdown.  CPU tests use host arrays; the GPU test renders the frames."""
import os

import numpy as np
import pytest

from surf_renderer_amd.frame_writer import FrameWriter, encode_frame, read_png, write_png


def _frames(n, h=24, w=32, seed=0):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        image = rng.rand(h, w, 3).astype(np.float32)
        depth = (1.0 + 9.0 * rng.rand(h, w)).astype(np.float32)
        depth[rng.rand(h, w) < 0.3] = 1001.0               # far + 1 background of the torch semantics
        out.append((image, depth))
    return out


def test_png_round_trip():
    rng = np.random.RandomState(1)
    for shape in ((5, 7, 3), (6, 4)):
        img = rng.randint(0, 256, size=shape).astype(np.uint8)
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"srh_png_{len(shape)}.png")
        write_png(path, img)
        np.testing.assert_array_equal(read_png(path), img)
        os.remove(path)


def test_encode_frame_follows_the_reference_conversions():
    image, depth = _frames(1)[0]
    im, dz = encode_frame(image, depth, 1000.0)
    np.testing.assert_array_equal(im, np.uint8(255.0 * image))
    d = depth.copy()
    d[d >= 1000.0] = d.min()
    np.testing.assert_array_equal(dz, np.uint8(255.0 * (d - d.min()) / (d.max() - d.min())))
    assert dz[depth >= 1000.0].max() == 0                  # background sits at the minimum


def test_n_frames_written_equal_n_frames_submitted(tmp_path):
    frames = _frames(9)
    with FrameWriter(str(tmp_path), queue_size=2, also_npy=True) as w:      # queue smaller than the job: back-pressure
        for i, (image, depth) in enumerate(frames):
            w.put(f"_{i}", image, depth, 1000.0)
        assert w.submitted == 9
    names = sorted(os.listdir(tmp_path))
    assert len([n for n in names if n.endswith(".png")]) == 18
    for i, (image, depth) in enumerate(frames):
        im, dz = encode_frame(image, depth, 1000.0)
        np.testing.assert_array_equal(read_png(str(tmp_path / f"img_{i}.png")), im)
        np.testing.assert_array_equal(read_png(str(tmp_path / f"depth_{i}.png")), dz)
        np.testing.assert_array_equal(np.load(tmp_path / f"img_{i}.npy"), image)
    with pytest.raises(RuntimeError):
        w.put("_x", *frames[0], 1000.0)                    # closed


@pytest.mark.gpu
def test_rendered_views_reach_the_files(tmp_path):
    import torch
    from surf_renderer_amd import render, synthetic
    from surf_renderer_amd.frame_writer import render_views_to_files
    scene = synthetic.bunny_splat_scene(48, 40)
    rng = np.random.RandomState(3)
    cams = []
    for _ in range(7):
        eye = rng.normal(size=3)
        eye = 10.0 * eye / np.linalg.norm(eye)
        cams.append(dict(scene["camera"], eye=[float(eye[0]), float(eye[1]), float(eye[2]), 1.0]))
    assert render_views_to_files(scene, cams, str(tmp_path), batch=3, device="cuda:0") == 7
    for i, cam in enumerate(cams):
        res = render({**scene, "camera": cam}, device="cuda:0")
        torch.cuda.synchronize()
        im, dz = encode_frame(res["image"].cpu().numpy(), res["depth"].cpu().numpy(), cam["far"])
        np.testing.assert_array_equal(read_png(str(tmp_path / f"img_{i}.png")), im)
        np.testing.assert_array_equal(read_png(str(tmp_path / f"depth_{i}.png")), dz)


@pytest.mark.gpu
def test_shadowed_views_reach_the_files(tmp_path):
    """batch_render.py renders with shadow=b_shadow, and --shadow defaults to True (:59,104-106,138): the loop on
    render_views has to do the same."""
    import torch
    from surf_renderer_amd import render, synthetic
    from surf_renderer_amd.frame_writer import render_views_to_files
    scene = synthetic.demo_scene(56, 40, with_planes=True)
    rng = np.random.RandomState(4)
    cams = []
    for _ in range(4):
        eye = rng.normal(size=3)
        eye = 9.0 * eye / np.linalg.norm(eye)
        cams.append(dict(scene["camera"], eye=[float(eye[0]), abs(float(eye[1])) + 1.0, float(eye[2]), 1.0]))
    kw = dict(shading="torch", shadow=True)
    assert render_views_to_files(scene, cams, str(tmp_path), batch=3, device="cuda:0", **kw) == 4
    lit = 0
    for i, cam in enumerate(cams):
        res = render({**scene, "camera": cam}, device="cuda:0", **kw)
        plain = render({**scene, "camera": cam}, device="cuda:0", shading="torch")
        torch.cuda.synchronize()
        lit += int(not torch.equal(res["image"], plain["image"]))
        im, dz = encode_frame(res["image"].cpu().numpy(), res["depth"].cpu().numpy(), cam["far"])
        np.testing.assert_array_equal(read_png(str(tmp_path / f"img_{i}.png")), im)
        np.testing.assert_array_equal(read_png(str(tmp_path / f"depth_{i}.png")), dz)
    assert lit > 0                                           # the shadow pass changed something

