"""GPU: the frame pipeline bench.py times (surf_renderer_amd.pipeline.FramePipeline) -- hipGraph replays on three
streams must give the eager frame bit for bit.  bench.py runs the same check after its timed loop; this test makes a
wrong graph (stale slot / scratch pairing, a dropped node) fail the suite instead of printing a fast number."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(n=20000, w=512, h=384):
    from surf_renderer_amd import synthetic
    return synthetic.disk_cloud_scene(n, w, h)


def _pipeline(scene, **kw):
    from surf_renderer_amd import renderer
    from surf_renderer_amd.pipeline import FramePipeline
    buf = renderer.flatten_scene(scene, "cuda:0")
    cam = renderer.camera_struct(scene["camera"])
    return FramePipeline(buf, cam, **kw)


@pytest.mark.parametrize("rows", [None, (64, 208)])
def test_graph_replays_on_three_streams_equal_the_eager_frame(rows):
    pipe = _pipeline(_scene(), rows=rows, n_inflight=3, graphs=True, strict_graphs=True)
    assert pipe.captured == 3
    for _ in range(7):
        pipe.submit()
    pipe.poison()                      # from here on only replays write the slabs
    for _ in range(11):
        pipe.submit()
    assert pipe.verify() == 3


def test_poisoned_slab_is_detected():
    pipe = _pipeline(_scene(2000, 128, 96), n_inflight=3, graphs=True, strict_graphs=True)
    for _ in range(3):
        pipe.submit()
    pipe.sync()
    pipe.slabs[1].view(torch.int32)[5, 7] ^= 1      # what a stale / partial replay would look like
    with pytest.raises(RuntimeError, match="differs from the eager render"):
        pipe.verify()


def test_eager_frames_with_timing_events_interleave_with_replays():
    from surf_renderer_amd import _lib
    pipe = _pipeline(_scene(5000, 256, 256), n_inflight=3, graphs=True, strict_graphs=True)
    pairs = []
    pipe.poison()
    for i in range(12):
        ev = _lib.EventPair() if i % 4 == 0 else None
        if ev:
            pairs.append(ev)
        pipe.submit(ev)
    assert pipe.verify() == 3
    for ev in pairs:
        assert 0.0 < ev.elapsed_ms() < 1000.0
        ev.close()


def test_pipeline_frame_matches_render():
    """The slab layout [W x rgb | W x depth] per row holds the same image and depth that render() returns."""
    from surf_renderer_amd import render
    from surf_renderer_amd.pipeline import slab_views
    scene = _scene(3000, 160, 96)
    pipe = _pipeline(scene, n_inflight=2, graphs=True, strict_graphs=True)
    pipe.submit()
    pipe.sync()
    image, depth = slab_views(pipe.slabs[0], 160)
    want = render(scene, device="cuda:0")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(image.cpu().numpy(), want["image"].cpu().numpy())
    np.testing.assert_array_equal(depth.cpu().numpy(), want["depth"].cpu().numpy())


@pytest.mark.parametrize("kw", [dict(schedule="stages", render_streams=2), dict(schedule="stages", render_streams=1),
                                dict(schedule="stages", render_streams=2, prioritise_render=False),
                                dict(schedule="frames", rotate=2)])
def test_other_schedules_equal_the_eager_frame(kw):
    """The stage schedule (every frame's binning on one stream, the render kernels on others, tied by events; the
    binning and the render halves of a frame are separate graphs -- SrhParams.stages) and rotating scratch / slab
    pairs give the eager frame bit for bit, with timing-event frames (eager launches) mixed in."""
    from surf_renderer_amd import _lib
    pipe = _pipeline(_scene(), n_inflight=3, graphs=True, strict_graphs=True, mode="binned", **kw)
    for _ in range(5):
        pipe.submit()
    pipe.poison()
    pairs = []
    for i in range(13):
        ev = _lib.EventPair() if i % 5 == 2 else None
        if ev:
            pairs.append(ev)
        pipe.submit(ev)
    assert pipe.verify() == min(pipe.n, 13)
    for ev in pairs:
        assert 0.0 < ev.elapsed_ms() < 1000.0
        ev.close()
