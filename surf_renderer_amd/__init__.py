"""surf_renderer_amd -- MI355X-native `hip` backend for DiffRend's render(scene) hot path.

    from surf_renderer_amd import render
    res = render(scene)            # same scene dict as diffrend.numpy.renderer.render
    res['image'], res['depth'], res['nearest']

The names below are imported on first use (PEP 562): `surf_renderer_amd.frame_writer`'s writer process is spawned, imports
this package on the way to its own module and needs numpy only -- not torch and the render library.
"""
import importlib

_RENDERER = ("render", "render_views", "ResidentScene", "CapturedStep", "ViewScenes", "flatten_scene", "render_buffers", "camera_struct",
             "generate_rays")
_SCENE = ("load_scene", "load_model", "load_obj", "load_splat", "obj_to_triangle_spec")

__all__ = [*_RENDERER, *_SCENE]


def __getattr__(name):
    if name in _RENDERER:
        return getattr(importlib.import_module(".renderer", __name__), name)
    if name in _SCENE:
        return getattr(importlib.import_module(".scene", __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted([*globals(), *__all__])
