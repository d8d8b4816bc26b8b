"""surf_renderer_amd -- MI355X-native `hip` backend for DiffRend's render(scene) hot path.

    from surf_renderer_amd import render
    res = render(scene)            # same scene dict as diffrend.numpy.renderer.render
    res['image'], res['depth'], res['nearest']
"""
from .renderer import (render, render_views, flatten_scene, render_buffers, camera_struct, generate_rays,  # noqa: F401
                       ResidentScene)
from .scene import load_scene, load_model, load_obj, load_splat, obj_to_triangle_spec  # noqa: F401

__all__ = ["render", "render_views", "ResidentScene", "flatten_scene", "render_buffers", "camera_struct", "generate_rays",
           "load_scene", "load_model", "load_obj", "load_splat", "obj_to_triangle_spec"]
