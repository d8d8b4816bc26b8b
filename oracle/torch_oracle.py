"""Gradient oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A differentiable fp64 PyTorch restatement of the reference's numpy forward (``diffrend/numpy/renderer.py:204-272``)
used to check the hip backend's analytic backward.  Which primitive wins each pixel is taken from the numpy oracle
(``np_oracle.render``): the argmin and every mask of the reference are piecewise constant, so -- exactly as in the
reference's own differentiable backend (``diffrend/torch/renderer.py:136-355`` with ``torch/utils.py:238-366``,
where ``where`` is ``cond.float()*x + (1-cond)*y`` and ``min(0)`` routes the gradient to the winner) -- gradients flow
only through the winner's hit distance, hit point, normal, albedo and the lights.  Consequences that the analytic
backward reproduces (SURVEY.md section 8, row a-B): there are no silhouette gradients; a disc's radius gets zero
gradient; of a triangle only vertex 0 (the plane point, ``torch/utils.py:340``) and the supplied normal get
gradients.

Parity status: forward PINNED by the golden vectors (``tests/test_oracle_golden.py``); gradients PINNED by
``tests/golden/g9_torch_autograd.npz``, produced by running the reference's torch backend under autograd on a scene
where its shading model coincides with the numpy one (``oracle/gen_golden_grad.py``).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch

from . import np_oracle

LEAF_KEYS = {
    "disk": ("pos", "normal", "radius"),
    "plane": ("pos", "normal"),
    "sphere": ("pos", "radius"),
    "triangle": ("face", "normal"),
}


def make_leaves(scene: Dict[str, Any], requires_grad: bool = True) -> Dict[str, torch.Tensor]:
    """fp64 leaf tensors for every differentiable input of an ndarray-leaf scene, keyed '<kind>.<field>',
    'lights.pos', 'colors', 'materials.albedo'."""
    leaves: Dict[str, torch.Tensor] = {}
    for kind, grp in scene["objects"].items():
        for name in LEAF_KEYS[kind]:
            leaves[f"{kind}.{name}"] = torch.tensor(np.asarray(grp[name], dtype=np.float64), requires_grad=requires_grad)
    leaves["lights.pos"] = torch.tensor(np.asarray(scene["lights"]["pos"], dtype=np.float64), requires_grad=requires_grad)
    leaves["colors"] = torch.tensor(np.asarray(scene["colors"], dtype=np.float64), requires_grad=requires_grad)
    leaves["materials.albedo"] = torch.tensor(np.asarray(scene["materials"]["albedo"], dtype=np.float64),
                                              requires_grad=requires_grad)
    return leaves


def _unit(v: torch.Tensor) -> torch.Tensor:
    """ops.normalize: divide by the norm, by 1 where it is 0 (numpy/ops.py:18-26)."""
    n = torch.sqrt(torch.sum(v * v, dim=-1, keepdim=True))
    return v / torch.where(n > 0, n, torch.ones_like(n))


def render(scene: Dict[str, Any], leaves: Dict[str, torch.Tensor], ref: Optional[Dict[str, np.ndarray]] = None):
    """Differentiable image (H,W,3) and depth (H,W).  ``ref`` = {'nearest', 'depth'} of the same scene (which
    primitive wins each pixel, and -- through depth being finite -- whether the pixel is hit at all); computed with
    the numpy oracle if not given.  ``scene`` supplies camera, index arrays and tonemap; ``leaves`` the
    differentiable arrays."""
    cam = scene["camera"]
    eye_np, ray_np, H, W = np_oracle.generate_rays(cam)
    if ref is None:
        ref = np_oracle.render(scene)
    nearest = np.asarray(ref["nearest"]).reshape(-1)
    hit_np = np.isfinite(np.asarray(ref["depth"]).reshape(-1))
    npix = H * W
    eye = torch.tensor(eye_np[:3])
    d = torch.tensor(ray_np[:3].T.copy())                                   # (N,3)
    orig = eye[None, :].expand(npix, 3)                                      # the numpy backend is perspective only

    t = torch.zeros(npix, dtype=torch.float64)
    nrm = torch.zeros((npix, 3), dtype=torch.float64)
    mat = np.zeros(npix, dtype=np.int64)
    start = 0
    for kind, grp in scene["objects"].items():
        count = (grp["face"] if kind == "triangle" else grp["pos"]).shape[0]
        sel = np.nonzero((nearest >= start) & (nearest < start + count))[0]
        if sel.size:
            loc = torch.as_tensor(nearest[sel] - start)
            ds = d[sel]
            mat[sel] = np.asarray(grp["material_idx"])[nearest[sel] - start]
            if kind == "sphere":
                c = leaves["sphere.pos"][loc][:, :3]
                r = leaves["sphere.radius"][loc]
                oc = orig[sel] - c
                a = torch.sum(ds * ds, dim=-1)
                b = 2 * torch.sum(oc * ds, dim=-1)
                cc = torch.sum(oc * oc, dim=-1) - r * r
                disc = b * b - 4 * a * cc
                ok = disc >= 0
                root = torch.sqrt(torch.where(ok, disc, torch.zeros_like(disc)))
                t1 = (-b - root) / (2 * a)
                t2 = (-b + root) / (2 * a)
                one = torch.ones_like(t1)
                t1 = torch.where(ok & (t1 >= 0), t1, one)
                t2 = torch.where(ok & (t2 >= 0), t2, one)
                ts = torch.where(ok, torch.minimum(t1, t2), torch.zeros_like(t1))
                p = orig[sel] + ts[:, None] * ds
                v = p - c
                n = v / torch.sqrt(torch.sum(v * v, dim=-1, keepdim=True))
                n = torch.where(ok[:, None], n, torch.zeros_like(n))
            else:
                q = (leaves["triangle.face"][loc][:, 0, :3] if kind == "triangle" else leaves[f"{kind}.pos"][loc][:, :3])
                n = _unit(leaves[f"{kind}.normal"][loc])[:, :3]
                ts = torch.sum(n * (q - orig[sel]), dim=-1) / torch.sum(n * ds, dim=-1)
            t = t.index_put((torch.as_tensor(sel),), ts)
            nrm = nrm.index_put((torch.as_tensor(sel),), n)
        start += count

    hit = torch.as_tensor(hit_np)
    # pixels that are not hit are background: depth inf, image tonemap(0)
    p = orig + t[:, None] * d
    lpos = leaves["lights.pos"][:, :3]
    lcol = leaves["colors"][np.asarray(scene["lights"]["color_idx"])]
    alb = leaves["materials.albedo"][mat]
    l = lpos[None, :, :] - p[:, None, :]
    ln = torch.sqrt(torch.sum(l * l, dim=-1, keepdim=True))
    l = l / torch.where(ln > 0, ln, torch.ones_like(ln))
    s = torch.sum(nrm[:, None, :] * l, dim=-1)                               # (N,L)
    im = torch.sum(s[:, :, None] * lcol[None, :, :] * alb[:, None, :], dim=1)
    im = torch.where(hit[:, None], im, torch.zeros_like(im))
    im = torch.where(im < 0, torch.zeros_like(im), im)
    if "tonemap" in scene:
        g = float(np.ravel(scene["tonemap"]["gamma"])[0])
        im = torch.where(im > 0, im.clamp_min(1e-300) ** g, torch.zeros_like(im) if g > 0 else torch.ones_like(im))
    depth = torch.where(hit, t, torch.full_like(t, float("inf")))
    return im.reshape(H, W, 3), depth.reshape(H, W), hit.reshape(H, W)


def gradients(scene: Dict[str, Any], grad_image: np.ndarray, grad_depth: Optional[np.ndarray] = None,
              ref: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
    """d(sum(image*grad_image) + sum(depth*grad_depth over hit pixels)) / d(each leaf), as fp64 ndarrays."""
    leaves = make_leaves(scene)
    image, depth, hit = render(scene, leaves, ref)
    loss = torch.sum(image * torch.as_tensor(grad_image))
    if grad_depth is not None:
        gd = torch.as_tensor(grad_depth)
        loss = loss + torch.sum(torch.where(hit, depth * gd, torch.zeros_like(gd)))
    loss.backward()
    return {k: (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in leaves.items()}


# ------------------------------------------------------------------------------------------------------------------
# torch-backend semantics (render(scene, shading='torch')): fp64 torch restatement of np_oracle_tch.render
# ------------------------------------------------------------------------------------------------------------------
TCH_EXTRA_LEAVES = ("lights.attenuation", "lights.ambient", "materials.coeffs")


def make_leaves_tch(scene: Dict[str, Any], requires_grad: bool = True) -> Dict[str, torch.Tensor]:
    leaves = make_leaves(scene, requires_grad)
    leaves["lights.attenuation"] = torch.tensor(np.asarray(scene["lights"]["attenuation"], dtype=np.float64),
                                                requires_grad=requires_grad)
    leaves["lights.ambient"] = torch.tensor(np.asarray(scene["lights"]["ambient"], dtype=np.float64),
                                            requires_grad=requires_grad)
    leaves["materials.coeffs"] = torch.tensor(np.asarray(scene["materials"]["coeffs"], dtype=np.float64),
                                              requires_grad=requires_grad)
    return leaves


def _unit3_eps(v: torch.Tensor) -> torch.Tensor:
    """torch/utils.py:131-135: v / sqrt(sum(v^2 + 1e-10)) over xyz."""
    return v / torch.sqrt(torch.sum(v * v + 1e-10, dim=-1, keepdim=True))


def render_tch(scene: Dict[str, Any], leaves: Dict[str, torch.Tensor], ref: Optional[Dict[str, np.ndarray]] = None,
               double_sided: bool = False, use_quartic: bool = False, visibility: Optional[np.ndarray] = None):
    """Differentiable image (H,W,3), depth (H,W) and hit mask under the torch backend's semantics
    (diffrend/torch/renderer.py:82-125,136-355; see oracle/np_oracle_tch.py for the forward restatement and its two
    documented deviations).  Selection and every mask (per-light relu, double_sided sign, clip) are piecewise
    constant, as under the reference's autograd."""
    from . import np_oracle_tch
    cam = scene["camera"]
    if np_oracle_tch.is_ortho(cam):
        # torch/utils.py:461-468: every ray has its own origin on the image plane and the one direction at - eye
        eye_np, orig_np, dvec, H, W = np_oracle_tch.generate_rays_ortho(cam)
        orig = torch.tensor(np.ascontiguousarray(orig_np))                  # (N,3)
        d = torch.tensor(np.broadcast_to(dvec[None, :], orig_np.shape).copy())
    else:
        eye_np, ray_np, H, W = np_oracle_tch.generate_rays(cam)
        d = torch.tensor(ray_np.T.copy())                                   # (N,3), unit
        orig = torch.tensor(eye_np[:3])[None, :].expand(d.shape[0], 3)
    if ref is None:
        ref = np_oracle_tch.render(scene, double_sided=double_sided, use_quartic=use_quartic)
    nearest = np.asarray(ref["nearest"]).reshape(-1)
    hit_np = np.asarray(ref["depth"]).reshape(-1) <= cam["far"]
    npix = H * W
    eye = torch.tensor(eye_np[:3])

    t = torch.zeros(npix, dtype=torch.float64)
    nrm = torch.zeros((npix, 3), dtype=torch.float64)
    mat = np.zeros(npix, dtype=np.int64)
    start = 0
    for kind, grp in scene["objects"].items():
        count = (grp["face"] if kind == "triangle" else grp["pos"]).shape[0]
        sel = np.nonzero(hit_np & (nearest >= start) & (nearest < start + count))[0]
        if sel.size:
            loc = torch.as_tensor(nearest[sel] - start)
            ds = d[sel]
            mat[sel] = np.asarray(grp["material_idx"])[nearest[sel] - start]
            if kind == "sphere":
                c = leaves["sphere.pos"][loc][:, :3]
                r = leaves["sphere.radius"][loc]
                oc = orig[sel] - c
                a = torch.sum(ds * ds, dim=-1)
                b = 2 * torch.sum(oc * ds, dim=-1)
                cc = torch.sum(oc * oc, dim=-1) - r * r
                root = torch.sqrt(torch.clamp_min(b * b - 4 * a * cc, 0.0))
                t1 = (-b - root) / (2 * a)
                t2 = (-b + root) / (2 * a)
                ts = torch.where(t1 >= 0, t1, t2)                             # the smaller non-negative root
                p = orig[sel] + ts[:, None] * ds
                n = _unit3_eps(p - c)
            else:
                q = (leaves["triangle.face"][loc][:, 0, :3] if kind == "triangle" else leaves[f"{kind}.pos"][loc][:, :3])
                n = _unit3_eps(leaves[f"{kind}.normal"][loc][:, :3])
                ts = torch.sum(n * (q - orig[sel]), dim=-1) / torch.sum(n * ds, dim=-1)
            t = t.index_put((torch.as_tensor(sel),), ts)
            nrm = nrm.index_put((torch.as_tensor(sel),), n)
        start += count

    hit = torch.as_tensor(hit_np)
    p = orig + t[:, None] * d
    lpos = leaves["lights.pos"][:, :3]
    lcol = leaves["colors"][np.asarray(scene["lights"]["color_idx"])]
    att = leaves["lights.attenuation"]
    amb = leaves["lights.ambient"]
    alb = leaves["materials.albedo"][mat]
    cf = leaves["materials.coeffs"][mat]
    ldir = lpos[None, :, :] - p[:, None, :]                                  # (N,L,3)
    lnorm = torch.sqrt(torch.sum(ldir * ldir, dim=-1, keepdim=True))
    ldir = ldir / torch.where(lnorm > 0, lnorm, torch.ones_like(lnorm))
    powv = 4 if use_quartic else 2
    den = att[None, :, 0:1] + lnorm * att[None, :, 1:2] + (lnorm ** powv) * att[None, :, 2:3]
    afac = 1.0 / torch.where(den.abs() > 0, den, torch.ones_like(den))
    ldn = torch.sum(nrm[:, None, :] * ldir, dim=-1)                          # (N,L)
    ndotl = afac[..., 0] * ldn
    cdir = _unit3_eps(eye[None, :] - p)                                      # (N,3)
    cdotn = torch.sum(cdir * nrm, dim=-1)
    rdotc = 2.0 * ldn * cdotn[:, None] - torch.sum(cdir[:, None, :] * ldir, dim=-1)
    if double_sided:
        sgn = torch.sign(cdotn).detach()[:, None]
        ndotl = sgn * ndotl
        rdotc = sgn * rdotc
    ndotl = torch.relu(ndotl)
    rdotc = torch.relu(rdotc)
    spec = cf[:, None, 1] * rdotc ** cf[:, None, 2]                          # torch.pow: 0 ** 0 = 1, masked gradients at 0
    w = cf[:, None, 0] * ndotl + spec                                        # (N,L)
    if visibility is not None:                                               # (L,N) constants: shadow rays
        w = w * torch.as_tensor(np.asarray(visibility, dtype=np.float64).reshape(w.shape[1], -1).T)
    col = w[:, :, None] * (lcol[None, :, :] * alb[:, None, :]) + amb[None, None, :] * alb[:, None, :]
    im = torch.sum(col, dim=1)
    im = torch.where(hit[:, None], im, torch.zeros_like(im))
    im = torch.relu(im)
    if "tonemap" in scene:
        g = float(np.ravel(scene["tonemap"]["gamma"])[0])
        im = torch.where(im > 0, im.clamp_min(1e-300) ** g, torch.zeros_like(im) if g > 0 else torch.ones_like(im))
    depth = torch.where(hit, t, torch.full_like(t, float(cam["far"]) + 1.0))
    return im.reshape(H, W, 3), depth.reshape(H, W), hit.reshape(H, W)


def gradients_tch(scene: Dict[str, Any], grad_image: np.ndarray, grad_depth: Optional[np.ndarray] = None,
                  ref: Optional[Dict[str, np.ndarray]] = None, double_sided: bool = False,
                  use_quartic: bool = False, visibility: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    leaves = make_leaves_tch(scene)
    image, depth, hit = render_tch(scene, leaves, ref, double_sided, use_quartic, visibility)
    loss = torch.sum(image * torch.as_tensor(grad_image))
    if grad_depth is not None:
        gd = torch.as_tensor(grad_depth)
        loss = loss + torch.sum(torch.where(hit, depth * gd, torch.zeros_like(gd)))
    loss.backward()
    return {k: (v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))) for k, v in leaves.items()}
