"""CPU oracle for the reference's *torch* backend semantics -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp64 numpy restatement of the forward of ``diffrend/torch/renderer.py:136-355`` (perspective and orthographic
projection; shadows, see light_visibility), the superset shading model that the hip backend offers as ``render(scene, shading='torch')``
(SURVEY.md section 8, row f1).  Differences from the numpy backend that this file reproduces:
  * camera basis orthonormalised: x = unit(cross(unit(up), z)), y = cross(z, x)        (torch/utils.py:402-427)
  * normals normalised over xyz only, with the reference's eps: u / sqrt(sum(u^2 + 1e-10))   (torch/utils.py:131-135)
  * background depth far + 1                                                            (torch/renderer.py:180-183)
  * Phong fragment shader: attenuation, per-light relu, specular, ambient (added once PER LIGHT, as the reference
    does), double_sided, use_quartic                                                    (torch/renderer.py:82-125)
Deliberate deviations (documented in DESIGN.md): a missed primitive is a miss (the reference marks misses with the
literal distance 1001, which becomes a hit when far >= 1001, torch/utils.py:323,363), and a sphere root that is
negative is a miss (the reference substitutes max(t)+1 over the current pixel tile, torch/utils.py:266-268).

Parity status: PINNED with an fp32 tolerance -- the reference computes this path in float32;
``oracle/gen_golden_tch.py`` ran it unmodified and ``tests/test_oracle_tch.py`` compares.
"""
from __future__ import annotations

import numpy as np

from . import np_oracle


def unit3(u):
    """torch/utils.py:131-135: u / sqrt(sum(u^2 + 1e-10)), divisor 1 where it is 0."""
    u = np.asarray(u, dtype=np.float64)
    den = np.sqrt(np.sum(u ** 2 + 1e-10, axis=-1, keepdims=True))
    return u / np.where(np.abs(den) > 0, den, 1.0)


def cam_vec(v):
    """A camera vector as the torch backend holds it: a float32 tensor (make_torch_var, torch/render.py:81-100), whatever
    container the scene used -- its values, in float64."""
    return np.asarray(v, dtype=np.float64).astype(np.float32).astype(np.float64)


def is_ortho(camera):
    return str(camera.get('proj_type', 'perspective')) in ('ortho', 'orthographic')


def generate_rays_ortho(camera):
    """torch/utils.py:439-468 (orthographic branch): every ray has the direction normalize(at - eye) and its own
    origin eye + x*X + y*Y on the image plane through the eye (lookat_inv, :385-400)."""
    vp = camera['viewport']
    W, H = vp[2] - vp[0], vp[3] - vp[1]
    h = np.tan(camera['fovy'] / 2) * 2 * camera['focal_length']
    w = h * (float(W) / float(H))
    x, y = np.meshgrid(np.linspace(-1, 1, W), np.linspace(1, -1, H))
    x = x.ravel() * (w / 2)
    y = y.ravel() * (h / 2)
    eye = cam_vec(camera['eye'])[:3]
    at = cam_vec(camera['at'])[:3]
    up = cam_vec(camera['up'])[:3]
    z = unit3(eye - at)
    xb = unit3(np.cross(unit3(up), z))
    yb = np.cross(z, xb)
    orig = eye[None, :] + x[:, None] * xb[None, :] + y[:, None] * yb[None, :]      # (N,3)
    return eye, orig, unit3(at - eye), H, W


def _hits_general(segs, total, orig, dirs):
    """Ray distances (M,n) of rays with their own origins `orig` (n,3) and directions `dirs` (n,3) against every
    primitive, torch semantics (torch/utils.py:238-366) with this oracle's two deviations: a miss is +inf, a
    negative sphere root is a miss."""
    n = orig.shape[0]
    t_all = np.empty((total, n))
    for kind, start, count, g in segs:
        if kind == 'sphere':
            oc = orig[None, :, :] - g['pos'][:, None, :3]                          # (M,n,3)
            a = np.sum(dirs ** 2, axis=-1)[None, :]
            b = 2 * np.sum(oc * dirs[None, :, :], axis=-1)
            c = np.sum(oc ** 2, axis=-1) - (g['radius'] ** 2)[:, None]
            disc = b ** 2 - 4 * a * c
            ok = disc >= 0
            root = np.sqrt(np.where(ok, disc, 0.0))
            t1 = (-b - root) / (2 * a)
            t2 = (-b + root) / (2 * a)
            t = np.minimum(np.where(ok & (t1 >= 0), t1, np.inf), np.where(ok & (t2 >= 0), t2, np.inf))
        else:
            nrm = unit3(g['normal'][:, :3])
            q = g['face'][:, 0, :3] if kind == 'triangle' else g['pos'][:, :3]
            den = nrm @ dirs.T                                                     # (M,n)
            t = (np.sum(q * nrm, axis=1)[:, None] - nrm @ orig.T) / den
            p = orig[None, :, :] + t[:, :, None] * dirs[None, :, :]
            if kind == 'disk':
                inside = np.sum((p - g['pos'][:, None, :3]) ** 2, axis=-1) <= (g['radius'] ** 2)[:, None]
                t = np.where(inside, t, np.inf)
            elif kind == 'triangle':
                inside = np.ones_like(t, dtype=bool)
                for i in range(3):
                    edge = (g['face'][:, (i + 1) % 3, :3] - g['face'][:, i, :3])[:, None, :]
                    rel = p - g['face'][:, i, :3][:, None, :]
                    inside &= np.sum(np.cross(edge, rel) * nrm[:, None, :], axis=-1) >= 0
                t = np.where(inside, t, np.inf)
        t_all[start:start + count] = t
    return t_all


def light_visibility(scene, res):
    """torch/renderer.py:291-314 (`shadow=True`): from every fragment a ray towards every light, started 0.1 along
    it; the light is visible unless some primitive OTHER than the fragment's own is hit closer than the light
    (distances measured from the shifted origin, the light's distance from the fragment itself, as the reference
    does).  Returns (L, H*W) bool.  PINNED by tests/golden/s1*.npz: outputs of the reference's render(shadow=True),
    generated on the CPU by oracle/gen_golden_shadow.py, which aliases `torch.cuda.FloatTensor` to `torch.FloatTensor`
    in its own process because the reference casts the mask with `.type(torch.cuda.FloatTensor)` (:311); the
    reference returns only the shaded image, which is what tests/test_oracle_tch.py compares."""
    cam = scene['camera']
    objs = {k: {f: np.asarray(v, dtype=np.float64) if f != 'material_idx' else np.asarray(v) for f, v in g.items()}
            for k, g in scene['objects'].items()}
    segs, total = np_oracle._segments(objs)
    p = res['pos'].reshape(-1, 3)
    nearest = res['nearest'].reshape(-1)
    lpos = np.asarray(scene['lights']['pos'], dtype=np.float64)[:, :3]
    vis = np.ones((lpos.shape[0], p.shape[0]), dtype=bool)
    with np.errstate(all='ignore'):
        for l in range(lpos.shape[0]):
            v = lpos[l][None, :] - p
            dist = np.sqrt(np.sum(v ** 2, axis=-1))
            dirs = v / dist[:, None]
            t = _hits_general(segs, total, p + 0.1 * dirs, dirs)
            t = np.where((t > 0) & (t < dist[None, :]), t, np.inf)
            blocker = np.argmin(t, axis=0)
            vis[l] = ~np.isfinite(t[blocker, np.arange(p.shape[0])]) | (blocker == nearest)
    return vis


def shade(scene, res, vis=None, double_sided=False, use_quartic=False):
    """The image (H,W,3) of a frame whose geometry buffers (`pos`, `normal`, `nearest`, `depth`) are in `res`, with
    optional light visibility (L, H*W)."""
    cam = scene['camera']
    H, W = res['depth'].shape
    hit = (res['depth'] <= cam['far']).reshape(-1)
    material_idx = np.concatenate([np.asarray(g['material_idx']) for g in scene['objects'].values()]).astype(np.int64)
    eye = cam_vec(cam['eye'])[:3]
    with np.errstate(all='ignore'):
        im = _fragment_shader(scene, eye, res['pos'].reshape(-1, 3), res['normal'].reshape(-1, 3),
                              material_idx[res['nearest'].reshape(-1)], double_sided, use_quartic, vis)
        im = np.maximum(np.where(hit[:, None], im, 0.0), 0.0)
        if 'tonemap' in scene:
            im = im ** float(np.ravel(scene['tonemap']['gamma'])[0])
    return im.reshape(H, W, 3)


def _render_ortho(scene, double_sided, use_quartic):
    """The orthographic frame: same pipeline as `render` with per-ray origins and one direction."""
    cam = scene['camera']
    eye, orig, dvec, H, W = generate_rays_ortho(cam)
    npix = H * W
    near, far = cam['near'], cam['far']
    objs = {k: {f: np.asarray(v, dtype=np.float64) if f != 'material_idx' else np.asarray(v) for f, v in g.items()}
            for k, g in scene['objects'].items()}
    segs, total = np_oracle._segments(objs)
    material_idx = np.concatenate([g['material_idx'] for _, _, _, g in segs], axis=0).astype(np.int64)
    with np.errstate(all='ignore'):
        t_all = _hits_general(segs, total, orig, np.broadcast_to(dvec[None, :], orig.shape))
        valid = (near <= t_all) & (t_all <= far)
        t_all[~valid] = np.inf
        win = np.argmin(t_all, axis=0)
        z = t_all[win, np.arange(npix)]
        hit = np.isfinite(z)
        p = orig + np.where(hit, z, 0.0)[:, None] * dvec[None, :]
        fn = np.zeros((npix, 3))
        for kind, start, count, g in segs:
            sel = hit & (win >= start) & (win < start + count)
            if np.any(sel):
                loc = win[sel] - start
                fn[sel] = unit3(p[sel] - g['pos'][loc, :3]) if kind == 'sphere' else unit3(g['normal'][:, :3])[loc]
        im = _fragment_shader(scene, eye, p, fn, material_idx[win], double_sided, use_quartic)
        im = np.maximum(np.where(hit[:, None], im, 0.0), 0.0)
        if 'tonemap' in scene:
            im = im ** float(np.ravel(scene['tonemap']['gamma'])[0])
    return {'image': im.reshape(H, W, 3), 'depth': np.where(hit, z, far + 1.0).reshape(H, W),
            'nearest': np.where(hit, win, 0).reshape(H, W), 'normal': np.where(hit[:, None], fn, 0.0).reshape(H, W, 3),
            'pos': np.where(hit[:, None], p, 0.0).reshape(H, W, 3)}


def _fragment_shader(scene, eye, p, fn, mat, double_sided, use_quartic, vis=None):
    """torch/renderer.py:82-125 for fragments p (n,3) with normals fn (n,3) and material indices mat (n,);
    `vis` (L,n) multiplies the light-times-albedo term (:116-118), not the ambient one."""
    lpos = np.asarray(scene['lights']['pos'], dtype=np.float64)[:, :3]
    lcol = np.asarray(scene['colors'], dtype=np.float64)[np.asarray(scene['lights']['color_idx'])]
    att = np.asarray(scene['lights']['attenuation'], dtype=np.float64)
    ambient = np.asarray(scene['lights']['ambient'], dtype=np.float64)
    alb = np.asarray(scene['materials']['albedo'], dtype=np.float64)[mat]
    cf = np.asarray(scene['materials']['coeffs'], dtype=np.float64)[mat]
    ldir = lpos[:, None, :] - p[None, :, :]
    lnorm = np.sqrt(np.sum(ldir ** 2, axis=-1))[..., None]
    ldir = ldir / np.where(np.abs(lnorm) > 0, lnorm, 1.0)
    powv = 4 if use_quartic else 2
    den = att[:, 0][:, None, None] + lnorm * att[:, 1][:, None, None] + (lnorm ** powv) * att[:, 2][:, None, None]
    afac = 1.0 / np.where(np.abs(den) > 0, den, 1.0)
    ndotl = np.sum(fn[None, :, :] * (afac * ldir), axis=-1)
    refl = -2 * np.sum(-ldir * fn[None, :, :], axis=-1)[..., None] * fn[None, :, :] - ldir
    cdir = unit3(eye[None, None, :] - p[None, :, :])
    rdotc = np.sum(cdir * refl, axis=-1)
    if double_sided:
        sgn = np.sign(np.sum(cdir * fn[None, :, :], axis=-1))
        ndotl = sgn * ndotl
        rdotc = sgn * rdotc
    ndotl = np.maximum(ndotl, 0.0)
    rdotc = np.maximum(rdotc, 0.0)
    lav = lcol[:, None, :] * alb[None, :, :]
    if vis is not None:
        lav = lav * np.asarray(vis, dtype=np.float64)[:, :, None]
    col = (cf[:, 0][None, :, None] * ndotl[:, :, None] +
           cf[:, 1][None, :, None] * (rdotc[:, :, None] ** cf[:, 2][None, :, None])) * lav + \
        ambient[None, None, :] * alb[None, :, :]
    return np.sum(col, axis=0)


def generate_rays(camera):
    """torch/utils.py:439-478 (perspective branch) with lookat_rot_inv (:402-427)."""
    vp = camera['viewport']
    W, H = vp[2] - vp[0], vp[3] - vp[1]
    h = np.tan(camera['fovy'] / 2) * 2 * camera['focal_length']
    w = h * (float(W) / float(H))
    x, y = np.meshgrid(np.linspace(-1, 1, W), np.linspace(1, -1, H))
    x = x.ravel() * (w / 2)
    y = y.ravel() * (h / 2)
    eye = cam_vec(camera['eye'])[:3]
    at = cam_vec(camera['at'])[:3]
    up = cam_vec(camera['up'])[:3]
    z = unit3(eye - at)
    xb = unit3(np.cross(unit3(up), z))
    yb = np.cross(z, xb)
    rot = np.stack((xb, yb, z), axis=-1)
    d = rot @ np.stack((x, y, -np.ones(x.size) * camera['focal_length']), axis=0)
    d /= np.sqrt(np.sum(d ** 2, axis=0))
    return eye, d, H, W


def render(scene, double_sided=False, use_quartic=False, tile=2048, shadow=False):
    """Returns image (H,W,3), depth (H,W) with far+1 background, nearest (H,W), normal (H,W,3), pos (H,W,3)
    (normal / pos are 0 where nothing is hit).  With `shadow` also `visibility` (L,H,W) bool, and the image is
    shaded with it (see light_visibility)."""
    if shadow:
        res = render(scene, double_sided, use_quartic, tile)
        vis = light_visibility(scene, res)
        res['image'] = shade(scene, res, vis, double_sided, use_quartic)
        res['visibility'] = vis.reshape((-1,) + res['depth'].shape)
        return res
    cam = scene['camera']
    if is_ortho(cam):
        return _render_ortho(scene, double_sided, use_quartic)
    eye, ray_dir, H, W = generate_rays(cam)
    npix = H * W
    near, far = cam['near'], cam['far']
    eye4 = np.append(eye, 1.0)

    objs = {k: {f: np.asarray(v, dtype=np.float64) if f != 'material_idx' else np.asarray(v) for f, v in g.items()}
            for k, g in scene['objects'].items()}
    segs, total = np_oracle._segments(objs)
    material_idx = np.concatenate([g['material_idx'] for _, _, _, g in segs], axis=0).astype(np.int64)
    # unit normals per primitive (3-D, eps form); spheres handled per hit
    lpos = np.asarray(scene['lights']['pos'], dtype=np.float64)[:, :3]
    lcol = np.asarray(scene['colors'], dtype=np.float64)[np.asarray(scene['lights']['color_idx'])]
    att = np.asarray(scene['lights']['attenuation'], dtype=np.float64)
    ambient = np.asarray(scene['lights']['ambient'], dtype=np.float64)
    albedo = np.asarray(scene['materials']['albedo'], dtype=np.float64)
    coeffs = np.asarray(scene['materials']['coeffs'], dtype=np.float64)

    image = np.zeros((npix, 3))
    depth = np.full(npix, far + 1.0)
    nearest = np.zeros(npix, dtype=np.int64)
    normal_out = np.zeros((npix, 3))
    pos_out = np.zeros((npix, 3))
    with np.errstate(all='ignore'):
        for s in range(0, npix, tile):
            d3 = ray_dir[:, s:s + tile]
            d4 = np.concatenate([d3, np.zeros((1, d3.shape[1]))], axis=0)
            n = d3.shape[1]
            t_all = np.empty((total, n))
            for kind, start, count, g in segs:
                nrm4 = None
                if kind != 'sphere':
                    nrm4 = np.concatenate([unit3(g['normal'][:, :3]), np.zeros((count, 1))], axis=1)
                if kind == 'disk':
                    t = np_oracle.hit_disk(eye4, d4, g['pos'], nrm4, g['radius'])
                elif kind == 'plane':
                    t = np_oracle.hit_plane(eye4, d4, g['pos'], nrm4)
                elif kind == 'triangle':
                    t = np_oracle.hit_triangle(eye4, d4, g['face'], nrm4)
                else:
                    # torch/utils.py:238-279 with bad roots treated as misses
                    oc = eye[None, :] - g['pos'][:, :3]
                    a = np.sum(d3 ** 2, axis=0)
                    b = 2 * (oc @ d3)
                    c = (np.sum(oc ** 2, axis=1) - g['radius'] ** 2)[:, None]
                    disc = b ** 2 - 4 * a * c
                    ok = disc >= 0
                    root = np.sqrt(np.where(ok, disc, 0.0))
                    t1 = (-b - root) / (2 * a)
                    t2 = (-b + root) / (2 * a)
                    t1 = np.where(ok & (t1 >= 0), t1, np.inf)
                    t2 = np.where(ok & (t2 >= 0), t2, np.inf)
                    t = np.minimum(t1, t2)
                t_all[start:start + count] = t
            valid = (near <= t_all) & (t_all <= far)
            t_all[~valid] = np.inf
            win = np.argmin(t_all, axis=0)
            z = t_all[win, np.arange(n)]
            hit = np.isfinite(z)
            p = eye[None, :] + np.where(hit, z, 0.0)[:, None] * d3.T
            fn = np.zeros((n, 3))
            for kind, start, count, g in segs:
                sel = hit & (win >= start) & (win < start + count)
                if not np.any(sel):
                    continue
                loc = win[sel] - start
                if kind == 'sphere':
                    fn[sel] = unit3(p[sel] - g['pos'][loc, :3])
                else:
                    fn[sel] = unit3(g['normal'][:, :3])[loc]
            alb = albedo[material_idx[win]]
            cf = coeffs[material_idx[win]]
            # fragment_shader, torch/renderer.py:82-125
            ldir = lpos[:, None, :] - p[None, :, :]                                # (L,n,3)
            lnorm = np.sqrt(np.sum(ldir ** 2, axis=-1))[..., None]
            ldir = ldir / np.where(np.abs(lnorm) > 0, lnorm, 1.0)
            powv = 4 if use_quartic else 2
            den = att[:, 0][:, None, None] + lnorm * att[:, 1][:, None, None] + (lnorm ** powv) * att[:, 2][:, None, None]
            afac = 1.0 / np.where(np.abs(den) > 0, den, 1.0)
            ndotl = np.sum(fn[None, :, :] * (afac * ldir), axis=-1)
            refl = -2 * np.sum(-ldir * fn[None, :, :], axis=-1)[..., None] * fn[None, :, :] - ldir
            cdir = unit3(eye[None, None, :] - p[None, :, :])
            rdotc = np.sum(cdir * refl, axis=-1)
            if double_sided:
                sgn = np.sign(np.sum(cdir * fn[None, :, :], axis=-1))
                ndotl = sgn * ndotl
                rdotc = sgn * rdotc
            ndotl = np.maximum(ndotl, 0.0)
            rdotc = np.maximum(rdotc, 0.0)
            lav = lcol[:, None, :] * alb[None, :, :]
            col = (cf[:, 0][None, :, None] * ndotl[:, :, None] +
                   cf[:, 1][None, :, None] * (rdotc[:, :, None] ** cf[:, 2][None, :, None])) * lav + \
                ambient[None, None, :] * alb[None, :, :]
            im = np.sum(col, axis=0)
            im = np.where(hit[:, None], im, 0.0)
            im = np.maximum(im, 0.0)
            image[s:s + n] = im
            depth[s:s + n] = np.where(hit, z, far + 1.0)
            nearest[s:s + n] = np.where(hit, win, 0)
            normal_out[s:s + n] = np.where(hit[:, None], fn, 0.0)
            pos_out[s:s + n] = np.where(hit[:, None], p, 0.0)
        if 'tonemap' in scene:
            image = image ** float(np.ravel(scene['tonemap']['gamma'])[0])
    return {'image': image.reshape(H, W, 3), 'depth': depth.reshape(H, W), 'nearest': nearest.reshape(H, W),
            'normal': normal_out.reshape(H, W, 3), 'pos': pos_out.reshape(H, W, 3)}


def norm_depth_image(depth, far):
    """`norm_depth_image_only` (diffrend/torch/renderer.py:245-249): background (depth >= far, i.e. the far + 1 fill)
    takes the minimum depth, then (d - min) / (max - min).  Restated from the source text: the reference's own call
    raises TypeError before reaching these lines (oracle/check_ref_kwargs.py), so this function is NOT pinned by a
    reference output."""
    depth = np.asarray(depth, dtype=np.float64)
    lo = depth.min()
    img = np.where(depth >= far, lo, depth)
    with np.errstate(all="ignore"):
        return (img - lo) / (depth.max() - lo)
