"""CPU oracle for the render(scene) hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of ``diffrend.numpy.renderer.render`` (reference file
``diffrend/numpy/renderer.py``; every function below cites the lines it follows).  It exists so
that the hip backend can be checked on the GPU box, where the reference itself is absent.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; nothing under ``surf_renderer_amd/`` does.

Parity status: PINNED.  ``oracle/gen_golden.py`` ran the unmodified reference in the build
container and committed its outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this file against every one of them (image/depth to 1e-12, ``nearest`` exactly).

Differences from the reference that do not change results:
  * pixels are processed in tiles, so memory is O(M * tile) instead of O(M * N);
  * per-pair normals are not materialised (the reference broadcasts them to (M,N,4), :71); the
    winner's normal is gathered instead -- same values;
  * nothing is printed (the reference prints two matrices per call, :163);
  * the three (M,N) outputs ``ray_dist`` / ``obj_dist`` / ``valid_pixels`` are not returned.
All arithmetic quirks are kept: see SURVEY.md section 8, quirk table Q1-Q17.
"""
from __future__ import annotations

import numpy as np


# ------------------------------------------------------------------------------------------
# The reference's four matrix products (numpy/renderer.py:20,64,69,160) go through np.dot, i.e. the BLAS of the
# machine, whose dgemm kernels accumulate with fused multiply-adds in an order of their own: the last bit of such a
# product is a property of the BLAS build, not of the algorithm.  `render(..., dots="ordered")` evaluates the same
# products as explicit sums, ((a0 b0 + a1 b1) + a2 b2) + a3 b3 with every operation rounded -- the order the hip
# backend documents for its fp64 path (srh_device.h: dot3).  Both variants are checked against the reference's golden
# outputs (tests/test_oracle_golden.py); they differ only where a result is decided by the last bit of a cancelling sum
# (tests/test_hip_adversarial.py compares the GPU with the ordered variant, and the two variants with each other).
# ------------------------------------------------------------------------------------------
_DOTS = "blas"


def _dot(a, b):
    """np.dot(a, b) for a (M,4) or (4,4) and b (4,) or (4,n) -- or the same contraction in a fixed order."""
    if _DOTS == "blas":
        return np.dot(a, b)
    b2 = b if b.ndim == 2 else b[:, np.newaxis]
    out = ((a[:, 0:1] * b2[0:1] + a[:, 1:2] * b2[1:2]) + a[:, 2:3] * b2[2:3]) + a[:, 3:4] * b2[3:4]
    return out if b.ndim == 2 else out[:, 0]


# ------------------------------------------------------------------------------------------
# numpy/ops.py helpers
# ------------------------------------------------------------------------------------------
def nonzero_divide(x, y):
    """numpy/ops.py:18-20 -- divide, but by 1 where the divisor is exactly 0."""
    return x / np.where(np.abs(y) > 0, y, np.ones_like(y))


def normalize(u):
    """numpy/ops.py:23-26 -- unit length over the *whole* last axis (w included, Q8)."""
    u = np.array(u)
    return nonzero_divide(u, np.sqrt(np.sum(np.abs(u) ** 2, axis=-1))[..., np.newaxis])


def lookat_inv(eye, at, up):
    """numpy/ops.py:88-115 -- camera-to-world matrix with the reference's non-orthonormal basis
    (Q1) and its float32 detour for list-typed arguments (Q11)."""
    if type(eye) is list:
        eye = np.array(eye, dtype=np.float32)
    if type(at) is list:
        at = np.array(at, dtype=np.float32)
    if type(up) is list:
        up = np.array(up, dtype=np.float32)
    if up.size == 4:
        assert up[3] == 0
        up = up[:3]
    z = eye - at
    z = (z / np.linalg.norm(z, 2))[:3]
    y = up / np.linalg.norm(up, 2)
    x = np.cross(y, z)
    m = np.eye(4)
    m[:3, :3] = np.stack((x, y, z), axis=1)
    m[:3, 3] = eye[:3] / eye[3]
    return m


# ------------------------------------------------------------------------------------------
# rays
# ------------------------------------------------------------------------------------------
def generate_rays(camera, dtype=np.float64):
    """numpy/renderer.py:145-169.  Returns eye (4,), ray_dir (4,N) row-major over (H,W), H, W.
    Pixel grid = linspace(-1,1,W) x linspace(1,-1,H), samples on the frustum edges (Q12)."""
    vp = camera['viewport']
    W, H = vp[2] - vp[0], vp[3] - vp[1]
    aspect = W / float(H)
    focal = camera['focal_length']
    h = np.tan(camera['fovy'] / 2) * 2 * focal
    w = h * aspect
    x, y = np.meshgrid(np.linspace(-1, 1, W), np.linspace(1, -1, H))
    x *= w / 2
    y *= h / 2
    eye = np.array(camera['eye'])
    d = np.stack((x.ravel(), y.ravel(), -np.ones(x.size) * focal, np.zeros(x.size)), axis=0)
    d = _dot(lookat_inv(eye=eye, at=camera['at'], up=camera['up']), d)
    d /= np.sqrt(np.sum(d ** 2, axis=0))
    return eye.astype(dtype), d.astype(dtype), H, W


# ------------------------------------------------------------------------------------------
# intersections: each returns t (M,n) for a tile of n rays; misses carry the reference's sentinel
# ------------------------------------------------------------------------------------------
def _along(eye, d, t):
    """numpy/renderer.py:5-6 -- eye + t * dir as (M,n,4)."""
    return eye[np.newaxis, np.newaxis, :] + t[..., np.newaxis] * d.T[np.newaxis, ...]


def hit_plane(eye, d, pos, normal):
    """numpy/renderer.py:53-74.  t = (pos.n - n.eye) / (n.d); no masking (denom == 0 gives inf/nan,
    rejected later by the near/far test)."""
    n = normalize(normal)
    dist = np.sum(pos * n, axis=1)
    denom = _dot(n, d)
    return (dist[:, np.newaxis] - _dot(n, eye)[:, np.newaxis]) / denom


def hit_disk(eye, d, pos, normal, radius):
    """numpy/renderer.py:77-93.  Plane hit, kept where |p - c|^2 <= r^2 (4-D norm), else inf."""
    t = hit_plane(eye, d, pos, normal)
    p = _along(eye, d, t)
    dist_sqr = np.sum((p - pos[:, np.newaxis, :]) ** 2, axis=-1)
    t[~(dist_sqr <= radius[:, np.newaxis] ** 2)] = np.inf
    return t


def hit_triangle(eye, d, face, normal):
    """numpy/renderer.py:96-130.  Plane through vertex 0 with the *supplied* normal (Q9); inside
    iff dot(cross(edge_i, p - v_i), n) >= 0 for the three edges; else inf."""
    t = hit_plane(eye, d, face[:, 0, :], normal)
    n3 = normalize(normal)[:, np.newaxis, :3]
    p = _along(eye, d, t)
    inside = None
    for i in range(3):
        edge = (face[:, (i + 1) % 3, :3] - face[:, i, :3])[:, np.newaxis, :]
        rel = (p - face[:, i, :][:, np.newaxis, :])[..., :3]
        cond = np.sum(np.cross(edge, rel) * n3, axis=-1) >= 0
        inside = cond if inside is None else (inside & cond)
    t[~inside] = np.inf
    return t


def hit_sphere(eye, d, pos, radius):
    """numpy/renderer.py:9-50.  Quadratic in t with the reference's sentinels (Q2): a root that is
    negative (or a line that misses) is replaced by the scalar 1.0, and a line that misses gives
    t = 0."""
    oc = eye - pos
    a = np.sum(d ** 2, axis=0)
    b = 2 * _dot(oc, d)
    c = (np.sum(oc ** 2, axis=1) - radius ** 2)[:, np.newaxis]
    disc = b ** 2 - 4 * a * c
    ok = disc >= 0
    root = np.sqrt(np.where(ok, disc, np.zeros_like(disc)))
    inv = 1. / (2 * a)
    t1 = (-b - root) * inv
    t2 = (-b + root) * inv
    one = np.ones_like(np.max(t1) + 1)
    t1 = np.where(ok & (t1 >= 0), t1, one)
    t2 = np.where(ok & (t2 >= 0), t2, one)
    t = np.minimum(t1, t2)
    return np.where(ok, t, np.zeros_like(t))


def _segments(objects):
    """Global primitive numbering: dict order, running offset (numpy/renderer.py:172-201)."""
    segs, start = [], 0
    for kind in objects:
        grp = objects[kind]
        count = grp['face'].shape[0] if kind == 'triangle' else grp['pos'].shape[0]
        segs.append((kind, start, count, grp))
        start += count
    return segs, start


def _winner_normals(segs, nearest, p_win, eye, d, dtype):
    """Normal of the winning primitive per pixel: planar types carry their unit normal
    (numpy/renderer.py:71), spheres (p - c)/|p - c|, zero where the ray's line misses the
    sphere (:45-47; such a pixel can only win when near <= 0)."""
    n = p_win.shape[0]
    out = np.zeros((n, 4), dtype=dtype)
    for kind, start, count, grp in segs:
        sel = (nearest >= start) & (nearest < start + count)
        if not np.any(sel):
            continue
        local = nearest[sel] - start
        if kind == 'sphere':
            c = grp['pos'][local]
            v = p_win[sel] - c
            v = v / np.sqrt(np.sum(v ** 2, axis=-1))[..., np.newaxis]
            oc, ds = eye[np.newaxis, :] - c, d[:, sel].T
            b = 2 * np.sum(oc * ds, axis=-1)
            disc = b ** 2 - 4 * np.sum(ds ** 2, axis=-1) * (np.sum(oc ** 2, axis=-1) - grp['radius'][local] ** 2)
            v[~(disc >= 0)] = 0
            out[sel] = v
        else:
            out[sel] = normalize(grp['normal'])[local]
    return out


def render(scene, tile=2048, dtype=np.float64, rows=None, window=None, dots="blas"):
    """See _render; ``dots`` = "blas" (np.dot, as the reference) or "ordered" (explicit sums, see the top of the file)."""
    global _DOTS
    assert dots in ("blas", "ordered")
    prev, _DOTS = _DOTS, dots
    try:
        return _render(scene, tile, dtype, rows, window)
    finally:
        _DOTS = prev


def _render(scene, tile=2048, dtype=np.float64, rows=None, window=None):
    """Restatement of numpy/renderer.py:204-272 for the ndarray-leaf scene dict the reference
    consumes.  ``rows=(r0, r1)`` renders only image rows [r0, r1) of the full camera (outputs then
    have r1 - r0 rows); ``window=(p0, p1)`` renders only the flat pixel range [p0, p1) of the row-major image
    (outputs then come back flat, used for bounded CPU timing samples); ``dtype=np.float32`` is a diagnostic mode that repeats the same formulas in
    single precision.  Returns image (h,W,3), depth (h,W), nearest (h,W) int64, ray_dir (4,n)."""
    cam = scene['camera']
    eye, ray_dir, H, W = generate_rays(cam, dtype)
    r0, r1 = (0, H) if rows is None else rows
    ray_dir = ray_dir[:, r0 * W:r1 * W] if window is None else ray_dir[:, window[0]:window[1]]
    npix = ray_dir.shape[1]
    near, far = cam['near'], cam['far']

    objs = {}
    for kind, grp in scene['objects'].items():
        objs[kind] = {k: (np.asarray(v).astype(dtype) if k != 'material_idx' else np.asarray(v))
                      for k, v in grp.items()}
    segs, total = _segments(objs)
    material_idx = np.concatenate([g['material_idx'] for _, _, _, g in segs], axis=0)

    light_pos = np.asarray(scene['lights']['pos']).astype(dtype)
    light_colors = np.asarray(scene['colors']).astype(dtype)[scene['lights']['color_idx']]
    albedo = np.asarray(scene['materials']['albedo']).astype(dtype)

    image = np.zeros((npix, 3), dtype=dtype)
    depth = np.zeros(npix, dtype=dtype)
    nearest = np.zeros(npix, dtype=np.int64)

    with np.errstate(all='ignore'):
        for s in range(0, npix, tile):
            d = ray_dir[:, s:s + tile]
            n = d.shape[1]
            t_all = np.empty((total, n), dtype=dtype)
            for kind, start, count, g in segs:
                if kind == 'disk':
                    t = hit_disk(eye, d, g['pos'], g['normal'], g['radius'])
                elif kind == 'plane':
                    t = hit_plane(eye, d, g['pos'], g['normal'])
                elif kind == 'triangle':
                    t = hit_triangle(eye, d, g['face'], g['normal'])
                elif kind == 'sphere':
                    t = hit_sphere(eye, d, g['pos'], g['radius'])
                else:
                    raise KeyError(kind)
                t_all[start:start + count] = t
            # :219-228 near/far on Euclidean ray distance (Q15); argmin keeps the lowest index on
            # ties and gives 0 for an all-miss column (Q6); background depth is inf (Q3)
            valid = (near <= t_all) & (t_all <= far)
            t_all[~valid] = np.inf
            win = np.argmin(t_all, axis=0)
            cols = np.arange(n)
            z = t_all[win, cols]
            # :243-245 fragments.  The reference gathers p from the per-pair array; for a hit
            # pixel that is eye + t * d.  For an all-miss pixel it is garbage that :256 zeroes.
            p = eye[np.newaxis, :] + z[:, np.newaxis] * d.T
            frag_n = _winner_normals(segs, win, p, eye, d, dtype)
            frag_albedo = albedo[material_idx[win]]
            # :248-255 Lambert over all lights; |l| <= 0 -> 1 (Q7); no per-light clamp (Q4)
            l = light_pos[np.newaxis, :] - p[:, np.newaxis, :]
            l_norm = np.sqrt(np.sum(l ** 2, axis=-1))[..., np.newaxis]
            l_norm[l_norm <= 0] = 1
            l = nonzero_divide(l, l_norm)
            col = np.sum(frag_n[:, np.newaxis, :] * l, axis=-1)[..., np.newaxis] * \
                light_colors[np.newaxis, ...] * frag_albedo[:, np.newaxis, :]
            im = np.sum(col, axis=1)
            im[(z < near) | (z > far)] = 0          # :256
            im[im < 0] = 0                          # :259 clip after the light sum
            image[s:s + n] = im
            depth[s:s + n] = z
            nearest[s:s + n] = win
        if 'tonemap' in scene:                       # :262-263, :140-142
            tm = scene['tonemap']
            if tm['type'] == 'gamma':
                image = image ** np.asarray(tm['gamma']).astype(dtype).ravel()[0]

    if window is not None:
        return {'image': image, 'depth': depth, 'nearest': nearest, 'ray_dir': ray_dir}
    h = r1 - r0
    return {'image': image.reshape(h, W, 3), 'depth': depth.reshape(h, W),
            'nearest': nearest.reshape(h, W), 'ray_dir': ray_dir}
