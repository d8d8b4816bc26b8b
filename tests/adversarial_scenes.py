"""Directed adversarial scenes: primitives with coordinates / radii / vertices of 2^30 .. 2^66 (1e9 .. 7e19) that
nevertheless CROSS THE VIEW, so that the fp64 arithmetic of the reference path (numpy/renderer.py:9-130) cancels and its
hit / miss decisions are made at a granularity between a fraction of a pixel and the whole image.

Random fuzz does not find these: a primitive only reaches the view when its huge numbers cancel exactly (a centre at
(3R, 4R, 0) with radius 5R, a triangle edge from (-R, -R) to (R, R), a plane through (4R, 3R, 0) with normal
(3, -4, 0)), which needs numbers that are exact in float32.  The band that matters is in the middle -- at 1e20 every
derived quantity is pure noise and is recognised as such, at 1e6 nothing cancels; in between the reject records of the
accelerated modes (srh_reject.h) have to carry the fp64 rounding of BOTH sides in their margins.

Used by tests/test_hip_adversarial.py (-m gpu) and tools/fuzz_campaign.py (other seeds)."""
import numpy as np

KINDS = ("disk", "sphere", "triangle", "plane")


def f32(a):
    return np.asarray(a, dtype=np.float32)


def _axes(rng):
    """a random permutation of the axes with random signs: (index, sign) triples"""
    p = rng.permutation(3)
    s = rng.choice([-1.0, 1.0], 3)
    return [(int(p[i]), float(s[i])) for i in range(3)]


def _vec(ax, vals):
    """vector with vals[i] placed on axis ax[i] (with its sign)"""
    v = np.zeros(3)
    for (j, sg), x in zip(ax, vals):
        v[j] = sg * x
    return v


def _tiny(rng, R):
    """what may sit in a normal component that 'should' be zero: exactly zero, denormal, tiny, or just enough to tilt the
    plane by O(1) over a distance R"""
    c = rng.randint(5)
    if c == 0:
        return 0.0
    if c == 1:
        return float(rng.choice([-1, 1])) * 3e-39
    if c == 2:
        return float(rng.choice([-1, 1])) * 1e-30
    return float(rng.uniform(-2, 2)) / R


def _ulps(x, j):
    """float32 neighbour j steps away from x"""
    x = np.float32(x)
    for _ in range(abs(j)):
        x = np.nextafter(x, np.float32(np.inf if j > 0 else -np.inf))
    return float(x)


def _camera(rng):
    W, H = [(64, 48), (80, 64), (96, 64), (64, 80)][rng.randint(4)]
    e = rng.normal(size=3)
    e = e / np.linalg.norm(e) * float(rng.choice([1.5, 3.0, 6.0]))
    return {"viewport": [0, 0, W, H], "fovy": float(np.deg2rad(rng.choice([20, 45, 70]))),
            "focal_length": float(rng.choice([0.5, 1.0])), "eye": [*map(float, e), 1.0],
            "at": [*map(float, rng.normal(size=3) * 0.15), 1.0], "up": [*map(float, rng.normal(size=3)), 0.0],
            "near": float(rng.choice([0.01, 0.1, 1.0])), "far": float(rng.choice([50.0, 1000.0, 1e30]))}


def _ordinary(rng, kind, n):
    pos = np.concatenate([rng.uniform(-1.0, 1.0, (n, 3)), np.ones((n, 1))], 1)
    nrm = np.concatenate([rng.normal(size=(n, 3)), np.zeros((n, 1))], 1)
    if kind == "disk":
        return {"pos": pos, "normal": nrm, "radius": np.exp(rng.uniform(np.log(0.05), np.log(0.6), n))}
    if kind == "sphere":
        return {"pos": pos, "radius": np.exp(rng.uniform(np.log(0.05), np.log(0.4), n))}
    if kind == "plane":
        pos[:, :3] *= 2.0
        return {"pos": pos, "normal": nrm}
    c = rng.uniform(-1.0, 1.0, (n, 1, 3))
    v = c + rng.normal(size=(n, 3, 3)) * 0.4
    fn = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]) * rng.choice([-1, 1], (n, 1))
    return {"face": np.concatenate([v, np.ones((n, 3, 1))], 2), "normal": np.concatenate([fn, np.zeros((n, 1))], 1)}


def _huge_disk(rng, R):
    ax = _axes(rng)
    c = rng.randint(5)
    z0 = float(rng.uniform(-0.5, 0.5))
    if c == 0:      # centre R along one axis, normal across it, radius within a few float32 steps of R: rim through the view
        pos = _vec(ax, [R, rng.uniform(-.5, .5), rng.uniform(-.5, .5)])
        nrm = _vec(ax, [_tiny(rng, R), rng.normal(), rng.normal()])
        rad = _ulps(R, int(rng.randint(-2, 3)))
    elif c == 1:    # centre (3R, 4R, z0), radius 5R: the rim passes through the origin exactly
        pos = _vec(ax, [3 * R, 4 * R, z0])
        nrm = _vec(ax, [_tiny(rng, R), _tiny(rng, R), 1.0])
        rad = _ulps(5 * R, int(rng.randint(-2, 3)))
    elif c == 2:    # ordinary centre, huge radius: the disc is its plane
        pos = _vec(ax, [rng.uniform(-.5, .5), rng.uniform(-.5, .5), z0])
        nrm = rng.normal(size=3)
        rad = R
    elif c == 3:    # centre far away on two axes, radius 2R: covers the view, |oc|^2 and r^2 both ~ R^2
        pos = _vec(ax, [R, R * float(rng.choice([1, 0.5, 2])), z0])
        nrm = _vec(ax, [_tiny(rng, R), _tiny(rng, R), 1.0])
        rad = 2 * R * float(rng.choice([1, 2]))
    else:           # a huge component in the NORMAL (normalisation makes it an axis), ordinary disc
        pos = _vec(ax, [rng.uniform(-.5, .5), rng.uniform(-.5, .5), z0])
        nrm = _vec(ax, [rng.normal(), rng.normal(), R])
        rad = float(rng.uniform(0.2, 1.0))
    rad *= float(rng.choice([1, -1]))           # the reference squares the radius
    return {"pos": [[*pos, 1.0]], "normal": [[*nrm, 0.0]], "radius": [rad]}


def _huge_sphere(rng, R):
    ax = _axes(rng)
    c = rng.randint(4)
    if c == 0:      # centre R along an axis, radius ~ R: the surface passes through the view, nearly flat
        pos = _vec(ax, [R, rng.uniform(-.5, .5), rng.uniform(-.5, .5)])
        rad = _ulps(R, int(rng.randint(-2, 3)))
    elif c == 1:    # centre (2R, 3R, 6R), radius 7R: through the origin exactly
        pos = _vec(ax, [2 * R, 3 * R, 6 * R])
        rad = _ulps(7 * R, int(rng.randint(-2, 3)))
    elif c == 2:    # ordinary centre, huge radius: the eye is deep inside
        pos = _vec(ax, [rng.uniform(-.5, .5), rng.uniform(-.5, .5), rng.uniform(-.5, .5)])
        rad = R
    else:           # centre R away, radius 2R
        pos = _vec(ax, [R, R * float(rng.choice([0, 1])), 0.0])
        rad = 2 * R
    return {"pos": [[*pos, 1.0]], "radius": [rad]}


def _huge_triangle(rng, R):
    ax = _axes(rng)
    c = rng.randint(6)
    z0 = float(rng.uniform(-0.5, 0.5))
    s1, s2 = float(rng.choice([0.5, 1, 2])), float(rng.choice([0.5, 1, 2]))
    o = lambda: float(rng.uniform(-.6, .6))                  # noqa: E731
    if c == 0:      # wedge with its apex in the view, the other two vertices far away on two axes
        v = [_vec(ax, [o(), o(), z0]), _vec(ax, [R, R * s1, z0]), _vec(ax, [-R * s2, R, z0])]
    elif c == 1:    # an edge from (-R, -R) to (R, R): it passes through the origin, decided by cancelling products
        v = [_vec(ax, [-R, -R, z0]), _vec(ax, [R, R, z0]), _vec(ax, [R * float(rng.choice([-1, 1])), -R * float(rng.choice([-1, 1])) * s1, z0])]
    elif c == 2:    # the same edge, third vertex ordinary
        v = [_vec(ax, [-R, -R, z0]), _vec(ax, [R, R, z0]), _vec(ax, [o() + 1.0, o() - 1.0, z0])]
    elif c == 3:    # all three far away, the view deep inside
        v = [_vec(ax, [-R, -R * s1, z0]), _vec(ax, [R, -R * s2, z0]), _vec(ax, [0.0, R, z0])]
    elif c == 4:    # one huge coordinate only
        v = [_vec(ax, [o(), o(), z0]), _vec(ax, [o(), R, z0]), _vec(ax, [o() + 1.0, o(), z0])]
    else:           # a slanted plane through the origin: normal (1, -1, 0), in-plane directions (1, 1, 0) and (0, 0, 1)
        v = [_vec(ax, [-R, -R, -R * s1]), _vec(ax, [R, R, -R * s2]), _vec(ax, [o(), o(), R])]
        v[2][ax[1][0]] = ax[1][1] * (ax[0][1] * v[2][ax[0][0]])         # third vertex on the plane x = y as well
    k = int(rng.randint(3))
    v = v[k:] + v[:k]                           # which vertex is vertex 0 (the plane point of the reference)
    if c == 5:
        nrm = _vec(ax, [1.0, -1.0, _tiny(rng, R)])
    else:
        nrm = _vec(ax, [_tiny(rng, R), _tiny(rng, R), 1.0])
    nrm = nrm * float(rng.choice([-1, 1]))      # the reference takes the supplied normal: either orientation
    return {"face": [[[*p, 1.0] for p in v]], "normal": [[*nrm, 0.0]]}


def _huge_plane(rng, R):
    ax = _axes(rng)
    c = rng.randint(3)
    if c == 0:      # through (4R, 3R, z) with normal (3, -4, 0): the offset n.pos cancels to rounding noise
        pos = _vec(ax, [4 * R, 3 * R, rng.uniform(-.5, .5)])
        nrm = _vec(ax, [3.0, -4.0, _tiny(rng, R)])           # _vec puts the same signs on both: n.pos = 12R - 12R
    elif c == 1:    # far away along an axis it contains
        pos = _vec(ax, [R, rng.uniform(-.5, .5), rng.uniform(-.5, .5)])
        nrm = _vec(ax, [_tiny(rng, R), rng.normal(), rng.normal()])
    else:           # a huge component in the normal
        pos = _vec(ax, [rng.uniform(-.5, .5), rng.uniform(-.5, .5), rng.uniform(-.5, .5)])
        nrm = _vec(ax, [rng.normal(), rng.normal(), R])
    return {"pos": [[*pos, 1.0]], "normal": [[*nrm, 0.0]]}


_HUGE = {"disk": _huge_disk, "sphere": _huge_sphere, "triangle": _huge_triangle, "plane": _huge_plane}


def huge_scene(rng, kind):
    """One scene: 1-2 huge primitives of `kind` that cross the view, 0-4 ordinary ones of the same kind around them (in
    random order), and now and then a batch of another kind, so that the winner is decided between them."""
    groups = {}
    kinds = [kind] + ([str(rng.choice([k for k in KINDS if k != kind]))] if rng.randint(3) == 0 else [])
    for kd in rng.permutation(kinds):
        parts = []
        if kd == kind:
            for _ in range(int(rng.randint(1, 3))):
                R = float(2.0 ** int(rng.randint(30, 67)))
                parts.append(_HUGE[kd](rng, R))
        n = int(rng.randint(0 if kd == kind else 1, 5))
        if n:
            parts.append(_ordinary(rng, kd, n))
        g = {k: np.concatenate([np.asarray(p[k], dtype=np.float64) for p in parts], 0) for k in parts[0]}
        order = rng.permutation(len(next(iter(g.values()))))
        g = {k: f32(a[order]) for k, a in g.items()}
        g["material_idx"] = rng.randint(0, 3, len(order))
        groups[str(kd)] = g
    return {"camera": _camera(rng), "lights": {"pos": f32([[3, 4, 5, 1], [-4, 2, 3, 1]]), "color_idx": np.array([1, 2])},
            "colors": f32([[0, 0, 0], [.8, .5, .4], [.3, .6, .9]]),
            "materials": {"albedo": f32([[.5, .5, .5], [.9, .3, .2], [.2, .7, .4]])},
            "objects": groups, "tonemap": {"type": "gamma", "gamma": 0.8}}
