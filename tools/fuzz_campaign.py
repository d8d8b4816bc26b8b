"""One-off parity campaign on a GPU box: random mixed scenes beyond the seeds the test suite fixes.

  python tools/fuzz_campaign.py --seed 1 --seconds 240 [--big]

Per scene: the binned mode (both launch shapes), the row-slab path and the torch-shading variants against the all-pairs
fp64 mode of the same library, bit for bit (`nearest`, `depth`, `image`), plus -- for small scenes -- the numpy oracle at
the tests' tolerance.  The scene generator is the test suite's (`tests/test_hip_parity.py: _random_scene`); `--big` draws
larger frames and more primitives (crowded bins, saturated ordinals).  Prints a progress line every 20 scenes and the
first failing (seed, scene number), then exits non-zero.  Test infrastructure: the oracle is only the checker here.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--shadows", action="store_true",
                    help="instead: the shadow pass with candidates from light-space tile bins against its all-pairs form "
                         "(visibility bits and re-shaded image bit for bit), random lights incl. inside the cloud")
    ap.add_argument("--views", action="store_true",
                    help="instead: render_views (one batch of 2-5 random cameras, perspective or orthographic under torch "
                         "shading) against per-view render(), bit for bit")
    ap.add_argument("--adversarial", action="store_true",
                    help="instead: the directed scenes of tests/adversarial_scenes.py (primitives with coordinates of "
                         "2^30..2^66 that cross the view), all four kinds in turn, with the checks of "
                         "tests/test_hip_adversarial.py")
    args = ap.parse_args()
    if args.adversarial and not args.shadows:
        from adversarial_scenes import KINDS, huge_scene
        import test_hip_adversarial as A
        rng = np.random.RandomState(args.seed)
        t0 = time.time()
        it = off = total = 0
        while time.time() - t0 < args.seconds:
            kind = KINDS[it % 4]
            sc = huge_scene(rng, kind)
            ref, bad = A.check_scene(sc, f"{kind} seed {args.seed} scene {it}")
            # the same scene as a row slab (the multi-GPU partition: its bounding-ball pre-cull runs only there) and
            # under the torch backend's semantics, binned against all pairs
            H = sc["camera"]["viewport"][3]
            r0 = int(rng.randint(0, H - 1))
            r1 = int(rng.randint(r0 + 1, H + 1))
            slab = A._render(sc, rows=(r0, r1))
            tb, te = A._render(sc, mode="binned", shading="torch"), A._render(sc, mode="exact", shading="torch")
            for what, got, want in (("rows", slab, {k: ref[k][r0:r1] for k in ref}), ("torch shading", tb, te)):
                for k in ("nearest", "depth", "image"):
                    ne = ~((got[k] == want[k]) | (np.isnan(got[k]) & np.isnan(want[k])))
                    if ne.any():
                        print(f"FAIL {kind} seed {args.seed} scene {it} ({what} {r0}:{r1}): {k} differs on {ne.sum()} values", flush=True)
                        sys.exit(1)
            off += bad
            total += ref["depth"].size
            it += 1
            if it % 200 == 0:
                print(f"[adversarial seed {args.seed}] {it} scenes, {off} of {total} pixels off the oracle (all won by huge "
                      f"primitives), modes bit-identical", flush=True)
        print(f"[adversarial seed {args.seed}] DONE: {it} scenes, modes bit-identical; {off} of {total} pixels "
              f"({off / max(total, 1):.2e}) differ from the oracle, every one of them won by a huge primitive", flush=True)
        return
    import test_hip_parity as T
    from surf_renderer_amd import render
    from surf_renderer_amd.scene import scene_to_numpy
    from oracle import np_oracle

    rng = np.random.RandomState(args.seed)
    t0 = time.time()
    it = n_oracle = 0

    def same(a, b, what):
        for k in ("nearest", "depth", "image"):
            x = a[k].cpu().numpy() if torch.is_tensor(a[k]) else a[k]
            y = b[k].cpu().numpy() if torch.is_tensor(b[k]) else b[k]
            if not np.array_equal(x, y, equal_nan=True):
                print(f"FAIL seed {args.seed} scene {it} ({what}): {k} differs on {(x != y).sum()} values", flush=True)
                sys.exit(1)

    if args.views:
        from surf_renderer_amd import render_views
        while time.time() - t0 < args.seconds:
            sc = T._random_scene(rng)
            kw = {}
            if rng.randint(2):
                sc["lights"]["attenuation"] = np.array([[1, 0, 0], [0.5, 0.1, 0.01]], dtype=np.float32)
                sc["lights"]["ambient"] = np.array([0.01, 0.02, 0.01], dtype=np.float32)
                sc["materials"]["coeffs"] = np.array([[1, 0, 0], [0.7, 0.3, 5], [0.5, 0.5, 20]], dtype=np.float32)
                kw = {"shading": "torch", "double_sided": bool(rng.randint(2))}
                if rng.randint(2):
                    sc["camera"]["proj_type"] = "ortho"
                    sc["camera"]["near"] = max(sc["camera"]["near"], 0.01)
            cams = []
            for _ in range(int(rng.randint(2, 6))):
                e = rng.normal(size=3)
                e = e / np.linalg.norm(e) * rng.choice([0.5, 2.0, 3.0, 6.0])
                cams.append(dict(sc["camera"], eye=[*map(float, e), 1.0]))
            vb = render_views(sc, cams, device="cuda:0", **kw)
            for i, cam in enumerate(cams):
                one = render({**sc, "camera": cam}, device="cuda:0", **kw)
                same({"image": vb["image"][i], "depth": vb["depth"][i], "nearest": vb["nearest"][i].to(torch.int64)},
                     one, f"view {i} of {len(cams)} {kw} {sc['camera'].get('proj_type', 'perspective')}")
            it += 1
            if it % 20 == 0:
                print(f"[fuzz seed {args.seed} views] {it} batches ok, {time.time() - t0:.0f} s", flush=True)
        print(f"[fuzz seed {args.seed} views] DONE: {it} batches of 2-5 views, all equal to per-view renders", flush=True)
        return

    if args.shadows:
        import test_hip_torch_shading as TS
        n_shadowed = 0
        if args.adversarial:                   # --shadows --adversarial: the directed huge-primitive scenes as occluders
            from adversarial_scenes import KINDS, huge_scene
        while time.time() - t0 < args.seconds:
            sc = huge_scene(rng, KINDS[it % 4]) if args.adversarial else T._random_scene(rng)
            sc["camera"]["near"] = max(sc["camera"]["near"], 0.01)
            n_l = int(rng.randint(1, 5))
            lp = rng.normal(size=(n_l, 3))
            lp = lp / np.linalg.norm(lp, axis=1, keepdims=True) * rng.choice([0.2, 1.0, 2.5, 6.0, 12.0], size=(n_l, 1))
            sc["lights"] = {"pos": np.concatenate([lp, np.ones((n_l, 1))], 1).astype(np.float32),
                            "color_idx": rng.randint(1, 3, n_l)}
            sc = TS._with_torch_inputs(sc)
            kw = {"double_sided": bool(rng.randint(2)), "use_quartic": bool(rng.randint(2))}
            (img_b, vis_b, depth), (img_a, vis_a, _) = TS._shadow_both_ways(sc, **kw)
            if not torch.equal(vis_b, vis_a) or not torch.equal(img_b.view(torch.int32), img_a.view(torch.int32)):
                print(f"FAIL seed {args.seed} scene {it} (shadows {kw}): visibility differs on "
                      f"{int((vis_b != vis_a).sum())} pixels, image on "
                      f"{int((img_b.view(torch.int32) != img_a.view(torch.int32)).sum())} values", flush=True)
                sys.exit(1)
            hit = depth <= float(sc["camera"]["far"])
            n_shadowed += int(((vis_a[hit] & ((1 << n_l) - 1)) != (1 << n_l) - 1).sum())
            it += 1
            if it % 20 == 0:
                print(f"[fuzz seed {args.seed} shadows] {it} scenes ok, {n_shadowed} shadowed pixels so far, "
                      f"{time.time() - t0:.0f} s", flush=True)
        print(f"[fuzz seed {args.seed} shadows] DONE: {it} scenes, {n_shadowed} shadowed pixels, binned = all pairs "
              f"bit for bit", flush=True)
        return

    while time.time() - t0 < args.seconds:
        sc = T._random_scene(rng)
        if args.big:
            W, H = int(rng.choice([256, 333, 512, 640])), int(rng.choice([192, 256, 400, 512]))
            sc["camera"]["viewport"] = [0, 0, W, H]
            if "disk" in sc["objects"] and rng.randint(2):
                n = int(rng.choice([20000, 60000]))
                f32 = lambda a: np.asarray(a, dtype=np.float32)          # noqa: E731
                sc["objects"]["disk"] = {
                    "pos": f32(np.concatenate([rng.uniform(-1.5, 1.5, (n, 3)), np.ones((n, 1))], 1)),
                    "normal": f32(np.concatenate([rng.normal(size=(n, 3)), np.zeros((n, 1))], 1)),
                    "material_idx": rng.randint(0, 3, n),
                    "radius": f32(np.exp(rng.uniform(np.log(0.002), np.log(0.2), n)))}
        H = sc["camera"]["viewport"][3]
        ref = render(sc, device="cuda:0", mode="exact")
        for wpt in (1, 4):
            same(render(sc, device="cuda:0", mode="binned", waves_per_tile=wpt), ref, f"binned wpt {wpt}")
        r0 = int(rng.randint(0, H - 1))
        r1 = int(rng.randint(r0 + 1, H + 1))
        same(render(sc, device="cuda:0", rows=(r0, r1)), {k: ref[k][r0:r1] for k in ("nearest", "depth", "image")},
             f"rows {r0}:{r1}")
        sc["lights"]["attenuation"] = np.array([[1, 0, 0], [0.5, 0.1, 0.01]], dtype=np.float32)
        sc["lights"]["ambient"] = np.array([0.01, 0.02, 0.01], dtype=np.float32)
        sc["materials"]["coeffs"] = np.array([[1, 0, 0], [0.7, 0.3, 5], [0.5, 0.5, 20]], dtype=np.float32)
        ds = bool(rng.randint(2))
        same(render(sc, device="cuda:0", mode="binned", shading="torch", double_sided=ds),
             render(sc, device="cuda:0", mode="exact", shading="torch", double_sided=ds), "torch shading")
        W = sc["camera"]["viewport"][2]
        if not args.big and W * H <= 64 * 80 and sum(len(g["material_idx"]) for g in sc["objects"].values()) <= 800:
            for k in ("attenuation", "ambient"):
                sc["lights"].pop(k)
            sc["materials"].pop("coeffs")
            T.assert_parity({k: ref[k].cpu().numpy() for k in ("image", "depth", "nearest")},
                            np_oracle.render(scene_to_numpy(sc, round_fp32=True)))
            n_oracle += 1
        it += 1
        if it % 20 == 0:
            print(f"[fuzz seed {args.seed}{' big' if args.big else ''}] {it} scenes ok ({n_oracle} also against the "
                  f"oracle), {time.time() - t0:.0f} s", flush=True)
    print(f"[fuzz seed {args.seed}{' big' if args.big else ''}] DONE: {it} scenes, {n_oracle} against the oracle, "
          f"all identical", flush=True)


if __name__ == "__main__":
    main()
