"""Which part of a captured render + loss + backward step breaks hipStreamEndCapture?  Each variant runs in its own
process (a variant may die with a segmentation fault).  usage: python tools/diag_capture.py [variant]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = ["fwd_only", "fwd_loss", "torch_bwd_only", "custom_bwd_sum", "custom_bwd_full", "custom_bwd_nozero",
            "bwd_call_direct"]


def run(variant):
    import numpy as np
    import torch
    from surf_renderer_amd import ResidentScene, synthetic, renderer
    scene = synthetic.bunny_mesh_scene(160, 128)
    tri = scene["objects"]["triangle"]
    face = torch.tensor(np.asarray(tri["face"], dtype=np.float32), device="cuda:0", requires_grad=True)
    normal = torch.tensor(np.asarray(tri["normal"], dtype=np.float32), device="cuda:0", requires_grad=True)
    scene["objects"]["triangle"] = dict(tri, face=face, normal=normal)
    rs = ResidentScene(scene, device="cuda:0")
    target = torch.rand((128, 160, 3), device="cuda:0")
    x = torch.rand((128, 160, 3), device="cuda:0", requires_grad=True)

    def body():
        if variant == "fwd_only":
            with torch.no_grad():
                return rs.render()["image"]
        if variant == "fwd_loss":
            with torch.no_grad():
                return ((rs.render()["image"] - target) ** 2).sum()
        if variant == "torch_bwd_only":
            loss = ((x - target) ** 2).sum()
            loss.backward()
            return loss
        if variant == "custom_bwd_sum":
            loss = rs.render()["image"].sum()
            loss.backward()
            return loss
        if variant == "custom_bwd_full":
            res = rs.render()
            loss = ((res["image"] - target) ** 2).sum() + 0.01 * res["depth"].clamp(max=50.0).sum()
            loss.backward()
            return loss
        if variant == "custom_bwd_nozero":
            res = rs.render()
            loss = res["image"].sum()
            g, = torch.autograd.grad(loss, [face])
            return g
        if variant == "bwd_call_direct":
            # the library's backward launched by hand, no autograd at all
            with torch.no_grad():
                image, depth, nearest = renderer.render_buffers(rs.buf, rs.cam)
            ctx = type("Ctx", (), {})()
            ctx.buf, ctx.cam, ctx.rows, ctx.mode, ctx.shade = rs.buf, rs.cam, None, "auto", rs.shade
            ctx.has_vis = False
            ctx.saved_tensors = (depth, nearest)
            ctx.needs_input_grad = (False,) * 5 + tuple(t.requires_grad for t in rs.inputs)
            out = renderer._RenderFunction.backward(ctx, torch.ones_like(image), None, None)
            return [o for o in out if o is not None][0]
        raise SystemExit("unknown variant")

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            face.grad = normal.grad = x.grad = None
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    face.grad = normal.grad = x.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.autograd.set_multithreading_enabled(False), torch.cuda.graph(g):
        out = body()
    g.replay()
    torch.cuda.synchronize()
    print(f"{variant}: ok ({float(out.float().abs().sum()):.6g})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for v in VARIANTS:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), v], capture_output=True, text=True, timeout=300)
            tail = (p.stdout.strip().splitlines() or [""])[-1]
            print(f"{v}: rc {p.returncode} {tail if p.returncode == 0 else ' | '.join(p.stderr.strip().splitlines()[-3:])[:400] if p.stderr.strip() else ''}", flush=True)
