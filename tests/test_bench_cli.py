"""bench.py's command line (no GPU needed): the help text renders and the contract's flags exist."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_renders_and_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    for flag in ("--gpus", "--steps", "--warmup", "--slabs", "--check"):
        assert flag in out.stdout


def test_bench_refuses_to_run_without_a_gpu_or_with_a_wrong_world():
    import torch
    env = dict(os.environ, WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
    if not torch.cuda.is_available():
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True,
                             timeout=300, env=env)
        assert out.returncode != 0 and "needs a GPU" in (out.stderr + out.stdout)


def test_render_cli_refuses_torch_only_flags_on_other_backends():
    scene = os.path.join(ROOT, "assets", "scenes", "basic.json")
    for extra in (["--use", "np", "--shadow"], ["--use", "hip", "--double_sided"]):
        out = subprocess.run([sys.executable, "-m", "surf_renderer_amd.render_cli", "--scene", scene, *extra],
                             capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert out.returncode == 2 and "--shading torch" in out.stderr
