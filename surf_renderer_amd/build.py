"""Build recipe for libsrh.so (hipcc, gfx950 only, in-tree so the .so travels with the checkout)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(REPO, "include")
LIB_PATH = os.path.join(PKG, "libsrh.so")

SOURCES = [os.path.join(CSRC, "srh.hip")]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "srh.h")]

# -ffp-contract=off: the fp64 truth path mirrors numpy's unfused arithmetic; kernels that want FMAs
# ask for them explicitly.
FLAGS = ["-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the hip backend needs the ROCm toolchain")
    return exe


def command(extra: List[str] = (), out: str = LIB_PATH) -> List[str]:
    return [hipcc(), *FLAGS, *extra, "-I", INCLUDE, "-I", CSRC, *SOURCES, "-o", out]


def stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > built for p in SOURCES + HEADERS)


def build_lib(force: bool = False, verbose: bool = False, extra: List[str] = ()) -> str:
    """Compile libsrh.so if missing or older than its sources; returns its path."""
    if force or stale():
        # Compile next to the target and rename into place: several ranks of one job may find the library missing at
        # the same time, and none of them may ever dlopen a half-written file.  One process compiles (file lock), into a
        # FIXED temporary name: hipcc derives the code object's identity from the output path, so a name with the pid in
        # it gave every build another sha256 (bench.py quotes the hash the PMC counters were taken with).
        import fcntl
        with open(LIB_PATH + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or stale():                   # somebody else may have built it while we waited
                    tmp = LIB_PATH + ".build"
                    cmd = command(list(extra), out=tmp)
                    if verbose:
                        print(" ".join(cmd).replace(tmp, LIB_PATH), file=sys.stderr)
                    proc = subprocess.run(cmd, capture_output=True, text=True)
                    if proc.returncode != 0:
                        if os.path.exists(tmp):
                            os.remove(tmp)
                        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
                    os.replace(tmp, LIB_PATH)
                    if verbose and proc.stderr:
                        print(proc.stderr, file=sys.stderr)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True,
                    extra=[a for a in sys.argv[1:] if a != "--force"]))
