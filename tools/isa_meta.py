#!/usr/bin/env python3
"""Register / spill metadata of every kernel of a libsrh build.
usage: tools/isa_meta.py NAME [SRC_DIR] [-DFLAG ...]   -> build/isa/NAME.s, prints one line per kernel
SRC_DIR defaults to the working tree (surf_renderer_amd/csrc + include); build/src_<x> of tools/mkref.sh also works."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
rest = sys.argv[2:]
src = os.path.join(ROOT, "surf_renderer_amd", "csrc")
inc = os.path.join(ROOT, "include")
if rest and not rest[0].startswith("-"):
    src, inc = os.path.join(rest[0], "csrc"), os.path.join(rest[0], "include")
    rest = rest[1:]
os.makedirs(os.path.join(ROOT, "build", "isa"), exist_ok=True)
out = os.path.join(ROOT, "build", "isa", name + ".s")
cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-I", inc, "-I", src,
       "--cuda-device-only", "-S", os.path.join(src, "srh.hip"), "-o", out, *rest]
subprocess.run(cmd, check=True)
text = open(out).read()
rows = []
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.agpr_count|\Z)", text, re.S):
    pass
# the metadata block at the end: one YAML entry per kernel
meta = text[text.rfind("amdhsa.kernels:"):]
for ent in meta.split("  - .agpr_count:")[1:]:
    def f(k):
        mm = re.search(r"\." + k + r":\s+(\S+)", ent)
        return mm.group(1) if mm else "?"
    sym = f("name")
    dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("(anonymous namespace)::", "").replace("void ", "").replace("srh::", "")
    dem = re.sub(r"\(.*", "", dem)
    rows.append((dem, f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"),
                 f("private_segment_fixed_size"), f("group_segment_fixed_size")))
print(f"{'kernel':58s} vgpr vspill sgpr sspill scratch lds")
for r in sorted(rows):
    print(f"{r[0][:58]:58s} {r[1]:>4s} {r[2]:>6s} {r[3]:>4s} {r[4]:>6s} {r[5]:>7s} {r[6]:>5s}")
