#!/bin/bash
# build diagnostic variants of libsrh.so: tools_mkabl.sh name "-DFLAG ..." [name flags]...
mkdir -p build/abl
F="-O3 --offload-arch=gfx950 -shared -fPIC -std=c++17 -ffp-contract=off -I include -I surf_renderer_amd/csrc surf_renderer_amd/csrc/srh.hip"
while [ $# -gt 1 ]; do
  hipcc $F $2 -o build/abl/$1.so 2>/dev/null || echo "build $1 failed"
  shift 2
done
ls build/abl
