#!/bin/bash
# build libsrh.so as of a git revision into build/abl/<name>.so (for A/B runs on one GPU box: SRH_LIB=build/abl/<name>.so)
# usage: tools/mkref.sh NAME REV ["-DFLAG ..."]
set -e
NAME=$1; REV=$2; shift 2
D=build/src_$NAME
rm -rf $D; mkdir -p $D/csrc $D/include build/abl
for f in $(git ls-tree --name-only $REV surf_renderer_amd/csrc/); do git show $REV:$f > $D/csrc/$(basename $f); done
git show $REV:include/srh.h > $D/include/srh.h
hipcc -O3 --offload-arch=gfx950 -shared -fPIC -std=c++17 -ffp-contract=off -I $D/include -I $D/csrc $D/csrc/srh.hip "$@" -o build/abl/$NAME.so
ls -la build/abl/$NAME.so
