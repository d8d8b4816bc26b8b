#!/bin/bash
for v in base bt4 base bt4; do
if [ $v = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
echo "== $v"; timeout -k 10 200 python tools/prof_bwd_shadow.py --cases bwd_mesh_resident_tch --steps 40 2>&1 | tail -1 | cut -c1-220
done
