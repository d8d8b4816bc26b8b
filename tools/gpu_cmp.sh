#!/bin/bash
# usage: tools/gpu_cmp.sh variant...  ("base" = in-tree lib, others build/abl/<name>.so): default bench, 20-step bench, one frame in flight
run() { timeout -k 10 120 python bench.py --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $2', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['schedule'], d['config']['check'][:14])"; }
for v in "$@"; do
  if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
  run $v ""
  run $v "--schedule frames"
  run $v "--steps 20 --warmup 5"
  run $v "--inflight 1"
done
