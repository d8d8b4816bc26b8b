#!/bin/bash
# usage: tools_abl.sh variant...   (variants are build/abl/<name>.so; "base" = in-tree lib)
for v in "$@"; do
  if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --mode ${MODE:-binned} --no-cpu-baseline ${BENCH_ARGS} 2>gpurun_out/abl_$v.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
done
