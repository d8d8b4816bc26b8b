#!/bin/bash
OUT=gpurun_out/${1:-sched2}; mkdir -p $OUT
run() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>$OUT/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))" || tail -3 $OUT/err.txt; }
for n in 1 2 3 4; do run --schedule render-only --inflight $n --no-check; done
for n in 1 2 3 4; do run --schedule bin-only --inflight $n --no-check; done
for n in 1 2 3; do run --schedule frames --inflight $n; done
