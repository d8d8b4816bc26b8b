#!/bin/bash
OUT=gpurun_out/${1:-sched}; mkdir -p $OUT
run() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>$OUT/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['launch'][:18], d['config']['check'][:10])" || tail -3 $OUT/err.txt; }
run --schedule frames --inflight 2
run --schedule frames --inflight 3
run --schedule frames --inflight 4
run --schedule frames --inflight 6
run --schedule stages --inflight 2
run --schedule stages --inflight 3
run --schedule stages --inflight 4
run --schedule stages --inflight 3 --steps 20 --warmup 5
run --schedule frames --inflight 3 --steps 20 --warmup 5
run --schedule stages --inflight 2 --steps 20 --warmup 5
