#!/bin/bash
OUT=gpurun_out/${1:-sched3}; mkdir -p $OUT
run() { timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>$OUT/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['check'][:9])" || tail -3 $OUT/err.txt; }
run --schedule frames --inflight 3
run --schedule stages --inflight 3 --render-streams 1
run --schedule stages --inflight 3 --render-streams 2
run --schedule stages --inflight 4 --render-streams 2
run --schedule stages --inflight 4 --render-streams 3
run --schedule stages --inflight 3 --render-streams 2 --flat-priority
run --schedule stages --inflight 3 --render-streams 2 --steps 20 --warmup 5
