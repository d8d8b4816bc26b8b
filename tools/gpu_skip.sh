#!/bin/bash
run() { timeout -k 10 120 python bench.py --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $2', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['check'][:14])"; }
for i in 1 2 3; do run frames "--steps 20 --warmup 5"; done
for i in 1 2 3; do run stages "--schedule stages --steps 20 --warmup 5"; done
for i in 1 2; do run stages-flat "--schedule stages --flat-priority --steps 20 --warmup 5"; done
run stages "--schedule stages"
run stages3 "--schedule stages --render-streams 3 --inflight 4"
