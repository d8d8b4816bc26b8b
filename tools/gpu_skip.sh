#!/bin/bash
run() { timeout -k 10 120 python bench.py --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $2', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['schedule'], d['config']['check'][:14])"; }
for i in 1 2 3; do run auto "--steps 20 --warmup 5"; done
run auto ""
run frames "--schedule frames --steps 20 --warmup 5"
run small "--prims 2000 --width 256 --height 256 --steps 50"
run fast "--mode fast --prims 2000 --width 256 --height 256 --steps 50"
run graphoff "--graph off --steps 50"
