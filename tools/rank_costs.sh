#!/bin/bash
# usage: tools/rank_costs.sh P [bench args]  -> ms/step of every rank's slab of a P-rank job, rehearsed on one GPU
P=$1; shift
for ((r=0; r<P; r++)); do
  python bench.py --as-rank $r/$P --steps 400 --warmup 40 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rank $r/$P ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"
done
