#!/bin/bash
run() { env $3 timeout -k 10 120 python bench.py --no-cpu-baseline --no-check $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $2', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))"; }
for fl in 1 2 3 4; do run base "--schedule render-only --inflight $fl" A=1; done
for fl in 1 2 3; do run base "--schedule bin-only --inflight $fl" A=1; done
