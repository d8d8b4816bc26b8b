#!/bin/bash
run() { env $3 timeout -k 10 120 python bench.py --no-cpu-baseline $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $2', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['check'][:14])"; }
for v in base pw6 pw8 base; do
if [ $v = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
run $v "--inflight 3" A=1
done
