#!/bin/bash
# A/B on one GPU box: tools/ab.sh TAG variant...  ("base" = in-tree lib; others = build/abl/<name>.so).
# Per variant, interleaved over two rounds so that drift of the box shows: default bench (400 steps), 20-step bench;
# then once per variant the secondary configs.
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { timeout -k 10 150 python bench.py --no-cpu-baseline $2 2>>$OUT/$1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$1] $2', '-> ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['check'][:14])" | tee -a $OUT/ab.txt; }
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
    run $v ""
    run $v "--steps 20 --warmup 5"
  done
done
if [ -z "$AB_NO_CONFIGS" ]; then
for v in "$@"; do
  if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
  timeout -k 10 300 python tools/bench_configs.py > $OUT/configs_$v.jsonl 2>$OUT/configs_$v.err
  python - <<PY | tee -a $OUT/ab.txt
import json
for l in open("$OUT/configs_$v.jsonl"):
    d=json.loads(l); print("[$v]", d["config"][:40], {k:round(v,4) for k,v in d.items() if k.startswith("ms_")})
PY
done
fi
echo "[ab] done"
