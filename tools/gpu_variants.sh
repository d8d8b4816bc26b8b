#!/bin/bash
# usage: tools/gpu_variants.sh TAG variant...   ("base" = in-tree lib; others = build/abl/<name>.so)
# per variant: ms/step at 3 and 1 frames in flight, then rocprof per-kernel durations at 1 in flight
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
  for fl in 3 1; do
    timeout -k 10 120 python bench.py --no-cpu-baseline --inflight $fl 2>$OUT/${v}_$fl.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v inflight $fl', 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['config']['check'][:12])" | tee -a $OUT/variants.txt
  done
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -o r -- python3 bench.py --no-cpu-baseline --no-check --inflight 1 --graph off --steps 200 > $OUT/trace_$v.log 2>&1
  python - <<PY | tee -a $OUT/variants.txt
import csv
try:
    rows=list(csv.DictReader(open("$OUT/trace_$v/r_kernel_stats.csv")))
    print("$v kernels(us):", ", ".join(f"{r['Name'].split('(')[0].split('::')[-1][:18]}={float(r['AverageNs'])/1e3:.1f}" for r in rows[:7]))
except Exception as e: print("$v trace ERR", e)
PY
done
echo "[variants] done"
