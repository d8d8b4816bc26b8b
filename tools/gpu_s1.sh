#!/bin/bash
# round-2 session 1: microbenchmarks, baseline, ablation breakdown, backward / shadow baseline
set -o pipefail
OUT=gpurun_out/s1; mkdir -p $OUT
echo "[s1] ubench"; 
timeout -k 10 120 build/ubench/issue > $OUT/ubench_issue.txt 2>&1 || echo "issue failed"
timeout -k 10 120 build/ubench/valu > $OUT/ubench_valu.txt 2>&1 || echo "valu failed"
timeout -k 10 300 build/ubench/divcheck > $OUT/ubench_div.txt 2>&1 || echo "div failed"
tail -3 $OUT/ubench_issue.txt
echo "[s1] baseline"
timeout -k 10 300 python bench.py > $OUT/base.json 2> $OUT/base.err || echo "base failed"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/base20.json 2> $OUT/base20.err || echo "base20 failed"
python - <<'PY'
import json
for f in ("base","base20"):
    try:
        d=json.loads(open(f"gpurun_out/s1/{f}.json").read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["roofline"]["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
echo "[s1] ablations"
for v in base renderonly binonly noshade noconfirm noloop ro_noshade; do
  for fl in 3 1; do
    if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --inflight $fl 2>$OUT/abl_${v}_$fl.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v inflight $fl', 'ms/step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4))" | tee -a $OUT/abl.txt
  done
done
unset SRH_LIB
echo "[s1] backward / shadow"
timeout -k 10 400 python tools/prof_bwd_shadow.py > $OUT/bwd_shadow.jsonl 2> $OUT/bwd_shadow.err || echo "bwd_shadow failed"
cat $OUT/bwd_shadow.jsonl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bwd -o r -- python3 tools/prof_bwd_shadow.py --cases bwd_mesh,bwd_mesh_tch,bwd_plane,bwd_discs --steps 10 > $OUT/trace_bwd.log 2>&1 || echo "trace failed"
cp $OUT/trace_bwd/r_kernel_stats.csv $OUT/bwd_kernel_stats.csv 2>/dev/null
head -12 $OUT/bwd_kernel_stats.csv | cut -c1-160
echo "[s1] done"
