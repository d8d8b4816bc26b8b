#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh TAG COMMIT
#   -> gpurun_out/TAG/{bench.json,bench20.json,kernel_stats.csv,pmc.json,configs.jsonl,bwd_shadow.jsonl,
#                      bwd_shadow_kernel_stats.csv,ranks.txt,views.txt}
# a third argument picks a half (gpurun calls are limited to 20 minutes): 1 = bench, trace, PMC, configs; 2 = the rest
TAG=${1:-r}; COMMIT=${2:-unknown}; PART=${3:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$PART" != 2 ]; then
echo "[profile] bench"
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "[profile] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python3 bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/r_kernel_stats.csv $OUT/kernel_stats.csv
echo "[profile] pmc"
timeout -k 10 900 python3 tools/collect_pmc.py --commit $COMMIT --out $OUT/pmc.json > $OUT/pmc.log 2>&1; tail -2 $OUT/pmc.log | cut -c1-300
echo "[profile] configs"
timeout -k 10 400 python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
fi
if [ "$PART" != 1 ]; then
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CASES=bwd_mesh,bwd_mesh_resident,bwd_mesh_resident_tch,bwd_mesh_captured,bwd_mesh_captured_tch,bwd_plane,bwd_discs,shadow_mesh_allpairs,shadow_discs_allpairs,shadow_cfg5,shadow_mesh,shadow_discs
timeout -k 10 500 python tools/prof_bwd_shadow.py --cases $CASES > $OUT/bwd_shadow.jsonl 2> $OUT/bwd_shadow.err
echo "[profile] backward / shadow kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bwd -o r -- python3 tools/prof_bwd_shadow.py --cases bwd_mesh_resident,bwd_mesh_resident_tch,shadow_cfg5,shadow_mesh > $OUT/trace_bwd.log 2>&1
cp $OUT/trace_bwd/r_kernel_stats.csv $OUT/bwd_shadow_kernel_stats.csv
echo "[profile] rank rehearsal"
{ for P in 2 4 8; do for S in contiguous cost; do for R in $(seq 0 $((P/2-1))); do
    timeout -k 10 120 python bench.py --no-cpu-baseline --as-rank $R/$P --slabs $S --batch-call on 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('P=$P slabs=$S rank $R:', round(d['ms_per_step']*1e3,1), 'us per frame')"
  done; done; done; } > $OUT/ranks.txt 2>&1
echo "[profile] views"
timeout -k 10 300 python tools/bench_views.py --views 4 --streams 2 --calls 50 > $OUT/views.txt 2>&1
fi
if [ "$PART" != 2 ]; then
python - <<PY
import json
for f in ("bench","bench20"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1]); print(f, "ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), d["config"].get("check"), d.get("executed_pair_tests",{}).get("fraction"))
    except Exception as e: print(f, "ERR", e)
PY
fi
echo "[profile] done"
