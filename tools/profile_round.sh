#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh TAG COMMIT -> gpurun_out/TAG/{bench.json,bench20.json,kernel_stats.csv,pmc.json,configs.jsonl,bwd_shadow.jsonl,ranks.txt}
TAG=${1:-r}; COMMIT=${2:-unknown}
OUT=gpurun_out/$TAG
mkdir -p $OUT
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
timeout -k 10 500 python tools/prof_bwd_shadow.py --cases bwd_mesh,bwd_mesh_resident,bwd_mesh_resident_tch,bwd_plane,bwd_discs,shadow_mesh_allpairs,shadow_discs_allpairs,shadow_cfg5 > $OUT/bwd_shadow.jsonl 2> $OUT/bwd_shadow.err
echo "[profile] rank rehearsal"
{ for P in 2 4 8; do tools/rank_costs.sh $P; done; } > $OUT/ranks.txt 2>&1
python - <<PY
import json
for f in ("bench","bench20"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1]); print(f, "ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), d["config"].get("check"))
    except Exception as e: print(f, "ERR", e)
PY
echo "[profile] done"
