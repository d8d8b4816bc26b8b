#!/bin/bash
# usage: tools/gpu_timeline.sh TAG [bench args]: kernel timeline of a few steady-state frames
OUT=gpurun_out/${1:-tl}; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python3 bench.py --no-cpu-baseline --no-check --steps 60 --warmup 20 --warmup-ms 0 "$@" > $OUT/trace.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/trace/r_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=lambda n: n.split("(")[0].split("::")[-1][:16]
# steady-state window: last 40% of dispatches
sel=rows[int(len(rows)*0.6):int(len(rows)*0.6)+36]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:9.1f} {e:9.1f} {e-s:7.1f}  q={r.get('Queue_Id','?'):>3} {names(r['Kernel_Name'])}")
PY
