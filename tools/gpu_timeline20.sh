#!/bin/bash
# usage: tools/gpu_timeline20.sh TAG [bench args]: kernel timeline of the driver-style 20-step timed region (the last
# dispatches of the run: binning + render kernel of each timed frame), start / end / duration in us from the first one
OUT=gpurun_out/${1:-tl20}; shift; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o r -- python3 bench.py --no-cpu-baseline --no-check --steps 20 --warmup 5 "$@" > $OUT/trace.log 2>&1
python3 - <<PY | tee $OUT/timeline.txt
import csv
rows=list(csv.DictReader(open("$OUT/trace/r_kernel_trace.csv")))
rows=[r for r in rows if "k_prep" in r["Kernel_Name"] or "k_render" in r["Kernel_Name"] or "k_put_frame" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=lambda n: n.split("(")[0].split("::")[-1][:18]
n=20
# the timed region: the last 20 render kernels and everything from the first of their preps on
rk=[i for i,r in enumerate(rows) if "k_render" in r["Kernel_Name"]]
first=rk[-n]
# walk back to the prep that belongs to it (same queue, just before)
q=rows[first].get("Queue_Id")
j=first
while j>0 and not ("k_prep" in rows[j]["Kernel_Name"] and rows[j].get("Queue_Id")==q): j-=1
sel=rows[j:]
t0=int(sel[0]["Start_Timestamp"])
for r in sel:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print(f"{s:9.1f} {e:9.1f} {e-s:7.1f}  q={r.get('Queue_Id','?'):>3} {names(r['Kernel_Name'])}")
PY
