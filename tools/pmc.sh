#!/bin/bash
# usage: tools_pmc.sh "<counters>" [bench args]   -> per-kernel mean counter values
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C="$1"; shift
rm -rf gpurun_out/pmc_tmp
timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_tmp -- python3 bench.py --no-cpu-baseline --no-check --warmup-ms 0 --steps 3 --warmup 1 "$@" > gpurun_out/pmc_tmp.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_tmp/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"].split("(")[0][-28:]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in acc:
    print(k, {c: round(v / n[k][c]) for c, v in acc[k].items()})
PY
