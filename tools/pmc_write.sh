#!/bin/bash
# usage (GPU box): tools/pmc_write.sh variant...  -> WRITE_SIZE / FETCH_SIZE (KiB per launch) of the render kernel per variant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset SRH_LIB; else export SRH_LIB=$PWD/build/abl/$v.so; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf gpurun_out/pmc_w; timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --no-cpu-baseline --no-check --warmup-ms 0 --steps 3 --warmup 1 --graph off --inflight 1 > /dev/null 2>&1
    python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/pmc_w/*/*counter_collection.csv")
rows=[r for r in csv.DictReader(open(f[0])) if "k_render_binned" in r["Kernel_Name"] and r["Counter_Name"]=="$c"] if f else []
print("$v $c", round(sum(float(r["Counter_Value"]) for r in rows)/max(len(rows),1),1), "KiB per launch over", len(rows), "launches")
PY
  done
done
