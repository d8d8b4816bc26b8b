#!/bin/bash
# correctness of the current build (all GPU tests), then the bench in its three usual forms
set -o pipefail
OUT=gpurun_out/${1:-s2}; mkdir -p $OUT
echo "[s2] tests"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
echo "[s2] bench"
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err || echo "bench20 failed"
timeout -k 10 300 python bench.py --no-cpu-baseline --inflight 1 > $OUT/bench_fl1.json 2> $OUT/bench_fl1.err || echo "bench_fl1 failed"
python - <<PY
import json
for f in ("bench","bench20","bench_fl1"):
    try:
        d=json.loads(open("$OUT/"+f+".json").read().strip().splitlines()[-1]); print(f, "ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],4), d["config"].get("check"), d["config"].get("warmup_steps_run"))
    except Exception as e: print(f, "ERR", e)
PY
echo "[s2] done"
