#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh TAG   -> gpurun_out/TAG/{bench.json,kernel_stats.csv,pmc.txt,configs.jsonl,ranks.txt}
TAG=${1:-r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python3 bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/r_kernel_stats.csv $OUT/kernel_stats.csv
{
  echo "rocprofv3 --pmc <counters> --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --graph off   (tools/pmc.sh); mean per launch"
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "== $c"; tools/pmc.sh "$c" --graph off --inflight 1
  done
} > $OUT/pmc.txt 2>&1
python tools/bench_configs.py > $OUT/configs.jsonl 2> $OUT/configs.err
{ for P in 2 4 8; do tools/rank_costs.sh $P; done; } > $OUT/ranks.txt 2>&1
tail -c 600 $OUT/bench.json
