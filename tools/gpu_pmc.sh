#!/bin/bash
# usage: tools/gpu_pmc.sh TAG [lib.so]  -> gpurun_out/TAG/pmc.txt : SQ counters of every kernel of one frame (inflight 1, eager)
set -o pipefail
OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT
[ -n "$2" ] && export SRH_LIB=$PWD/$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ ! -f gpurun_out/counters_list.txt ]; then timeout -k 10 120 rocprofv3 -L > gpurun_out/counters_list.txt 2>&1; fi
{
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "== $c"; tools/pmc.sh "$c" --graph off --inflight 1
  done
} > $OUT/pmc.txt 2>&1
grep -A8 "== SQ_INSTS_VALU" $OUT/pmc.txt | cut -c1-250
