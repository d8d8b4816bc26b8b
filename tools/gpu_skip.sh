#!/bin/bash
export SRH_LIB=$PWD/build/abl/dclk.so SRH_DIAG_CLK=1
mkdir -p gpurun_out/ring
for fl in 3 1; do for d in none p32dummy,filldummy; do
SRH_DIAG_SKIP=$d timeout -k 10 100 python tools/diag_time.py --inflight $fl 2>&1 | grep -E "diag|clock"; done; done
