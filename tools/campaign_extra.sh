#!/bin/bash
# extra campaign legs (seed base in $1); every leg writes its own file under gpurun_out/fz2 (progress lines keep the run alive)
S=${1:-5100}
mkdir -p gpurun_out/fz2
timeout -k 10 330 python tools/fuzz_campaign.py --seed $((S+1)) --seconds 280 --adversarial > gpurun_out/fz2/adv.txt 2>&1; tail -1 gpurun_out/fz2/adv.txt
timeout -k 10 260 python tools/fuzz_campaign.py --seed $((S+2)) --seconds 200 --big > gpurun_out/fz2/big.txt 2>&1; tail -1 gpurun_out/fz2/big.txt
timeout -k 10 260 python tools/fuzz_campaign.py --seed $((S+3)) --seconds 200 > gpurun_out/fz2/plain.txt 2>&1; tail -1 gpurun_out/fz2/plain.txt
timeout -k 10 200 python tools/fuzz_campaign.py --seed $((S+4)) --seconds 150 --shadows > gpurun_out/fz2/shadows.txt 2>&1; tail -1 gpurun_out/fz2/shadows.txt
timeout -k 10 200 python tools/fuzz_campaign.py --seed $((S+5)) --seconds 100 --shadows --adversarial > gpurun_out/fz2/shadows_adv.txt 2>&1; tail -1 gpurun_out/fz2/shadows_adv.txt
