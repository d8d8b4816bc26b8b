#!/bin/bash
# extra campaign legs; every leg writes its own file under gpurun_out/fz2 (progress lines keep the run alive)
mkdir -p gpurun_out/fz2
timeout -k 10 330 python tools/fuzz_campaign.py --seed 5101 --seconds 280 --adversarial > gpurun_out/fz2/adv.txt 2>&1; tail -1 gpurun_out/fz2/adv.txt
timeout -k 10 260 python tools/fuzz_campaign.py --seed 5202 --seconds 220 --big > gpurun_out/fz2/big.txt 2>&1; tail -1 gpurun_out/fz2/big.txt
timeout -k 10 260 python tools/fuzz_campaign.py --seed 5303 --seconds 220 > gpurun_out/fz2/plain.txt 2>&1; tail -1 gpurun_out/fz2/plain.txt
timeout -k 10 200 python tools/fuzz_campaign.py --seed 5404 --seconds 160 --views > gpurun_out/fz2/views.txt 2>&1; tail -1 gpurun_out/fz2/views.txt
