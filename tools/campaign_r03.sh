#!/bin/bash
# round-3 parity campaign on the final library: seeds outside the suite's -> gpurun_out/r03_fuzz/campaign.txt
OUT=gpurun_out/r03_fuzz; mkdir -p $OUT
{
echo "# tools/fuzz_campaign.py on one MI355X, library sha256 $(sha256sum surf_renderer_amd/libsrh.so | cut -c1-12), commit $1"
timeout -k 10 200 python tools/fuzz_campaign.py --seed 3101 --seconds 120 | tail -1
timeout -k 10 200 python tools/fuzz_campaign.py --seed 3202 --seconds 120 --big | tail -1
timeout -k 10 200 python tools/fuzz_campaign.py --seed 3303 --seconds 100 --shadows | tail -1
timeout -k 10 200 python tools/fuzz_campaign.py --seed 3404 --seconds 100 --views | tail -1
timeout -k 10 300 python tools/fuzz_campaign.py --seed 3505 --seconds 200 --adversarial | tail -1
timeout -k 10 200 python tools/fuzz_campaign.py --seed 3707 --seconds 100 --shadows --adversarial | tail -1
for s in 2101 2102; do SRH_FUZZ_NONFINITE_SEED=$s SRH_FUZZ_NONFINITE_SCENES=2000 timeout -k 10 300 python -m pytest tests/test_hip_parity.py::test_fuzz_non_finite_and_degenerate_primitives -q 2>&1 | tail -1 | sed "s/^/non-finite seed $s (2000 scenes): /"; done
for s in 1101 1102; do SRH_FUZZ_BWD_SEED=$s SRH_FUZZ_BWD_SCENES=200 timeout -k 10 300 python -m pytest tests/test_hip_backward.py::test_fuzz_backward_against_the_gradient_oracles -q 2>&1 | tail -1 | sed "s/^/backward seed $s (200 scenes x 2 semantics): /"; done
} 2>&1 | tee $OUT/campaign.txt
