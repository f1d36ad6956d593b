#!/bin/bash
# A/B matrix for the output-stationary conv (scripts/layer_bench.py on the 64-channel layers): weight-slice ring depth x
# ablation flags (1 = gathers from row 0, 2 = no slice staging, 4 = no MFMA, 8 = no accumulator write-back)
export FRAMES=${FRAMES:-12} ONLY=${ONLY:-b2tr,b2,c2tr}
for nbuf in 2 3; do
  for ab in 0 1 2 4 3 7 15; do
    echo "== NBUF=$nbuf ABLATE=$ab"
    APR_OS_NBUF=$nbuf APR_OS_ABLATE=$ab timeout -k 10 120 python scripts/layer_bench.py 2>&1 | grep " os R=" | sed -e 's/|.*| os/| os/' | cut -c1-40,60-200
  done
done
