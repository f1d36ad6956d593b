#!/bin/bash
for A in 15 31 47 63 79 16 64; do
echo "ablate $A"
APR_WREG_ABLATE=$A FRAMES=12 ONLY=f2tr timeout -k 10 300 python scripts/layer_bench.py 2>&1 | grep -E "f2tr" | sed 's/.*| os/os/' | cut -c1-60
done
