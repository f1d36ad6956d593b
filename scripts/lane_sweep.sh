#!/bin/bash
# headline loop with the matching/RANSAC batch dealt over 1 / 2 / 3 / 4 lanes, single stream and three streams
set -e
out=gpurun_out/lane_sweep.log
: > $out
for cfg in "1 1 1" "1 1 2" "1 1 3" "1 1 4" "1 3 3" "3 1 1" "3 1 2" "3 1 3" "3 1 1" "3 1 3"; do
  set -- $cfg
  echo "== streams $1 depth $2 lanes $3" >> $out
  APR_MATCH_LANES=$3 timeout -k 10 150 python bench.py --streams $1 --depth $2 --steps 120 --no-workloads --no-cpu-baseline --no-roofline 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out
done
