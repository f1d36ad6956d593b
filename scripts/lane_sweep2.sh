#!/bin/bash
# three streams: HW queue count x lanes
set -e
out=gpurun_out/lane_sweep2.log
: > $out
for cfg in "4 1" "8 1" "8 3" "12 3" "12 2" "2 1" "8 1"; do
  set -- $cfg
  echo "== hwq $1 lanes $2" >> $out
  GPU_MAX_HW_QUEUES=$1 APR_MATCH_LANES=$2 timeout -k 10 150 python bench.py --steps 120 --no-workloads --no-cpu-baseline --no-roofline 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out
done
