#!/bin/bash
# headline loop over pairs-per-step x streams (no extra workloads)
out=gpurun_out/bs_sweep.log
: > $out
for cfg in "6 3" "8 3" "8 4" "6 4" "4 4" "10 3" "6 3"; do
  set -- $cfg
  echo "== pairs/step $1 streams $2" >> $out
  timeout -k 10 150 python bench.py --pairs-per-step $1 --streams $2 --steps 150 --no-workloads --no-cpu-baseline --no-roofline 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out || echo failed >> $out
done
