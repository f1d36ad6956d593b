#!/bin/bash
# pairs per step x steps in flight sweep of the headline loop (no roofline / workloads / cpu baseline)
for B in ${BS:-4 6 8 10 12}; do
  for S in ${SS:-3}; do
    python bench.py --pairs-per-step $B --streams $S --steps ${STEPS:-120} --no-roofline --no-workloads --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=$B S=$S', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],3), 'ms/step')"
  done
done
