#!/bin/bash
# bench.py headline loop under --streams / --depth / --host combinations (no extra workloads): one JSON line each
set -e
out=gpurun_out/depth_sweep.log
: > $out
for cfg in "1 1 threads" "1 2 threads" "1 3 threads" "1 4 threads" "1 3 pipelined" "3 1 threads" "3 2 threads" "2 2 threads"; do
  set -- $cfg
  echo "== streams $1 depth $2 host $3" >> $out
  timeout -k 10 150 python bench.py --streams $1 --depth $2 --host $3 --steps 120 --no-workloads --no-cpu-baseline --no-roofline 2>/dev/null \
    | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))" >> $out
done
