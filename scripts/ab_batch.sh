#!/bin/bash
# pairs/s over (pairs per step, streams) settings; args: "B:S" ...
for spec in "$@"; do
  B=${spec%%:*}; S=${spec#*:}
  v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps $((900 / B)) --pairs-per-step $B --streams $S </dev/null 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
  echo "B=$B streams=$S: $v pairs/s"
done
