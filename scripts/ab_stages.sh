#!/bin/bash
# A/B of APR_WS_STAGES sets on the whole pipeline (pairs/s); args: name=stagelist ...
for spec in "$@"; do
  name=${spec%%=*}; st=${spec#*=}
  if [ "$st" = "default" ]; then unset APR_WS_STAGES; else export APR_WS_STAGES=$st; fi
  for rep in 1 2; do
    v=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline --steps 150 </dev/null 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "$name rep$rep: $v pairs/s"
  done
done
