#!/bin/bash
# same-box A/B of environment settings on the default bench: ab_env.sh REPS "NAME=ENV1=V1,ENV2=V2" ...  ("NAME=" = defaults)
REPS=$1; shift
for i in $(seq $REPS); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    v=$(env $(echo "$envs" | tr ';' ' ') timeout -k 10 200 python bench.py --no-cpu-baseline --no-roofline </dev/null 2>/dev/null | python -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'],1))")
    echo "$name $v"
  done
done
