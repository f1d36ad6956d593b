#!/bin/bash
# PMC counters of one weight-stationary layer (scripts/ws_layer.py); usage: pmc_ws.sh LAYER "CTR1 CTR2 ..." [more sets]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
L=$1; shift
i=0
for SET in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_wsl_${L}_$i -o p -- python3 $R/scripts/ws_layer.py $L > $R/gpurun_out/pmc_wsl_${L}_$i.log 2>&1 </dev/null
  f=$(find $R/gpurun_out/pmc_wsl_${L}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "k_ws_" in r["Kernel_Name"]:
        k = (r["Kernel_Name"].split("::")[-1].split("(")[0], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (s, n) in sorted(acc.items()):
    print(f"{k[0]:18s} {k[1]:34s} avg {s / n:16.1f}  (n={n})")
PY
  else echo "no counter file for set $i"; tail -3 $R/gpurun_out/pmc_wsl_${L}_$i.log; fi
done
