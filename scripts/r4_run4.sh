#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
SHARES=0.0,0.3 REPS=10 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_match -o m --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/match_load_bench.py > $O/r4_prof_match.log 2>&1 || { tail -20 $O/r4_prof_match.log; exit 1; }
cd $GRAFT_REPO_ROOT
tail -3 $O/r4_prof_match.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_match/m_kernel_stats.csv')))
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:9.2f} us {float(r['Percentage']):6.2f}")
PY
timeout -k 10 300 python scripts/predator_host_split.py > $O/r4_pred_split.log 2>&1 || { tail -20 $O/r4_pred_split.log; exit 2; }
tail -22 $O/r4_pred_split.log
