#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_match_pose_gpu.py -m gpu -x -q -k "screen or reference_criteria or survivors" > $O/r4_gputest5.log 2>&1 || { tail -40 $O/r4_gputest5.log; exit 1; }
tail -2 $O/r4_gputest5.log
SHARES=0.0,0.3 REPS=10 timeout -k 10 300 python scripts/match_load_bench.py > $O/r4_matchload5.log 2>&1 || exit 2
tail -3 $O/r4_matchload5.log
timeout -k 10 600 python scripts/predator_rate_mt2.py > $O/r4_pred_mt2.log 2>&1 || { tail -20 $O/r4_pred_mt2.log; exit 3; }
tail -4 $O/r4_pred_mt2.log
cd /tmp && export TMPDIR=/tmp
SHARES=0.0,0.3 REPS=10 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_match2 -o m --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/match_load_bench.py > $GRAFT_REPO_ROOT/$O/r4_prof_match2.log 2>&1 || exit 4
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_match2/m_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for key in ('k_sample_screen',):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if key in r['Kernel_Name']]
    print(key, [round(x,1) for x in d])
PY
