#!/bin/bash
# rocprofv3 kernel stats of the Predator pair pipeline (scripts/bench_predator.py); output under gpurun_out/prof_pred
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_pred
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pred -o p -- python3 $R/scripts/bench_predator.py "$@" > $R/gpurun_out/prof_pred.log 2>&1 </dev/null
tail -8 $R/gpurun_out/prof_pred.log
