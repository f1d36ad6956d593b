#!/bin/bash
# rocprofv3 kernel stats of NN + RANSAC under load (scripts/match_load_bench.py), SHARES / REPS from the environment
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_match
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_match -o p -- python3 $R/scripts/match_load_bench.py > $R/gpurun_out/prof_match.log 2>&1 </dev/null
grep "share" $R/gpurun_out/prof_match.log
python3 $R/scripts/kstats.py k_ $R/gpurun_out/prof_match/p_kernel_stats.csv | head -30
