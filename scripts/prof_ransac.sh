#!/bin/bash
# rocprofv3 kernel stats of the RANSAC stage alone (scripts/ransac_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ransac
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ransac -o p -- python3 $R/scripts/ransac_bench.py > $R/gpurun_out/prof_ransac.log 2>&1 </dev/null
grep "max_iter" $R/gpurun_out/prof_ransac.log
python3 $R/scripts/kstats.py k_ $R/gpurun_out/prof_ransac/p_kernel_stats.csv | grep -i "sample\|fit\|score\|select\|pack"
