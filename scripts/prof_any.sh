#!/bin/bash
# rocprofv3 kernel stats of any script under scripts/: prof_any.sh <script.py> [pattern]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_any
cd $R/scripts
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_any -o p -- python3 $R/scripts/$1 > $R/gpurun_out/prof_any.log 2>&1 </dev/null
python3 $R/scripts/kstats.py "${2:-k_}" $R/gpurun_out/prof_any/p_kernel_stats.csv
