#!/bin/bash
# rocprofv3 kernel stats of a short single-stream bench run; prints the kernels matching $1 (default: all top 25)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_s1q
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_s1q -o b -- python3 $R/bench.py --streams 1 --depth 1 --match-lanes 1 --steps 30 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_s1q.json 2> $R/gpurun_out/prof_s1q.err </dev/null || { echo "bench failed"; tail -5 $R/gpurun_out/prof_s1q.err; exit 1; }
python3 $R/scripts/kstats.py "${1:-k_}" $R/gpurun_out/prof_s1q/b_kernel_stats.csv
