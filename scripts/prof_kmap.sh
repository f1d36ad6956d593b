#!/bin/bash
# rocprofv3 kernel stats of the kernel-map builds of one 6-pair step (scripts/kmap_bench.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_kmap
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kmap -o p -- python3 $R/scripts/kmap_bench.py > $R/gpurun_out/prof_kmap.log 2>&1 </dev/null
grep "map (\|total" $R/gpurun_out/prof_kmap.log
python3 $R/scripts/kstats.py k_ $R/gpurun_out/prof_kmap/p_kernel_stats.csv | grep -i "kernel_map\|fill_upper"
