#!/bin/bash
# rocprofv3 kernel stats of single weight-stationary layers (FRAMES frames per call); output under gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wsl_$L -o p -- python3 $R/scripts/ws_layer.py $L > $R/gpurun_out/prof_wsl_$L.log 2>&1 </dev/null
  f=$(find $R/gpurun_out/prof_wsl_$L -name "*kernel_stats.csv" | head -1)
  echo "== $L"
  if [ -n "$f" ]; then grep "k_ws" "$f" | cut -d, -f1-5; else echo "no stats file"; fi
done
